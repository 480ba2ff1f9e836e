from .engine import DiffusionTrainer, FlatBuffers, GradReducer  # noqa: F401
from .pointnerf_engine import PointNeRFTrainer  # noqa: F401
