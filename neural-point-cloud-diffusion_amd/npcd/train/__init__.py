from .engine import DiffusionTrainer, FlatBuffers, GradReducer  # noqa: F401
from .pointnerf_engine import PointNeRFTrainer  # noqa: F401
from .checkpoint import load_trainer_state, resume_latest, save_train_state, trainer_state_dict  # noqa: F401
