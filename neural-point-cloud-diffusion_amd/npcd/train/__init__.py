from .engine import DiffusionTrainer, FlatBuffers, GradReducer  # noqa: F401
