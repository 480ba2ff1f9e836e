from .diffusion_model import DiffusionModel, MinusOneToOneNormalization, UnitGaussianNormalization  # noqa: F401
from .gaussian_diffusion import GaussianDiffusion  # noqa: F401
from .transformer import NPCDTransformer  # noqa: F401
