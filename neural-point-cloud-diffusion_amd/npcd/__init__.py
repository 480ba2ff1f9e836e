"""npcd -- MI355X-native implementation of the neural-point-cloud-diffusion hot path.

Mirrors the reference's ``npcd.models`` class surface (NPCD, DiffusionModel, NPCDTransformer,
PointNeRF) so that ``train_diffusion.py`` / ``eval_pointnerf.py`` style drivers can call it as a
drop-in; the attention, k-NN, shading and ray-march operators run as hand-written HIP kernels
from ``libnpcd_hip.so`` (see ``npcd.hip``).  There is no CPU fallback: the operators raise when
the library or a GPU is missing.
"""
__all__ = ["hip", "models"]
