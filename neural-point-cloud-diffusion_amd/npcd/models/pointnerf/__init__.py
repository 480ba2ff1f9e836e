from .pointnerf import PointNeRF  # noqa: F401
