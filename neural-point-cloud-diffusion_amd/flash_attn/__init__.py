"""Drop-in for the `flash_attn` package as imported by the reference
(npcd/models/diffusion/denoisers/transformer.py:9-12): `from flash_attn import flash_attn_func`.
Backed by the gfx950 HIP kernels in libnpcd_hip.so."""
from npcd.hip.attention import flash_attn_func

__all__ = ["flash_attn_func"]
