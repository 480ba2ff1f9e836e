"""ctypes binding of libnpcd_hip.so (C ABI: include/npcd_hip.h).

The library is built in-tree by ``csrc/build.py`` (``__graft_entry__.build()``).  Nothing in here
falls back to PyTorch/CPU: if the shared object cannot be loaded, or an operator is handed a
non-GPU tensor, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# NPCD_HIP_LIB: development probes load a DIAGNOSTIC build of the library (tools/); never set in the product / tests / bench
LIB_PATH = os.environ.get("NPCD_HIP_LIB") or os.path.join(_PKG_ROOT, "lib", "libnpcd_hip.so")

NPCD_BF16, NPCD_F16, NPCD_F32 = 0, 1, 2
NPCD_GRID_FINE, NPCD_GRID_SCALED = 0, 1
_DTYPE_CODE = {torch.bfloat16: NPCD_BF16, torch.float16: NPCD_F16, torch.float32: NPCD_F32}


class GridParams(ctypes.Structure):
    """struct npcd_grid_params (include/npcd_hip.h)."""
    _fields_ = [("voxel_size", c_float * 3), ("voxel_scale", c_int32 * 3), ("kernel_size", c_int32 * 3),
                ("max_points_per_voxel", c_int32), ("max_occ_voxels_per_example", c_int32),
                ("range_min", c_float * 3), ("range_max", c_float * 3), ("dims", c_int32 * 3), ("cdims", c_int32 * 3),
                ("grid_level", c_int32)]


_P = c_void_p
# name -> (restype, argtypes); must list every symbol include/npcd_hip.h declares
SIGNATURES = {
    "npcd_abi_version": (c_int, []),
    "npcd_error_string": (c_char_p, [c_int]),
    "npcd_last_hip_error": (c_char_p, []),
    "npcd_attn_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int] + [c_int64] * 6 + [c_float, c_int, _P]),
    "npcd_attn_fwd_workspace_floats": (c_int64, [c_int, c_int, c_int]),
    "npcd_attn_fwd_ws": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int] + [c_int64] * 6 + [c_float, c_int, _P]),
    "npcd_ray_march_bwd": (c_int, [_P] * 8 + [c_int, c_int, c_int] + [_P] * 6 + [_P]),
    "npcd_leaky_bwd_blocks": (c_int, [c_int64]),
    "npcd_leaky_bwd_colsum": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_float, c_int, _P]),
    "npcd_pair_input_fwd": (c_int, [_P] * 5 + [c_int, c_int, c_int64, _P, _P, _P]),
    "npcd_pair_input_bwd": (c_int, [_P, _P, c_int, c_int, c_int64, _P, _P]),
    "npcd_pair_aggregate": (c_int, [c_int, _P, _P, _P, _P, c_int, c_int64, _P, _P]),
    "npcd_attn_bwd": (c_int, [_P] * 10 + [c_int] * 4 + [c_int64] * 9 + [c_float, c_int, _P]),
    "npcd_attn_bwd_workspace_floats": (c_int64, [c_int, c_int, c_int]),
    "npcd_attn_bwd_colsum_rows": (c_int, [c_int, c_int, c_int]),
    "npcd_attn_bwd_colsum": (c_int, [c_int] + [_P] * 11 + [c_int] * 4 + [c_int64] * 9 + [c_float, c_int, _P]),
    "npcd_attn_fwd_fp8_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "npcd_attn_fwd_fp8": (c_int, [_P] * 6 + [c_int] * 4 + [c_int64] * 6 + [c_float, c_int, _P]),
    "npcd_attn_bwd_fused_slab_floats": (c_int64, [c_int, c_int, c_int]),
    "npcd_attn_bwd_fused": (c_int, [_P] * 11 + [c_int] * 4 + [c_int64] * 9 + [c_float, c_int, _P]),
    "npcd_attn_bwd_pass": (c_int, [c_int] + [_P] * 10 + [c_int] * 4 + [c_int64] * 9 + [c_float, c_int, _P]),
    "npcd_grid_workspace_bytes": (c_int64, [POINTER(GridParams), c_int, c_int]),
    "npcd_grid_build": (c_int, [POINTER(GridParams), _P, _P, c_int, c_int, _P, _P]),
    "npcd_grid_query": (c_int, [POINTER(GridParams), _P, _P] + [c_int] * 6 + [c_float, c_int] + [_P] * 9 + [_P]),
    "npcd_grid_query_compact": (c_int, [POINTER(GridParams), _P, _P] + [c_int] * 6 + [c_float] + [_P] * 5 + [c_int32] + [_P] * 5 + [_P]),
    "npcd_grid_query_compact_ordered": (c_int, [POINTER(GridParams), _P, _P] + [c_int] * 6 + [c_float] + [_P] * 5 + [c_int32] + [_P] * 5
                                        + [_P, _P]),
    "npcd_grid_query_order_ws_bytes": (c_int64, [c_int] * 4),
    "npcd_ray_march_compact": (c_int, [_P] * 8 + [c_int, c_int, c_int, c_int] + [_P] * 4 + [_P]),
    "npcd_render_lim_words": (c_int64, [c_int, c_int]),
    "npcd_render_rays_query": (c_int, [POINTER(GridParams), _P, _P, c_int, c_int, _P, _P, c_int, c_int, c_float, c_int, c_int, c_int, c_float]
                               + [_P] * 5 + [c_int32] + [_P] * 7 + [_P]),
    "npcd_ray_march_compact_fused": (c_int, [_P] * 10 + [c_int, c_int, c_int, c_int, c_int] + [_P] * 4 + [_P]),
    "npcd_ray_gen": (c_int, [_P, _P, c_int, c_int, c_float] + [_P] * 5 + [_P]),
    "npcd_ray_gen_subset": (c_int, [_P, _P, c_int, c_int, c_float, _P, c_int] + [_P] * 5 + [_P]),
    "npcd_shade_wpack_bytes": (c_int64, [c_int, c_int, c_int]),
    "npcd_shade_workspace_bytes": (c_int64, [c_int, c_int]),
    "npcd_shade_pack_weights": (c_int, [POINTER(_P), POINTER(_P), c_int, c_int, c_int, _P]),
    "npcd_pairs_x2_wpack_bytes": (c_int64, [c_int]),
    "npcd_pairs_x2_pack": (c_int, [POINTER(_P), POINTER(_P), c_int, _P]),
    "npcd_pairs_x2": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "npcd_points_x2_wpack_bytes": (c_int64, []),
    "npcd_points_x2_pack": (c_int, [POINTER(_P), POINTER(_P), c_int, _P]),
    "npcd_points_x2": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "npcd_points_x2_train": (c_int, [_P, _P, c_int, _P, _P, _P]),
    "npcd_points_x2_pack_dev": (c_int, [POINTER(_P), POINTER(_P), c_int, _P, _P]),
    "npcd_shade_points": (c_int, [_P, c_int, c_int, c_int] + [_P] * 5 + [c_int, c_int] + [_P] * 3 + [_P] + [_P]),
    "npcd_shade_points_dir": (c_int, [_P, c_int, c_int, c_int] + [_P] * 5 + [c_int, c_int] + [_P] * 3 + [_P, _P] + [_P] + [_P]),
    "npcd_ray_march_ws_floats": (c_int64, [c_int]),
    "npcd_ray_gen_ws_floats": (c_int64, [c_int, c_int, c_int]),
    "npcd_ray_march": (c_int, [_P] * 8 + [c_int, c_int, c_int] + [_P] * 4 + [_P]),
    "npcd_add_ln_fwd": (c_int, [_P] * 8 + [c_int, c_int, c_float, _P]),
    "npcd_ln_bwd_blocks": (c_int, [c_int]),
    "npcd_ln_bwd": (c_int, [_P] * 11 + [c_int, c_int, _P]),
    "npcd_colsum_scratch_rows": (c_int, []),
    "npcd_colsum_finalize": (c_int, [_P, c_int, c_int, _P, c_int, _P]),
    "npcd_colsum_finalize_batch": (c_int, [_P, c_int, _P]),
    "npcd_ddpm_reverse_step": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_int, c_int64, _P, _P, _P, _P, _P, c_float, c_float, c_int, _P]),
    "npcd_gelu_fwd": (c_int, [_P, _P, c_int64, _P]),
    "npcd_colsum_blocks": (c_int, [c_int]),
    "npcd_gelu_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "npcd_colsum_bf16": (c_int, [_P, _P, c_int, c_int, _P]),
    "npcd_adamw_ema": (c_int, [_P] * 6 + [c_int64] + [c_float] * 5 + [c_int, c_float, c_int, _P]),
    "npcd_cast_f32_bf16": (c_int, [_P, _P, c_int64, _P]),
    "npcd_add_ln_fwd_dt": (c_int, [_P] * 8 + [c_int, c_int, c_float, c_int, _P]),
    "npcd_ln_bwd_dt": (c_int, [_P] * 11 + [c_int, c_int, c_int, _P]),
    "npcd_gelu_fwd_dt": (c_int, [_P, _P, c_int64, c_int, _P]),
    "npcd_gelu_bwd_dt": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "npcd_colsum_dt": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "npcd_adamw_ema_dt": (c_int, [_P] * 6 + [c_int, c_int64] + [c_float] * 5 + [c_int, c_float, c_int, _P]),
    "npcd_cast_f32_dt": (c_int, [_P, _P, c_int64, c_int, _P]),
    "npcd_wgrad_slices": (c_int, [c_int, c_int, c_int]),
    "npcd_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "npcd_wgrad_group": (c_int, [c_int, POINTER(_P), POINTER(_P), POINTER(_P), POINTER(c_int), POINTER(c_int), c_int, c_int, _P]),
    "npcd_sum_slices": (c_int, [_P, _P, c_int, c_int64, _P]),
    "npcd_linear_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "npcd_linear128_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "npcd_linear_gelu_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "npcd_linear_dgelu_rows": (c_int, [c_int]),
    "npcd_linear_dgelu_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "npcd_transpose_16": (c_int, [_P, _P, c_int, c_int, _P]),
    "npcd_small_wgrad_blocks": (c_int, [c_int]),
    "npcd_small_wgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "npcd_q_sample": (c_int, [_P] * 6 + [c_int, c_int64, _P]),
    "npcd_eps_mse_blocks": (c_int, []),
    "npcd_eps_mse_fwd": (c_int, [_P, _P, c_int, c_int64, _P, _P, _P, _P]),
    "npcd_eps_mse_bwd": (c_int, [_P, _P, c_int, c_int64, _P, _P, _P]),
    "npcd_split3_bf16": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P]),
    "npcd_add_ln_split3_bf16": (c_int, [_P] * 7 + [c_int64, c_int, c_float, _P]),
    "npcd_pair_mlp_wpack_bytes": (c_int64, [c_int, c_int]),
    "npcd_pair_mlp_pack": (c_int, [POINTER(_P), POINTER(_P), c_int, c_int, _P, _P]),
    "npcd_pair_mlp_fwd": (c_int, [_P, c_int, c_int] + [_P] * 5 + [c_int64, c_int, c_int64] + [_P] * 4 + [_P]),
    "npcd_pair_mlp_bwd_slabs": (c_int, [c_int64, c_int]),
    "npcd_pair_mlp_bwd_workspace_floats": (c_int64, [c_int, c_int64, c_int]),
    "npcd_pair_mlp_bwd": (c_int, [_P, c_int, c_int] + [_P] * 5 + [c_int64] + [_P] * 3 + [POINTER(_P), POINTER(_P), _P]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Load (once) and return the native library; raise loudly if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libnpcd_hip.so not found at {LIB_PATH}: build it with "
                f"`python neural-point-cloud-diffusion_amd/csrc/build.py` (or __graft_entry__.build()). "
                f"There is no CPU / PyTorch fallback for the NPCD hot path.")
        handle = ctypes.CDLL(LIB_PATH)
        missing = []
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                missing.append(name)
                continue
            fn.restype, fn.argtypes = res, args
        handle.npcd_missing = tuple(missing)      # calling a missing symbol raises AttributeError
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        L = lib()
        msg = L.npcd_error_string(rc).decode()
        if rc == -3:
            msg += ": " + L.npcd_last_hip_error().decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")


def require_gpu(*tensors: torch.Tensor):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("npcd.hip operators need GPU tensors (MI355X); got a tensor on "
                               f"{t.device}. There is no CPU fallback in the product path.")


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPE_CODE[t.dtype]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {t.dtype}") from None


def ptr(t):
    return c_void_p(0 if t is None else t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """torch's current HIP stream of the current device as a pointer.  (`torch.cuda.current_stream().cuda_stream` costs ~9 us of Python per
    call -- 126 calls = 1.2 ms of a rank's 17-ms step at per-GPU batch 8, which is host-bound; the raw accessor is one C call.)"""
    if _raw_stream is not None:
        return c_void_p(_raw_stream(torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream().cuda_stream)
