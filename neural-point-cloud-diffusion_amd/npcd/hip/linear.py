"""Wrappers of the own NT GEMMs with fused epilogues (csrc/gemm_nt.hip): the Linear layers of the residual block
(reference transformer.py:67, :107-115, :118-137) and their data gradients."""
import torch

from . import arena, check, dtype_code, lib, ptr, require_gpu, stream_ptr

_f32 = torch.float32

# When set to a dict {"fwd": [], "gelu_fwd": [], "dgelu_bwd": []}, each launch is bracketed by HIP events on the launch stream
KERNEL_EVENTS = None


def _timed(tag, fn):
    if KERNEL_EVENTS is None or tag not in KERNEL_EVENTS:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    KERNEL_EVENTS[tag].append((e0, e1))
    return rc


def supported(M: int, N: int, K: int) -> bool:
    return M >= 1 and N % 1024 == 0 and K % 64 == 0 and M * N < 2 ** 31 and M * K < 2 ** 31


def _check_operands(x, w):
    require_gpu(x, w)
    if x.dtype != w.dtype or x.dtype not in (torch.bfloat16, torch.float16):
        raise RuntimeError(f"npcd linear: 16-bit operands of one type expected, got {x.dtype} / {w.dtype}")
    if not (x.is_contiguous() and w.is_contiguous()) or x.shape[1] != w.shape[1]:
        raise RuntimeError("npcd linear: x [M, K] and w [N, K] must be contiguous row-major with the same K")


def linear_fwd(x, w, bias=None, out=None):
    """y [M, N] = x [M, K] @ w [N, K]^T (+ bias [N], 16 bit)."""
    _check_operands(x, w)
    M, K = x.shape
    N = w.shape[0]
    y = arena.empty((M, N), x.dtype, x.device) if out is None else out
    check(_timed("fwd", lambda: lib().npcd_linear_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), M, N, K, dtype_code(x), stream_ptr())), "npcd_linear_fwd")
    return y


def supported128(M: int, N: int, K: int) -> bool:
    return M >= 1 and N % 128 == 0 and K % 64 == 0 and M * N < 2 ** 31 and M * K < 2 ** 30 and N * K < 2 ** 30


def linear128_fwd(x, w, bias=None, out=None):
    """y [M, N] = x [M, K] @ w [N, K]^T (+ bias) on 128 x 128 tiles: the form for a few thousand rows (one rank of the 4- / 8-GPU job)."""
    _check_operands(x, w)
    M, K = x.shape
    N = w.shape[0]
    y = arena.empty((M, N), x.dtype, x.device) if out is None else out
    check(_timed("fwd128", lambda: lib().npcd_linear128_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), M, N, K, dtype_code(x), stream_ptr())), "npcd_linear128_fwd")
    return y


def linear_gelu_fwd(x, w, bias):
    """h = x @ w^T + bias (rounded to 16 bit), g = gelu_erf(h): (h, g)."""
    _check_operands(x, w)
    M, K = x.shape
    N = w.shape[0]
    h = arena.empty((M, N), x.dtype, x.device)
    g = arena.empty_like(h)
    check(_timed("gelu_fwd", lambda: lib().npcd_linear_gelu_fwd(ptr(x), ptr(w), ptr(bias), ptr(h), ptr(g), M, N, K, dtype_code(x), stream_ptr())),
          "npcd_linear_gelu_fwd")
    return h, g


def linear_dgelu_bwd(dy, wt, h, out=None, extra_part_rows=0):
    """dh = round16(dy @ wt^T) * gelu_erf'(h) and the fp32 column partial sums of dh: (dh, part, rows) -- wt [N, K] is the
    TRANSPOSED weight of the layer after the GELU; finish the bias gradient with npcd_colsum_finalize(part, rows, N, ...).
    `extra_part_rows`: room behind the kernel's `rows` partial rows for a caller that adds partial rows of other token ranges."""
    _check_operands(dy, wt)
    M, K = dy.shape
    N = wt.shape[0]
    L = lib()
    rows = L.npcd_linear_dgelu_rows(M)
    dh = arena.empty((M, N), dy.dtype, dy.device) if out is None else out
    part = arena.empty((rows + extra_part_rows + L.npcd_colsum_scratch_rows(), N), _f32, dy.device)
    check(_timed("dgelu_bwd", lambda: L.npcd_linear_dgelu_bwd(ptr(dy), ptr(wt), ptr(h), ptr(dh), ptr(part), M, N, K, dtype_code(dy), stream_ptr())),
          "npcd_linear_dgelu_bwd")
    return dh, part, rows


def transpose16(w, out=None):
    """[R, C] 16-bit -> [C, R]."""
    require_gpu(w)
    R, C = w.shape
    o = arena.empty((C, R), w.dtype, w.device) if out is None else out
    check(lib().npcd_transpose_16(ptr(w), ptr(o), R, C, stream_ptr()), "npcd_transpose_16")
    return o
