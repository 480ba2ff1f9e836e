"""Attention over points on the HIP kernels (csrc/attention.hip).

`attention_qkvpacked` is what `QKVMultiheadAttention` calls: it consumes the per-head interleaved
c_qkv output in place (transformer.py:71-72) and its backward writes one packed gradient buffer.
`flash_attn_func` keeps the third-party signature the reference imports (transformer.py:9-12,75).
"""
import math

import torch

from . import arena, check, dtype_code, lib, ptr, require_gpu, stream_ptr


def _fwd(q, k, v, scale):
    B, n, H, d = q.shape
    out = arena.empty((B, n, H, d), q.dtype, q.device)
    lse = arena.empty((B, H, n), torch.float32, q.device)
    assert q.stride() == k.stride() == v.stride() and q.stride(3) == 1
    if FWD_FP8 and q.dtype == torch.bfloat16:
        # opt-in: e4m3 operands on the block-scaled matrix instruction (csrc/attention.hip, attn_fwd_fp8_kernel); lengths it does not
        # cover (0 bytes of workspace) fall through to the bf16 kernel
        nbytes = lib().npcd_attn_fwd_fp8_workspace_bytes(B, n, H)
        if nbytes > 0:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
            check(_timed("fwd", lambda: lib().npcd_attn_fwd_fp8(ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(ws), B, n, H, d,
                                                                q.stride(0), q.stride(1), q.stride(2), out.stride(0), out.stride(1),
                                                                out.stride(2), scale, dtype_code(q), stream_ptr())), "npcd_attn_fwd_fp8")
            return out, lse
    nws = lib().npcd_attn_fwd_workspace_floats(B, n, H) if FWD_WS else 0
    ws = arena.empty(nws, torch.float32, q.device) if nws > 0 else None
    check(_timed("fwd", lambda: lib().npcd_attn_fwd_ws(ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(ws), B, n, H, d,
                                                       q.stride(0), q.stride(1), q.stride(2), out.stride(0), out.stride(1),
                                                       out.stride(2), scale, dtype_code(q), stream_ptr())), "npcd_attn_fwd_ws")
    return out, lse


# When set to a dict {"fwd": [], "dq": [], "dkdv": []}, every launch is bracketed by HIP events recorded
# on the launch stream (bench.py measures the kernels in situ with it).
KERNEL_EVENTS = None
KERNEL_TAGS = ("fwd", "dq", "dkdv", "bwd")      # the kernels _timed() distinguishes (bench.py builds the dict from this)


def _timed(tag, fn):
    if KERNEL_EVENTS is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn()
    e1.record()
    KERNEL_EVENTS[tag].append((e0, e1))
    return rc


import os

# "fused": the single-pass backward (attn_bwd_fused_kernel, 5 products); "twopass": dq pass + dk/dv pass (7 products)
BWD_MODE = os.environ.get("NPCD_ATTN_BWD", "twopass")
# "0": no scratch for the forward (a sequence of 256 j + 1 tokens then runs a workgroup for its last query row: A/B switch)
FWD_WS = os.environ.get("NPCD_ATTN_FWD_WS", "1") != "0"
# "1": the forward with fp8 (e4m3) operands (BASELINE configs[4] names fp8 attention); the backward stays on the bf16 kernels
FWD_FP8 = os.environ.get("NPCD_ATTN_FP8", "") == "1"


def colsum_part_for(dq):
    """Partial buffer for the column-sum by-product of _bwd (dq: the [B, n, H, 64] view of a packed [B, n, H, 192] gradient):
    (part [rows + scratch, 3 H 64] fp32, rows).  None in the single-pass mode."""
    if BWD_MODE == "fused":
        return None
    B, n, H, d = dq.shape
    rows = lib().npcd_attn_bwd_colsum_rows(B, n, H)
    return arena.empty((rows + lib().npcd_colsum_scratch_rows(), 3 * H * d), torch.float32, dq.device), rows


def _bwd(q, k, v, out, dout, lse, dq, dk, dv, scale, colsum_part=None):
    B, n, H, d = q.shape
    # row constants handed from pass 1 to pass 2 (+ the partial sums of the edge token's three gradient rows when n = 128 j + 1)
    delta = arena.empty(lib().npcd_attn_bwd_workspace_floats(B, n, H), torch.float32, q.device)
    assert dq.stride() == dk.stride() == dv.stride() and dout.stride() == out.stride()
    if BWD_MODE == "fused":
        nslab = lib().npcd_attn_bwd_fused_slab_floats(B, n, H)
        slab = torch.empty(nslab, dtype=torch.float32, device=q.device) if nslab > 0 else None
        check(_timed("bwd", lambda: lib().npcd_attn_bwd_fused(
            ptr(q), ptr(k), ptr(v), ptr(out), ptr(dout), ptr(lse), ptr(dq), ptr(dk), ptr(dv), ptr(delta), ptr(slab), B, n, H, d,
            q.stride(0), q.stride(1), q.stride(2), out.stride(0), out.stride(1), out.stride(2), dq.stride(0), dq.stride(1), dq.stride(2),
            scale, dtype_code(q), stream_ptr())), "npcd_attn_bwd_fused")
        return
    args = (ptr(q), ptr(k), ptr(v), ptr(out), ptr(dout), ptr(lse), ptr(dq), ptr(dk), ptr(dv),
            ptr(delta), B, n, H, d, q.stride(0), q.stride(1), q.stride(2),
            out.stride(0), out.stride(1), out.stride(2), dq.stride(0), dq.stride(1), dq.stride(2),
            scale, dtype_code(q), stream_ptr())
    if colsum_part is not None:       # column sums of the packed gradient as a by-product (csrc/attention.hip, AttnParams::colsum)
        cargs = args[:10] + (ptr(colsum_part),) + args[10:]
        if KERNEL_EVENTS is None:
            check(lib().npcd_attn_bwd_colsum(3, *cargs), "npcd_attn_bwd_colsum")
        else:
            check(_timed("dq", lambda: lib().npcd_attn_bwd_colsum(1, *cargs)), "npcd_attn_bwd_colsum(dq)")
            check(_timed("dkdv", lambda: lib().npcd_attn_bwd_colsum(2, *cargs)), "npcd_attn_bwd_colsum(dkdv)")
    elif KERNEL_EVENTS is None:
        check(lib().npcd_attn_bwd(*args), "npcd_attn_bwd")
    else:
        check(_timed("dq", lambda: lib().npcd_attn_bwd_pass(1, *args)), "npcd_attn_bwd_pass(dq)")
        check(_timed("dkdv", lambda: lib().npcd_attn_bwd_pass(2, *args)), "npcd_attn_bwd_pass(dkdv)")


def _check_input(x):
    require_gpu(x)
    if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise RuntimeError(f"HIP attention supports bf16/f16 (MFMA kernels) and fp32; got {x.dtype}")


def _fwd_f32(q, k, v, scale, want_lse=False):
    B, n, H, d = q.shape
    if d not in (32, 64, 128):
        raise RuntimeError(f"HIP attention is built for head dims 32, 64 and 128 (got head dim {d})")
    out = torch.empty((B, n, H, d), dtype=torch.float32, device=q.device)
    lse = torch.empty((B, H, n), dtype=torch.float32, device=q.device) if want_lse else None
    check(lib().npcd_attn_fwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), B, n, H, d, q.stride(0), q.stride(1), q.stride(2),
                              out.stride(0), out.stride(1), out.stride(2), scale, dtype_code(q), stream_ptr()), "npcd_attn_fwd(f32)")
    return (out, lse) if want_lse else out


def _f32_layout_ok(q, k, v):
    return (q.stride() == k.stride() == v.stride() and q.stride(3) == 1 and all(s % 4 == 0 for s in q.stride()[:3])
            and q.data_ptr() % 16 == 0 and k.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0)


class _AttnF32(torch.autograd.Function):
    """fp32 attention with gradients (`--dtype float32` training, where the reference runs its einsum path in fp32,
    transformer.py:76-83).  Forward: the exact-fp32 HIP kernel (+ the log-sum-exp of every row); backward: the two fp32
    matrix-instruction kernels of csrc/attention.hip (attn_bwd_f32_dq_kernel / attn_bwd_f32_dkdv_kernel: nothing is rounded to
    16 bits, deterministic).  The benchmarked training configuration is bf16 autocast and never comes here."""

    @staticmethod
    def forward(ctx, q, k, v, scale):
        if q.shape[-1] != 64:
            raise RuntimeError(f"HIP fp32 attention WITH GRADIENTS (--dtype float32 training) is built for head dim 64 (got head dim {q.shape[-1]}); "
                               "fp32 inference / sampling and the 16-bit kernels (bf16 / f16 autocast training) cover head dims 32, 64 and 128")
        if not _f32_layout_ok(q, k, v):
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out, lse = _fwd_f32(q, k, v, scale, want_lse=True)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, gout):
        q, k, v, out, lse = ctx.saved_tensors
        B, n, H, d = q.shape
        gout = gout.contiguous()
        dq, dk, dv = (torch.empty((B, n, H, d), dtype=torch.float32, device=q.device) for _ in range(3))
        delta = torch.empty(B * H * n, dtype=torch.float32, device=q.device)
        check(lib().npcd_attn_bwd(ptr(q), ptr(k), ptr(v), ptr(out), ptr(gout), ptr(lse), ptr(dq), ptr(dk), ptr(dv), ptr(delta),
                                  B, n, H, d, q.stride(0), q.stride(1), q.stride(2), out.stride(0), out.stride(1), out.stride(2),
                                  dq.stride(0), dq.stride(1), dq.stride(2), ctx.scale, dtype_code(q), stream_ptr()),
              "npcd_attn_bwd(f32)")
        return dq, dk, dv, None


def _f32(q, k, v, scale):
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _AttnF32.apply(q, k, v, scale)
    if not (q.stride() == k.stride() == v.stride() and q.stride(3) == 1):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    return _fwd_f32(q.detach(), k.detach(), v.detach(), scale)


class _AttnPacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, heads):
        B, n, w3 = qkv.shape
        d = w3 // heads // 3
        qkv = qkv.contiguous()
        x = qkv.view(B, n, heads, 3 * d)
        q, k, v = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
        scale = 1.0 / math.sqrt(d)
        out, lse = _fwd(q, k, v, scale)
        ctx.save_for_backward(qkv, out, lse)
        ctx.heads, ctx.scale = heads, scale
        return out.view(B, n, heads * d)

    @staticmethod
    def backward(ctx, gout):
        qkv, out, lse = ctx.saved_tensors
        B, n, w3 = qkv.shape
        H = ctx.heads
        d = w3 // H // 3
        x = qkv.view(B, n, H, 3 * d)
        q, k, v = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
        dqkv = torch.empty_like(qkv)
        g = dqkv.view(B, n, H, 3 * d)
        gout = gout.contiguous().view(B, n, H, d)
        _bwd(q, k, v, out, gout, lse, g[..., :d], g[..., d:2 * d], g[..., 2 * d:], ctx.scale)
        return dqkv, None


def attention_qkvpacked(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    """qkv [B, n, 3W] with head h at columns [3d*h, 3d*(h+1)) = q|k|v  ->  [B, n, W]."""
    _check_input(qkv)
    if qkv.dtype == torch.float32:
        B, n, w3 = qkv.shape
        d = w3 // heads // 3
        x = qkv.contiguous().view(B, n, heads, 3 * d)
        return _f32(x[..., :d], x[..., d:2 * d], x[..., 2 * d:], 1.0 / math.sqrt(d)).reshape(B, n, heads * d)
    return _AttnPacked.apply(qkv, heads)


class _AttnQKV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, scale):
        if not (q.stride() == k.stride() == v.stride() and q.stride(3) == 1
                and all(s % 8 == 0 for s in q.stride()[:3]) and q.data_ptr() % 16 == 0
                and k.data_ptr() % 16 == 0 and v.data_ptr() % 16 == 0):
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out, lse = _fwd(q, k, v, scale)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, gout):
        q, k, v, out, lse = ctx.saved_tensors
        B, n, H, d = q.shape
        g = torch.empty((B, n, H, 3 * d), dtype=q.dtype, device=q.device)
        dq, dk, dv = g[..., :d], g[..., d:2 * d], g[..., 2 * d:]
        _bwd(q, k, v, out, gout.contiguous(), lse, dq, dk, dv, ctx.scale)
        return dq, dk, dv, None


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, **unused):
    """Drop-in for flash_attn.flash_attn_func as the reference calls it (transformer.py:75):
    q, k, v [B, n, H, d] (any strides with a contiguous last dim), non-causal, no dropout."""
    if causal:
        raise NotImplementedError("the NPCD denoiser attends non-causally; causal=True is not implemented")
    if dropout_p:
        raise NotImplementedError("dropout is 0 everywhere in the reference (transformer.py:56,93,123,146,182)")
    for t in (q, k, v):
        _check_input(t)
    scale = 1.0 / math.sqrt(q.shape[-1]) if softmax_scale is None else float(softmax_scale)
    if q.dtype == torch.float32:
        return _f32(q, k, v, scale)
    return _AttnQKV.apply(q, k, v, scale)
