"""Wrappers of the HBM-bound elementwise kernels (csrc/elementwise.hip)."""
import torch

from . import check, lib, ptr, require_gpu, stream_ptr

_f32, _bf16 = torch.float32, torch.bfloat16


def add_ln_fwd(x_in, delta, gamma, beta, eps=1e-5, want_sum=True):
    """x_out = x_in + delta (fp32; None when delta is None), y = LayerNorm(x_out) bf16, mean, rstd."""
    require_gpu(x_in)
    T, W = x_in.shape
    dev = x_in.device
    x_out = torch.empty_like(x_in) if (delta is not None and want_sum) else None
    y = torch.empty((T, W), dtype=_bf16, device=dev)
    mean = torch.empty(T, dtype=_f32, device=dev)
    rstd = torch.empty(T, dtype=_f32, device=dev)
    check(lib().npcd_add_ln_fwd(ptr(x_in), ptr(delta), ptr(gamma), ptr(beta), ptr(x_out), ptr(y), ptr(mean), ptr(rstd), T, W,
                                float(eps), stream_ptr()), "npcd_add_ln_fwd")
    return x_out, y, mean, rstd


def ln_bwd(dy, x, mean, rstd, gamma, dres, dgamma_out, dbeta_out, dcol_out=None, want_bf16=True):
    """dx = LNbwd(dy) + dres (fp32) [+ bf16 copy]; writes dgamma/dbeta (and the column sum of dx) into the
    given fp32 [W] tensors (overwrite)."""
    T, W = x.shape
    dev = x.device
    L = lib()
    nblk = L.npcd_ln_bwd_blocks(T)
    dx = torch.empty((T, W), dtype=_f32, device=dev)
    dxb = torch.empty((T, W), dtype=_bf16, device=dev) if want_bf16 else None
    parts = torch.empty((3, nblk + L.npcd_colsum_scratch_rows(), W), dtype=_f32, device=dev)
    check(L.npcd_ln_bwd(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx), ptr(dxb), ptr(parts[0]), ptr(parts[1]),
                        ptr(parts[2]) if dcol_out is not None else ptr(None), T, W, stream_ptr()), "npcd_ln_bwd")
    s = stream_ptr()
    check(L.npcd_colsum_finalize(ptr(parts[0]), nblk, W, ptr(dgamma_out), 0, s), "npcd_colsum_finalize")
    check(L.npcd_colsum_finalize(ptr(parts[1]), nblk, W, ptr(dbeta_out), 0, s), "npcd_colsum_finalize")
    if dcol_out is not None:
        check(L.npcd_colsum_finalize(ptr(parts[2]), nblk, W, ptr(dcol_out), 0, s), "npcd_colsum_finalize")
    return dx, dxb


def gelu_fwd(h):
    g = torch.empty_like(h)
    check(lib().npcd_gelu_fwd(ptr(h), ptr(g), h.numel(), stream_ptr()), "npcd_gelu_fwd")
    return g


def gelu_bwd(dg, h, dbias_out):
    """dh = dg * gelu'(h) (bf16); dbias_out[N] (fp32) = column sum of dh."""
    T, N = h.shape
    L = lib()
    nblk = L.npcd_colsum_blocks(T)
    dh = torch.empty_like(h)
    part = torch.empty((nblk + L.npcd_colsum_scratch_rows(), N), dtype=_f32, device=h.device)
    check(L.npcd_gelu_bwd(ptr(dg), ptr(h), ptr(dh), ptr(part), T, N, stream_ptr()), "npcd_gelu_bwd")
    check(L.npcd_colsum_finalize(ptr(part), nblk, N, ptr(dbias_out), 0, stream_ptr()), "npcd_colsum_finalize")
    return dh


def colsum_bf16(a, out):
    """out[N] (fp32) = column sum of the bf16 matrix a [T,N]."""
    T, N = a.shape
    L = lib()
    nblk = L.npcd_colsum_blocks(T)
    part = torch.empty((nblk + L.npcd_colsum_scratch_rows(), N), dtype=_f32, device=a.device)
    check(L.npcd_colsum_bf16(ptr(a), ptr(part), T, N, stream_ptr()), "npcd_colsum_bf16")
    check(L.npcd_colsum_finalize(ptr(part), nblk, N, ptr(out), 0, stream_ptr()), "npcd_colsum_finalize")
    return out


def adamw_ema(p, g, m, v, ema, shadow, lr, beta1, beta2, eps, weight_decay, step, ema_decay, zero_grad=True):
    require_gpu(p)
    check(lib().npcd_adamw_ema(ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), ptr(shadow), p.numel(), float(lr), float(beta1), float(beta2),
                               float(eps), float(weight_decay), int(step), float(ema_decay if ema_decay is not None else 0.0),
                               int(bool(zero_grad)), stream_ptr()), "npcd_adamw_ema")


def cast_f32_bf16(src, dst):
    check(lib().npcd_cast_f32_bf16(ptr(src), ptr(dst), src.numel(), stream_ptr()), "npcd_cast_f32_bf16")
    return dst
