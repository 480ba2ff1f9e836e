"""Wrappers of the HBM-bound elementwise kernels (csrc/elementwise.hip)."""
import ctypes

import torch

from . import arena, check, dtype_code, lib, ptr, require_gpu, stream_ptr

_f32, _bf16, _f16 = torch.float32, torch.bfloat16, torch.float16
MAX_JOBS = 8          # NPCD_COLSUM_MAX_JOBS

# When set to a dict {"add_ln_fwd": [], "ln_bwd": [], "gelu_fwd": [], "gelu_bwd": []}, the main launch of each of these
# kernels is bracketed by HIP events recorded on the launch stream (bench.py measures the kernels in situ with it).
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


class ColsumJob(ctypes.Structure):
    """NpcdColsumJob (include/npcd_hip.h)."""
    _fields_ = [("part", ctypes.c_void_p), ("out", ctypes.c_void_p), ("nblk", ctypes.c_int), ("N", ctypes.c_int),
                ("accumulate", ctypes.c_int), ("reserved", ctypes.c_int)]


class ColsumBatch:
    """Collects the column sums a block's backward produces and finalises them with one pair of launches per
    MAX_JOBS.  Holds the partial buffers alive until `flush`."""

    def __init__(self):
        self.jobs = []

    def add(self, part, nblk, N, out, accumulate=False):
        self.jobs.append((part, nblk, N, out, accumulate))

    def flush(self):
        L, s = lib(), stream_ptr()
        for i in range(0, len(self.jobs), MAX_JOBS):
            chunk = self.jobs[i:i + MAX_JOBS]
            arr = (ColsumJob * len(chunk))()
            for a, (part, nblk, N, out, acc) in zip(arr, chunk):
                a.part, a.out, a.nblk, a.N, a.accumulate, a.reserved = part.data_ptr(), out.data_ptr(), nblk, N, int(acc), 0
            check(L.npcd_colsum_finalize_batch(ctypes.cast(arr, ctypes.c_void_p), len(chunk), s), "npcd_colsum_finalize_batch")
        self.jobs = []


def _finish(batch, part, nblk, N, out):
    if batch is not None:
        batch.add(part, nblk, N, out)
    else:
        check(lib().npcd_colsum_finalize(ptr(part), nblk, N, ptr(out), 0, stream_ptr()), "npcd_colsum_finalize")


def add_ln_fwd(x_in, delta, gamma, beta, eps=1e-5, want_sum=True, dtype=_bf16):
    """x_out = x_in + delta (fp32; None when delta is None), y = LayerNorm(x_out) in the 16-bit `dtype` (that of delta), mean, rstd."""
    require_gpu(x_in)
    T, W = x_in.shape
    dev = x_in.device
    if delta is not None:
        dtype = delta.dtype
    x_out = arena.empty_like(x_in) if (delta is not None and want_sum) else None
    y = arena.empty((T, W), dtype, dev)
    mean = arena.empty(T, _f32, dev)
    rstd = arena.empty(T, _f32, dev)
    check(_timed("add_ln_fwd" if delta is not None and want_sum else "ln_fwd",
                 lambda: lib().npcd_add_ln_fwd_dt(ptr(x_in), ptr(delta), ptr(gamma), ptr(beta), ptr(x_out), ptr(y), ptr(mean), ptr(rstd), T, W,
                                                  float(eps), dtype_code(y), stream_ptr())), "npcd_add_ln_fwd")
    return x_out, y, mean, rstd


def ln_bwd(dy, x, mean, rstd, gamma, dres, dgamma_out, dbeta_out, dcol_out=None, want_bf16=True, batch=None):
    """dx = LNbwd(dy) + dres (fp32) [+ bf16 copy]; writes dgamma/dbeta (and the column sum of dx) into the
    given fp32 [W] tensors (overwrite) -- immediately, or at `batch.flush()` when a ColsumBatch is given."""
    T, W = x.shape
    dev = x.device
    L = lib()
    nblk = L.npcd_ln_bwd_blocks(T)
    dx = arena.empty((T, W), _f32, dev)
    dxb = arena.empty((T, W), dy.dtype, dev) if want_bf16 else None          # (the run's 16-bit type)
    parts = arena.empty((3, nblk + L.npcd_colsum_scratch_rows(), W), _f32, dev)
    full = dres is not None and want_bf16          # the shape bench.py prices: dy, x, dres read; dx, dx(bf16) written
    check(_timed("ln_bwd" if full else "ln_bwd_partial",
                 lambda: L.npcd_ln_bwd_dt(ptr(dy), ptr(x), ptr(mean), ptr(rstd), ptr(gamma), ptr(dres), ptr(dx), ptr(dxb), ptr(parts[0]),
                                          ptr(parts[1]), ptr(parts[2]) if dcol_out is not None else ptr(None), T, W, dtype_code(dy), stream_ptr())),
          "npcd_ln_bwd")
    _finish(batch, parts[0], nblk, W, dgamma_out)
    _finish(batch, parts[1], nblk, W, dbeta_out)
    if dcol_out is not None:
        _finish(batch, parts[2], nblk, W, dcol_out)
    return dx, dxb


def gelu_fwd(h):
    g = arena.empty_like(h)
    check(_timed("gelu_fwd", lambda: lib().npcd_gelu_fwd_dt(ptr(h), ptr(g), h.numel(), dtype_code(h), stream_ptr())), "npcd_gelu_fwd")
    return g


def gelu_bwd(dg, h, dbias_out, batch=None, out=None, part_rows=None):
    """dh = dg * gelu'(h) (bf16); dbias_out[N] (fp32) = column sum of dh.  `part_rows`: a [>= npcd_colsum_blocks(T), N] fp32 view that
    receives the partial rows instead (the caller finishes the sum; dbias_out is then ignored): returns (dh, rows written)."""
    T, N = h.shape
    L = lib()
    nblk = L.npcd_colsum_blocks(T)
    dh = arena.empty_like(h) if out is None else out
    if part_rows is not None:
        check(_timed("gelu_bwd", lambda: L.npcd_gelu_bwd_dt(ptr(dg), ptr(h), ptr(dh), ptr(part_rows), T, N, dtype_code(h), stream_ptr())), "npcd_gelu_bwd")
        return dh, nblk
    part = arena.empty((nblk + L.npcd_colsum_scratch_rows(), N), _f32, h.device)
    check(_timed("gelu_bwd", lambda: L.npcd_gelu_bwd_dt(ptr(dg), ptr(h), ptr(dh), ptr(part), T, N, dtype_code(h), stream_ptr())), "npcd_gelu_bwd")
    _finish(batch, part, nblk, N, dbias_out)
    return dh


def colsum_bf16(a, out, batch=None):
    """out[N] (fp32) = column sum of the 16-bit (bf16 / f16) matrix a [T,N]."""
    T, N = a.shape
    L = lib()
    nblk = L.npcd_colsum_blocks(T)
    part = arena.empty((nblk + L.npcd_colsum_scratch_rows(), N), _f32, a.device)
    check(L.npcd_colsum_dt(ptr(a), ptr(part), T, N, dtype_code(a), stream_ptr()), "npcd_colsum")
    _finish(batch, part, nblk, N, out)
    return out


def adamw_ema(p, g, m, v, ema, shadow, lr, beta1, beta2, eps, weight_decay, step, ema_decay, zero_grad=True):
    require_gpu(p)
    check(lib().npcd_adamw_ema_dt(ptr(p), ptr(g), ptr(m), ptr(v), ptr(ema), ptr(shadow), dtype_code(shadow) if shadow is not None else 0,
                                  p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step),
                                  float(ema_decay if ema_decay is not None else 0.0), int(bool(zero_grad)), stream_ptr()), "npcd_adamw_ema")


def small_wgrad(dy, x):
    """dW [J, K] fp32 = dy^T @ x for bf16 dy [T, J <= 4], x [T, K] (K a power of two in 64..2048); None if the shape is not covered."""
    T, J = dy.shape
    K = x.shape[1]
    if not (1 <= J <= 4 and 64 <= K <= 2048 and K & (K - 1) == 0 and dy.dtype == _bf16 and x.dtype == _bf16 and dy.is_contiguous()
            and x.is_contiguous() and T > 0):
        return None
    L = lib()
    nblk = L.npcd_small_wgrad_blocks(T)
    part = arena.empty((nblk + L.npcd_colsum_scratch_rows(), J * K), _f32, x.device)
    out = arena.empty((J, K), _f32, x.device)
    check(L.npcd_small_wgrad(ptr(dy), ptr(x), ptr(part), T, J, K, stream_ptr()), "npcd_small_wgrad")
    check(L.npcd_colsum_finalize(ptr(part), nblk, J * K, ptr(out), 0, stream_ptr()), "npcd_colsum_finalize")
    return out


def wgrad(dy, x, out):
    """out [N, K] fp32 = dy[T, N]^T @ x[T, K] on the own split-T kernel (csrc/gemm.hip); False if the shape is not covered."""
    T, N = dy.shape
    K = x.shape[1]
    if (N % 256 or K % 256 or dy.dtype != x.dtype or dy.dtype not in (_bf16, _f16) or not dy.is_contiguous() or not x.is_contiguous()
            or not out.is_contiguous() or out.dtype != _f32):
        return False
    L = lib()
    S = L.npcd_wgrad_slices(T, N, K)
    ws = arena.empty((S, N, K), _f32, dy.device) if S > 1 else None
    check(L.npcd_wgrad(ptr(dy), ptr(x), ptr(out), ptr(ws), T, N, K, dtype_code(dy), stream_ptr()), "npcd_wgrad")
    return True


def wgrad_group(triples):
    """[(dy [T, N_g], x [T, K_g], out [N_g, K_g] fp32), ...] (<= 8, one T and one 16-bit dtype): all weight gradients in ONE launch of the
    own kernel, one workgroup per 256 x 256 tile over the whole token range (csrc/gemm.hip, npcd_wgrad_group); False if a shape is not
    covered (the caller's per-product path then runs)."""
    import ctypes
    n = len(triples)
    if not 1 <= n <= 8:
        return False
    T, dt = triples[0][0].shape[0], triples[0][0].dtype
    for dy, x, out in triples:
        if (dy.shape[0] != T or x.shape[0] != T or dy.dtype != dt or x.dtype != dt or dt not in (_bf16, _f16) or dy.shape[1] % 256 or x.shape[1] % 256
                or not dy.is_contiguous() or not x.is_contiguous() or not out.is_contiguous() or out.dtype != _f32
                or tuple(out.shape) != (dy.shape[1], x.shape[1])):
            return False
    P = ctypes.c_void_p * n
    I = ctypes.c_int * n
    check(lib().npcd_wgrad_group(n, P(*[ptr(t[0]) for t in triples]), P(*[ptr(t[1]) for t in triples]), P(*[ptr(t[2]) for t in triples]),
                                 I(*[t[0].shape[1] for t in triples]), I(*[t[1].shape[1] for t in triples]), T, dtype_code(triples[0][0]), stream_ptr()),
          "npcd_wgrad_group")
    return True


def sum_slices(part, out):
    """out = part.sum(dim=0) for fp32 part [S, ...] (S in 2, 4, 8), slices added in order; False if the shape is not covered."""
    S, n = part.shape[0], out.numel()
    if S not in (2, 4, 8) or n % 4 or not part.is_contiguous() or not out.is_contiguous() or part.dtype != _f32 or out.dtype != _f32:
        return False
    check(lib().npcd_sum_slices(ptr(part), ptr(out), S, n, stream_ptr()), "npcd_sum_slices")
    return True


def cast_f32_bf16(src, dst):
    """fp32 -> the 16-bit type of dst (bf16 or f16)"""
    check(lib().npcd_cast_f32_dt(ptr(src), ptr(dst), src.numel(), dtype_code(dst), stream_ptr()), "npcd_cast_f32")
    return dst


def split3(x, bias=None, gelu=False):
    """y = x [T, K] fp32 (+ bias [K]) (-> exact-erf GELU)  ->  [T, 3 K] bf16 = [hi(y) | lo(y) | hi(y)]: the activation operand of the
    fp32-class Linear layers (csrc/split.hip: one bf16 GEMM over the three cross products of split operands)."""
    require_gpu(x)
    x = x.contiguous()
    T, K = x.shape
    out = arena.empty((T, 3 * K), torch.bfloat16, x.device)
    b = None if bias is None else bias.to(torch.float32).contiguous()
    check(lib().npcd_split3_bf16(ptr(x), ptr(b), ptr(out), T, K, 1 if gelu else 0, stream_ptr()), "npcd_split3_bf16")
    return out


def add_ln_split3(x, gamma, beta, o=None, bias=None, eps=1e-5):
    """xnew = x (+ o + bias);  [T, 3 W] bf16 = [hi | lo | hi] of LayerNorm(xnew) * gamma + beta: (xnew or x itself, split) -- csrc/split.hip.
    Returns None for widths the kernel does not take (the caller composes it from torch's LayerNorm and split3)."""
    require_gpu(x)
    T, W = x.shape
    if W % 256 or W // 256 not in (1, 2, 3, 4, 8, 16):
        return None
    x = x.contiguous()
    out = arena.empty((T, 3 * W), torch.bfloat16, x.device)
    xnew = None
    if o is not None:
        o = o.contiguous()
        bias = bias.to(torch.float32).contiguous()
        xnew = arena.empty_like(x)
    check(lib().npcd_add_ln_split3_bf16(ptr(x), ptr(o), ptr(bias), ptr(gamma.contiguous()), ptr(beta.contiguous()), ptr(xnew), ptr(out), T, W,
                                        float(eps), stream_ptr()), "npcd_add_ln_split3_bf16")
    return (x if xnew is None else xnew), out


def ddpm_reverse_step(x_t, eps, noise, t, tables, clip=None, want_x0=False):
    """Fused reverse step of the sampler.  x_t / noise fp32 [B, ...], eps fp32 or bf16 (same shape), t int64 [B], tables = the five
    fp32 device tables (sqrt_recip_alphas_cumprod, sqrt_recipm1_alphas_cumprod, posterior_mean_coef1, posterior_mean_coef2,
    posterior_log_variance_clipped), clip = (lo, hi) floats or None -> x_{t-1} (and the clamped x0 when want_x0)."""
    require_gpu(x_t)
    x_t, noise, eps = x_t.contiguous(), noise.contiguous(), eps.contiguous()
    B = x_t.shape[0]
    out = arena.empty_like(x_t)
    x0 = arena.empty_like(x_t) if want_x0 else None
    code = {torch.float32: 2, torch.bfloat16: 0}[eps.dtype]
    lo, hi = (float(clip[0]), float(clip[1])) if clip is not None else (0.0, 0.0)
    check(lib().npcd_ddpm_reverse_step(ptr(x_t), ptr(eps), code, ptr(noise), ptr(out), ptr(x0), ptr(t), B, x_t.numel() // B,
                                       *[ptr(tb) for tb in tables], lo, hi, int(clip is not None), stream_ptr()), "npcd_ddpm_reverse_step")
    return out, x0


def q_sample(x0, noise, t, tab_sqrt_acp, tab_sqrt_1macp):
    """x_t = sqrt(acp[t]) x_0 + sqrt(1 - acp[t]) noise in one launch (coefficients looked up on the device).  fp32 [B, ...]."""
    require_gpu(x0, noise, t)
    x0, noise = x0.contiguous(), noise.contiguous()
    out = arena.empty_like(x0)
    B = x0.shape[0]
    check(lib().npcd_q_sample(ptr(x0), ptr(noise), ptr(t.contiguous()), ptr(tab_sqrt_acp), ptr(tab_sqrt_1macp), ptr(out), B, x0.numel() // B,
                              stream_ptr()), "npcd_q_sample")
    return out


class _EpsMSE(torch.autograd.Function):
    """mean((noise - eps)^2 / 2) with a one-launch forward (+ finalize) and a one-launch backward; eps fp32 or bf16."""

    @staticmethod
    def forward(ctx, eps, noise, want_pointwise):
        from . import dtype_code
        L = lib()
        eps_c, noise_c = eps.contiguous(), noise.contiguous()
        n = eps_c.numel()
        pw = arena.empty(eps_c.shape, _f32, eps.device) if want_pointwise else None
        part = arena.empty(L.npcd_eps_mse_blocks(), _f32, eps.device)
        loss = arena.empty(1, _f32, eps.device)
        check(L.npcd_eps_mse_fwd(ptr(noise_c), ptr(eps_c), dtype_code(eps_c), n, ptr(pw), ptr(part), ptr(loss), stream_ptr()), "npcd_eps_mse_fwd")
        ctx.save_for_backward(eps_c, noise_c)
        if pw is None:
            pw = loss.new_empty(0)
        ctx.mark_non_differentiable(pw)
        return loss[0], pw

    @staticmethod
    def backward(ctx, g, _gpw):
        from . import dtype_code
        eps, noise = ctx.saved_tensors
        grad = arena.empty_like(eps)
        up = g.reshape(1).to(_f32).contiguous()
        check(lib().npcd_eps_mse_bwd(ptr(noise), ptr(eps), dtype_code(eps), eps.numel(), ptr(up), ptr(grad), stream_ptr()), "npcd_eps_mse_bwd")
        return grad, None, None


def eps_mse(eps, noise, want_pointwise=True):
    """-> (loss scalar, pointwise fp32 tensor or None)."""
    loss, pw = _EpsMSE.apply(eps, noise, want_pointwise)
    return loss, (pw if want_pointwise else None)
