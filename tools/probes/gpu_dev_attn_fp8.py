"""Dev probe: the fp8 (e4m3, block-scaled MFMA) attention forward against the bf16 kernel and an fp32 reference: error and time.
usage: python3 tools/probes/gpu_dev_attn_fp8.py [n] [B] [reps]"""
import sys, os, math
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import attention as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
H, d = 16, 64
torch.manual_seed(0)
qkv = torch.randn(B, n, H, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
scale = 1 / math.sqrt(d)
nb = min(B, 2)
qq, kk, vv = (x[:nb].float().permute(0, 2, 1, 3) for x in (q, k, v))
sc = qq @ kk.transpose(-1, -2) * scale
ref = (torch.softmax(sc, -1) @ vv).permute(0, 2, 1, 3)
ref_lse = torch.logsumexp(sc, -1)
rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
res = {}
for mode in ("bf16", "fp8"):
    A.FWD_FP8 = mode == "fp8"
    out, lse = A._fwd(q, k, v, scale)
    torch.cuda.synchronize()
    print(f"{mode}: out rel-L2 {rel(out[:nb], ref):.3e}  max-abs {float((out[:nb].float() - ref).abs().max()):.3e} (max |ref| {float(ref.abs().max()):.2f})  "
          f"lse max-abs {float((lse[:nb] - ref_lse).abs().max()):.3e}  finite {bool(torch.isfinite(out).all())}", flush=True)
    A.KERNEL_EVENTS = {t: [] for t in A.KERNEL_TAGS}
    for _ in range(reps):
        A._fwd(q, k, v, scale)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in A.KERNEL_EVENTS["fwd"][3:])
    A.KERNEL_EVENTS = None
    res[mode] = ts[len(ts) // 2] * 1e3
    print(f"{mode}: median {res[mode]:.1f} us (min {ts[0] * 1e3:.1f}) = {4 * B * H * n * n * d / res[mode] / 1e6:.0f} TFLOP/s", flush=True)
print(f"fp8 / bf16 time: {res['fp8'] / res['bf16']:.3f}")
# gradients through the bf16 backward kernels against each forward's LSE
dout = torch.randn(nb, n, H, d, device="cuda").bfloat16()
qs = qkv[:nb].float().requires_grad_(True)
q3, k3, v3 = (qs[..., i * d:(i + 1) * d].permute(0, 2, 1, 3) for i in range(3))
(torch.softmax(q3 @ k3.transpose(-1, -2) * scale, -1) @ v3).backward(dout.float().permute(0, 2, 1, 3))
for mode in ("bf16", "fp8"):
    A.FWD_FP8 = mode == "fp8"
    x = qkv[:nb].clone()
    out, lse = A._fwd(x[..., :d], x[..., d:2 * d], x[..., 2 * d:], scale)
    g = torch.empty_like(x)
    A._bwd(x[..., :d], x[..., d:2 * d], x[..., 2 * d:], out, dout, lse, g[..., :d], g[..., d:2 * d], g[..., 2 * d:], scale)
    print(f"{mode} forward -> backward: " + "  ".join(f"d{nm} rel-L2 {rel(g[..., i * d:(i + 1) * d], qs.grad[..., i * d:(i + 1) * d]):.3e}" for i, nm in enumerate("qkv")))

