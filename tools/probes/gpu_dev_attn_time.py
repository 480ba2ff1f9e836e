"""Dev probe: per-kernel time of the attention kernels at cfg-D (B=64, n=513, H=16, d=64) + error vs an fp32 reference.
usage: python3 tools/probes/gpu_dev_attn_time.py [reps [n [B [head dim]]]]"""
import sys, os, math
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import attention as A
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, n, H, d = (int(sys.argv[3]) if len(sys.argv) > 3 else 64), (int(sys.argv[2]) if len(sys.argv) > 2 else 513), 16, (int(sys.argv[4]) if len(sys.argv) > 4 else 64)
torch.manual_seed(0)
qkv = torch.randn(B, n, H, 3 * d, device="cuda").bfloat16()
if os.environ.get("NPCD_ZERO_DATA"):      # clock check: all-zero operands draw less power (MI355X_MICROARCH.md, DVFS give-back)
    qkv.zero_()
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
dout = torch.randn(B, n, H, d, device="cuda").bfloat16()
if os.environ.get("NPCD_ZERO_DATA"):
    dout.zero_()
scale = 1 / math.sqrt(d)
# correctness on a slice (fp32 reference)
out, lse = A._fwd(q, k, v, scale)
g = torch.empty_like(qkv)
A._bwd(q, k, v, out, dout, lse, g[..., :d], g[..., d:2 * d], g[..., 2 * d:], scale)
qs = qkv[:2].float().requires_grad_(True)
qq, kk, vv = (qs[..., i * d:(i + 1) * d].permute(0, 2, 1, 3) for i in range(3))
ref = torch.softmax(qq @ kk.transpose(-1, -2) * scale, -1) @ vv
ref.backward(dout[:2].float().permute(0, 2, 1, 3))
rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
print(f"rel-L2 out {rel(out[:2], ref.permute(0, 2, 1, 3)):.2e}  dqkv {rel(g[:2], qs.grad):.2e}", flush=True)
A.KERNEL_EVENTS = {k: [] for k in A.KERNEL_TAGS}
for _ in range(reps):
    out, lse = A._fwd(q, k, v, scale)
    A._bwd(q, k, v, out, dout, lse, g[..., :d], g[..., d:2 * d], g[..., 2 * d:], scale)
torch.cuda.synchronize()
fl = 4 * B * H * n * n * d
for tag, mult in (("fwd", 1.0), ("dq", 1.5), ("dkdv", 2.0), ("bwd", 2.5)):      # executed products: 2 / 3 / 4 / 5 (fused) x (2 B H n^2 d)
    if not A.KERNEL_EVENTS[tag]:
        continue
    ts = sorted(a.elapsed_time(b) for a, b in A.KERNEL_EVENTS[tag][3:])
    med = ts[len(ts) // 2]
    print(f"{tag}: median {med * 1e3:.1f} us  min {ts[0] * 1e3:.1f} us  {fl * mult / med / 1e9:.0f} TFLOP/s", flush=True)
