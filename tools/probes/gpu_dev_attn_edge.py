"""Dev probe: the edge token's three gradient rows (sequence length 128 j + 1) against an fp32 reference."""
import sys, os, math
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import attention as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 129
B, H, d = 2, 3, 64
torch.manual_seed(0)
qkv = torch.randn(B, n, H, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
dout = torch.randn(B, n, H, d, device="cuda").bfloat16()
scale = 1 / math.sqrt(d)
out, lse = A._fwd(q, k, v, scale)
g = torch.zeros_like(qkv)
A._bwd(q, k, v, out, dout, lse, g[..., :d], g[..., d:2 * d], g[..., 2 * d:], scale)
qs = qkv.float().requires_grad_(True)
qq, kk, vv = (qs[..., i * d:(i + 1) * d].permute(0, 2, 1, 3) for i in range(3))
ref = torch.softmax(qq @ kk.transpose(-1, -2) * scale, -1) @ vv
ref.backward(dout.float().permute(0, 2, 1, 3))
rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
for name, i in (("dq", 0), ("dk", 1), ("dv", 2)):
    a, b = g[..., i * d:(i + 1) * d], qs.grad[..., i * d:(i + 1) * d]
    print(f"{name}: all rows {rel(a, b):.2e}   rows < n-1 {rel(a[:, :-1], b[:, :-1]):.2e}   edge row {rel(a[:, -1], b[:, -1]):.2e}")
    if rel(a[:, -1], b[:, -1]) > 1e-2:
        print("   got ", a[0, -1, 0, :8].float().tolist()); print("   want", b[0, -1, 0, :8].tolist())
        print("   ratio", (a[0, -1, 0, :64].float() / b[0, -1, 0, :64]).tolist())
