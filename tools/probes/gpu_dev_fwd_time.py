"""Dev probe: HIP-event time of the default attention forward at the benchmark shape (B = 64, n = 513, 16 heads) -- run once per
library build (NPCD_HIP_LIB) in alternation to compare two builds on one box.  usage: python3 tools/probes/gpu_dev_fwd_time.py [rounds]"""
import sys, os, math
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import attention as A
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B, n, H, d = 64, 513, 16, 64
torch.manual_seed(0)
qkv = torch.randn(B, n, H, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
scale = 1 / math.sqrt(d)
out, lse = A._fwd(q, k, v, scale)
qq, kk, vv = (qkv[:2].float()[..., i * d:(i + 1) * d].permute(0, 2, 1, 3) for i in range(3))
sc = qq @ kk.transpose(-1, -2) * scale
ref = (torch.softmax(sc, -1) @ vv).permute(0, 2, 1, 3)
print("rel-L2 out", float((out[:2].float() - ref).norm() / ref.norm()), "last row", float((out[:2, -1].float() - ref[:, -1]).norm() / ref[:, -1].norm()),
      "max |lse diff|", float((lse[:2] - torch.logsumexp(sc, -1)).abs().max()))
ts = []
for r in range(rounds):
    for _ in range(5):
        A._fwd(q, k, v, scale)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        A._fwd(q, k, v, scale)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1000 / 50)
print(os.environ.get("NPCD_HIP_LIB", "default"), "us per launch:", " ".join(f"{t:.1f}" for t in ts))
