"""Dev probe: GELU backward + bias-gradient column partials (colsum_kernel<true>) at cfg-D (T = 32832, N = 4096): time, effective HBM
rate, error against an fp64 reference.  NPCD_GELU_BWD_VARIANT selects the kernel variant (read once per process)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import elementwise as ew
T, N = int(sys.argv[1]) if len(sys.argv) > 1 else 32832, 4096
torch.manual_seed(0)
h = (torch.randn(T, N, device="cuda") * 1.5).bfloat16(); dg = torch.randn(T, N, device="cuda").bfloat16(); db = torch.empty(N, device="cuda")
for _ in range(3): dh = ew.gelu_bwd(dg, h, db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = int(os.environ.get("REPS", 30))
e0.record()
for _ in range(reps): dh = ew.gelu_bwd(dg, h, db)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
x = h[:2048].double()
ref = dg[:2048].double() * (0.5 * (1 + torch.erf(x * 0.7071067811865476)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327)
err = (dh[:2048].double() - ref).abs()
ulp = (err / ref.abs().clamp_min(1e-30))
print(f"variant {os.environ.get('NPCD_GELU_BWD_VARIANT', '1')}: {us:7.1f} us  {T * N * 6 / us / 1e6:5.2f} TB/s   max abs err {float(err.max()):.3e}  "
      f"rel-L2 {float((dh[:2048].double() - ref).norm() / ref.norm()):.3e}  share of elements off by more than one bf16 ulp {float((ulp > 2 ** -7).double().mean()):.2e}", flush=True)
