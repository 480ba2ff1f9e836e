"""Dev probe: the own weight-gradient kernel alone on one shape (for rocprofv3 --pmc / diagnostic builds).
usage: python3 tools/probes/gpu_dev_wgrad_one.py [T] [N] [K] [reps]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import elementwise as ew
T, N, K, reps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 32832), (2, 4096), (3, 1024), (4, 10)))
dy = torch.randn(T, N, device="cuda").bfloat16(); x = torch.randn(T, K, device="cuda").bfloat16()
out = torch.empty(N, K, device="cuda")
for _ in range(3): ew.wgrad(dy, x, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): ew.wgrad(dy, x, out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"T={T} N={N} K={K}: {us:.1f} us  {2 * T * N * K / us / 1e6:.0f} TF/s")
