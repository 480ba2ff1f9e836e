"""Dev probe: the fused AdamW + EMA pass on the benchmark's 310.8 M parameters, alone (2.27 ms = 5.75 TB/s at 42 bytes per parameter with the
gradient zeroed in the pass).  Round 5 swept the grid (1,024 ... 32,768 blocks) and non-temporal loads / stores of the moments and the EMA
with two temporary switches (NPCD_ADAMW_GRID / NPCD_ADAMW_NT, not kept): 2.22-2.63 ms, the shipped 8,192 blocks within 2 % of the best."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import elementwise as ew
n = 310_779_940 // 4 * 4
p, g, m, v, e = (torch.randn(n, device="cuda") * 0.01 for _ in range(5))
v.abs_()
sh = torch.empty(n, dtype=torch.bfloat16, device="cuda")
other = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
ts = []
for r in range(4):
    for _ in range(3):
        ew.adamw_ema(p, g, m, v, e, sh, 7e-5, 0.9, 0.999, 1e-8, 0.01, 5, 0.9999, True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ew.adamw_ema(p, g, m, v, e, sh, 7e-5, 0.9, 0.999, 1e-8, 0.01, 5, 0.9999, True)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
print(f"grid {os.environ.get('NPCD_ADAMW_GRID', '8192')} nt {os.environ.get('NPCD_ADAMW_NT', '0')}: " + " ".join(f"{t:.3f}" for t in ts) + f" ms  ({n * 42 / min(ts) / 1e9:.2f} TB/s at 42 B per parameter)")
