"""Dev probe: the stage-1 training step of bench.py in the fp32-class mode alone (for rocprofv3 kernel stats)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
r = bench.bench_stage1(torch.device("cuda", 0))
print({k: r[k] for k in ("ms_per_step", "ms_per_step_min_median_max")})
