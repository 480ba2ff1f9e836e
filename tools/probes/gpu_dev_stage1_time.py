"""Dev probe: the stage-1 training step of bench.py (8 objects x 50 views x 112 rays) alone."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
print(bench.bench_stage1(torch.device("cuda", 0)))
print(bench.bench_stage1(torch.device("cuda", 0), mlp_dtype=torch.bfloat16))
