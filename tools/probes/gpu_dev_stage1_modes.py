"""Dev probe: the stage-1 training step in its three numerics modes (fp32-class fused pair MLP / fp32 library GEMMs / bf16 opt-in)."""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
dev = torch.device("cuda", 0)
for name, dt in (("x2", None), ("library", "library"), ("bf16", torch.bfloat16), ("x2", None)):
    r = bench.bench_stage1(dev, mlp_dtype=dt)
    print(name, {k: r[k] for k in ("ms_per_step", "ms_per_step_min_median_max", "loss")}, flush=True)
