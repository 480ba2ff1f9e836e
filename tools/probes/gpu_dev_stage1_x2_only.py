"""Dev probe: the stage-1 training step in its default (fp32-class) mode only -- run in alternation with NPCD_STAGE1_LIBRARY_HEADS=1."""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
r = bench.bench_stage1(torch.device("cuda", 0), mlp_dtype=None)
print("heads:", "library" if os.environ.get("NPCD_STAGE1_LIBRARY_HEADS") else "fused", {k: r[k] for k in ("ms_per_step", "ms_per_step_min_median_max", "loss")}, flush=True)
