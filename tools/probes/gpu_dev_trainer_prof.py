"""Developer probe: kernel breakdown of the fused training step."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
import bench
dev = torch.device("cuda", 0)
B_ = int(os.environ.get("B", 64))
tr = bench.build_trainer(dev, B_)
coords, feats = bench.synthetic_batch(64, 0, 64 // B_, dev)
for _ in range(3): tr.step(coords, feats)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): tr.step(coords, feats)
torch.cuda.synchronize(); print("step ms", (time.time() - t0) / 5 * 1e3)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    tr.step(coords, feats); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
