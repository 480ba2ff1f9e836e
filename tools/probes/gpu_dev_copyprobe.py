"""Dev probe: which host ops launch the small device-to-device copies in the step?"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
tr = bench.build_trainer(dev, 8)
coords, feats = bench.synthetic_batch(64, 0, 8, dev)
for _ in range(3): tr.step(coords, feats)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.step(coords, feats); torch.cuda.synchronize()
ev = prof.events()
# map correlation: for each device event named copyBuffer/Memcpy find the enclosing cpu op
import collections
cnt = collections.Counter()
cpu_ops = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
for e in cpu_ops:
    for k in e.kernels:
        if "copy" in k.name.lower() or "memcpy" in k.name.lower():
            cnt[(e.name, k.name[:40])] += 1
for k, v in cnt.most_common(20): print(v, k)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
