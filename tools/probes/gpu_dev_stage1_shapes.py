"""Dev probe: which aten matrix products of the stage-1 training step (bf16 form) run which library kernel, with their shapes
(torch.profiler, record_shapes) - to see which calls are left on slow library tiles."""
import sys, os, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
bench.bench_stage1(dev, mlp_dtype=torch.bfloat16)            # warm
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    bench.bench_stage1(dev, mlp_dtype=torch.bfloat16)
torch.cuda.synchronize()
rows = collections.defaultdict(lambda: [0, 0.0])
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm", "aten::linear", "aten::matmul", "aten::sum", "aten::copy_", "aten::_to_copy", "aten::add", "aten::add_", "aten::index", "aten::fill_", "aten::zero_"):
        k = (e.key, str(e.input_shapes)[:110])
        rows[k][0] += e.count; rows[k][1] += getattr(e, "self_device_time_total", 0.0)
for (k, sh), (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:14s} {c / 23:6.1f}/step {t / 23:9.1f} us/step  {sh}")
