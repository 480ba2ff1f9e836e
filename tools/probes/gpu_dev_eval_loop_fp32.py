"""Dev probe: the generate -> render evaluation loop in the reference's numerics class (bench.bench_sample_and_render(fp32_class=True))."""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
print(json.dumps(bench.bench_sample_and_render(torch.device("cuda", 0), fp32_class=True), indent=1))
