"""Dev probe: one DDPM reverse step of the benchmark denoiser at batch 16 in its numerics modes (bench.bench_sampler)."""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
r = bench.bench_sampler(torch.device("cuda", 0))
print(json.dumps(r, indent=1))
