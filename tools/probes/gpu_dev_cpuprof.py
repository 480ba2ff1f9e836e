"""Dev probe: cProfile of the host side of one denoiser training step (per-GPU batch argv[1])."""
import sys, os, time, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
dev = torch.device("cuda", 0)
B = 8
tr = bench.build_trainer(dev, B)
coords, feats = bench.synthetic_batch(64, 0, 64 // B, dev)
for _ in range(3): tr.step(coords, feats)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): tr.step(coords, feats)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
