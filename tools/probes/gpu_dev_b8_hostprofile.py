"""Dev probe: where the HOST spends a rank's step at per-GPU batch 8 (cProfile over 20 steps, sorted by own time)."""
import sys, os, time, cProfile, pstats, io
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = 20
tr = bench.build_trainer(dev, B)
coords, feats = bench.synthetic_batch(64, 0, 64 // B, dev)
for _ in range(3): tr.step(coords, feats)
torch.cuda.synchronize()
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):      # the backward in THIS thread, where the profiler sees it
    for _ in range(2): tr.step(coords, feats)
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(K): tr.step(coords, feats)
    pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(60)
    print(s.getvalue()[:14000])
