"""Dev probe: denoiser step time at per-GPU batch 64 / 32 / 16 / 8 on one GPU (what each rank of a strong-scaling run computes)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(False); tun.read_file(os.path.join(R, "profiles", "tunableop_gfx950.csv"))     # as bench.py does
tun.set_filename("/tmp/npcd_tunableop_unused.csv")
dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or (8, 16, 32, 64):
    tr = bench.build_trainer(dev, B)
    coords, feats = bench.synthetic_batch(64, 0, 64 // B, dev)
    for _ in range(3): tr.step(coords, feats)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): tr.step(coords, feats)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print(f"per-GPU batch {B}: {dt*1e3:.2f} ms/step -> ideal {64//B}-GPU speedup vs B=64 single = (t64/{dt*1e3:.2f})", flush=True)
    del tr; torch.cuda.empty_cache()
