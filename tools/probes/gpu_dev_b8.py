"""Dev probe: is the small per-GPU-batch step launch-bound?  wall vs enqueue time, run under rocprofv3 for the GPU-busy sum."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tr = bench.build_trainer(dev, B)
coords, feats = bench.synthetic_batch(64, 0, 64 // B, dev)
for _ in range(3): tr.step(coords, feats)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(K): tr.step(coords, feats)
t1 = time.time()
torch.cuda.synchronize(); t2 = time.time()
print(f"B={B}: enqueue {(t1-t0)/K*1e3:.2f} ms/step, wall {(t2-t0)/K*1e3:.2f} ms/step", flush=True)
