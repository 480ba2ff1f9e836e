"""Dev probe: is the small per-GPU-batch step launch-bound?  wall vs enqueue time, run under rocprofv3 for the GPU-busy sum.
NPCD_B8_GRAPH=1: the step captured once into a HIP graph and replayed (TIMING ONLY: AdamW's bias-correction constants are host-side
kernel arguments, a replay repeats the captured step's) -- the GPU's own timeline with the host out of the way."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
tuned = os.path.join(R, "profiles", "tunableop_gfx950.csv")
if os.path.exists(tuned) and not os.environ.get("NPCD_NO_TUNED_GEMM"):
    import torch.cuda.tunable as tun
    tun.enable(True); tun.tuning_enable(False); tun.set_filename("/tmp/unused_tunable.csv"); tun.read_file(tuned)
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tr = bench.build_trainer(dev, B)
coords, feats = bench.synthetic_batch(64, 0, 64 // B, dev)
for _ in range(3): tr.step(coords, feats)
step = lambda: tr.step(coords, feats)
if os.environ.get("NPCD_B8_GRAPH"):
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): tr.step(coords, feats)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        tr.step(coords, feats)
    step = g.replay
    for _ in range(3): step()
# pure host cost of submitting a step: two steps into an EMPTY queue (no back-pressure from a full ring), three times
host = []
for _ in range(3):
    torch.cuda.synchronize(); h0 = time.time()
    step(); step()
    host.append((time.time() - h0) / 2 * 1e3)
    torch.cuda.synchronize()
print(f"B={B}: host-only submit {min(host):.2f} ms/step (two steps into an empty queue; runs {['%.2f' % h for h in host]})", flush=True)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(K): step()
t1 = time.time()
torch.cuda.synchronize(); t2 = time.time()
print(f"B={B}: enqueue {(t1-t0)/K*1e3:.2f} ms/step, wall {(t2-t0)/K*1e3:.2f} ms/step" + (" (graph replay)" if os.environ.get("NPCD_B8_GRAPH") else ""), flush=True)
