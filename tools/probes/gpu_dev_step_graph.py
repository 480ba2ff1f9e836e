"""Dev probe: what a HIP graph buys the denoiser training step.  The whole step (loss, fused forward / backward, AdamW + EMA) is
captured once into a torch.cuda.CUDAGraph and replayed; eager and replay are timed interleaved at per-GPU batch argv[1:].
TIMING ONLY: AdamW's bias-correction constants are kernel arguments computed on the host from the step count, so a replay repeats
the captured step's constants -- a production version would keep them in device memory."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
sys.argv = sys.argv[:1] + [a for a in sys.argv[1:]]
import bench
tuned = os.path.join(R, "profiles", "tunableop_gfx950.csv")
if os.path.exists(tuned):
    import torch.cuda.tunable as tun
    tun.enable(True); tun.tuning_enable(False); tun.set_filename("/tmp/unused_tunable.csv"); tun.read_file(tuned)
dev = torch.device("cuda", 0)
trainer = bench.build_trainer(dev, 64)
coords, feats = bench.synthetic_batch(64, 0, 1, dev)
for b in [int(a) for a in sys.argv[1:]] or [8, 64]:
    c, f = coords[:b].clone(), feats[:b].clone()
    for _ in range(4):
        trainer.step(c, f)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            trainer.step(c, f)
    torch.cuda.current_stream().wait_stream(s)
    try:
        with torch.cuda.graph(g):
            loss, _ = trainer.step(c, f)
    except Exception as e:      # noqa: BLE001
        print(f"batch {b}: capture failed: {type(e).__name__}: {str(e)[:300]}")
        continue

    def timed(fn, n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    n = 10
    res = {"eager": [], "graph": []}
    for _ in range(3):
        res["eager"].append(timed(lambda: trainer.step(c, f), n))
        res["graph"].append(timed(g.replay, n))
    print(f"per-GPU batch {b}: eager {min(res['eager']):.2f} ms/step (runs {['%.2f' % x for x in res['eager']]}), graph replay {min(res['graph']):.2f} ms/step "
          f"(runs {['%.2f' % x for x in res['graph']]}); loss after replays {float(loss):.4f}", flush=True)
