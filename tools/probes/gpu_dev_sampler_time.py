"""Dev probe: time of one DDPM reverse step (denoiser forward + posterior update) at cfg-D model size, batch 16."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.models.diffusion import DiffusionModel
torch.manual_seed(0)
m = DiffusionModel(3, 128, 512, 1024, 24, 16, True).cuda().eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
c = torch.randn(B, 3, 512, device="cuda"); f = torch.randn(B, 128, 512, device="cuda")
dp = m.diffusion_process
def run(n, ctx):
    cc, ff = c, f
    with torch.no_grad(), ctx:
        for i in range(999, 999 - n, -1):
            t = torch.full((B,), i, device="cuda", dtype=torch.long)
            cc, _, ff, _ = dp.p_sample(m.denoiser, cc, ff, t, None, None)
    return cc
import contextlib
for name, ctx in (("fp32", contextlib.nullcontext()), ("bf16 autocast", torch.autocast("cuda", dtype=torch.bfloat16))):
    run(3, ctx); torch.cuda.synchronize(); t0 = time.time(); out = run(10, ctx); torch.cuda.synchronize()
    dt = (time.time() - t0) / 10
    print(f"{name:14s} B={B}: {dt * 1e3:7.2f} ms per reverse step -> {1000 * dt / B:6.2f} s per generated cloud; finite={bool(torch.isfinite(out).all())}", flush=True)

# fused posterior update + HIP graph
import types
def loop(n, dtype, graph):
    dp.num_timesteps = n
    ctx = torch.autocast("cuda", dtype=dtype) if dtype else contextlib.nullcontext()
    with torch.no_grad(), ctx:
        return dp.p_sample_loop(m.denoiser, c, f, (-3.0, 3.0), (-1.0, 1.0), use_graph=graph)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for name, dtype, graph in (("bf16 fused", torch.bfloat16, False), ("bf16 fused+graph", torch.bfloat16, True)):
    loop(4, dtype, graph); torch.cuda.synchronize(); t0 = time.time(); out = loop(N, dtype, graph)[0]; torch.cuda.synchronize()
    dt = (time.time() - t0) / N
    print(f"{name:18s} B={B}: {dt * 1e3:7.2f} ms per reverse step (incl. capture amortised over {N} steps); finite={bool(torch.isfinite(out).all())}", flush=True)
