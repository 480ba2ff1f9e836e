"""Dev probe: time + effective HBM rate of the elementwise kernels at cfg-D (T = 32832, W = 1024)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import elementwise as ew
T, W = int(sys.argv[1]) if len(sys.argv) > 1 else 32832, 1024
dev = "cuda"
x = torch.randn(T, W, device=dev); delta = torch.randn(T, W, device=dev).bfloat16()
g = torch.randn(W, device=dev); b = torch.randn(W, device=dev)
dy = torch.randn(T, W, device=dev).bfloat16(); dres = torch.randn(T, W, device=dev)
dg, db, dc = (torch.empty(W, device=dev) for _ in range(3))
h = torch.randn(T, 4 * W, device=dev).bfloat16(); dgl = torch.randn(T, 4 * W, device=dev).bfloat16(); dbias = torch.empty(4 * W, device=dev)
def timeit(fn, nbytes, name, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"{name:14s} {us:8.1f} us   {nbytes / us / 1e6:6.2f} TB/s", flush=True)
x1, y, mean, rstd = ew.add_ln_fwd(x, delta, g, b)
timeit(lambda: ew.add_ln_fwd(x, delta, g, b), T * W * (4 + 2 + 4 + 2), "add_ln_fwd")
timeit(lambda: ew.ln_bwd(dy, x1, mean, rstd, g, dres, dg, db, dc), T * W * (2 + 4 + 4 + 4 + 2), "ln_bwd")
timeit(lambda: ew.gelu_fwd(h), T * 4 * W * 4, "gelu_fwd")
timeit(lambda: ew.gelu_bwd(dgl, h, dbias), T * 4 * W * 6, "gelu_bwd")
if not os.environ.get("NPCD_EW_ONLY_GELU_COLSUM"):      # (the traffic passes: rocprofv3 cannot tell the two colsum_kernel instantiations apart)
    timeit(lambda: ew.colsum_bf16(h, dbias), T * 4 * W * 2, "colsum_bf16")
