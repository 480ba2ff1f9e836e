"""Dev probe: a Linear layer in the fp32 class as ONE bf16 library GEMM over the three cross products of split operands,
[xh | xl | xh] [Wh | Wh | Wl]^T with fp32 output, against the fp32 library GEMM -- time and error at the sampler's shapes (T = 8,208 rows)."""
import sys, os, time
import torch
torch.manual_seed(0)
dev = "cuda"
def split2(t):
    hi = t.to(torch.bfloat16)
    return hi, (t - hi.float()).to(torch.bfloat16)
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8208
for (K, N) in ((1024, 3072), (1024, 1024), (1024, 4096), (4096, 1024)):
    x = torch.randn(T, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    ref = (x.double() @ w.double().t() + b.double())
    y32 = torch.addmm(b, x, w.t())
    wh, wl = split2(w)
    W3 = torch.cat((wh, wh, wl), dim=1).contiguous()
    def x2():
        xh, xl = split2(x)
        X3 = torch.cat((xh, xl, xh), dim=1)
        return torch.mm(X3, W3.t(), out_dtype=torch.float32) + b
    def x2_gemm_only(X3=torch.cat((*split2(x)[:2], split2(x)[0]), dim=1)):
        return torch.mm(X3, W3.t(), out_dtype=torch.float32)
    yx = x2()
    rel = lambda a: float((a.double() - ref).norm() / ref.norm())
    t32 = timeit(lambda: torch.addmm(b, x, w.t()))
    tx = timeit(x2)
    tg = timeit(x2_gemm_only)
    ybf = (x.bfloat16() @ w.bfloat16().t()).float() + b
    print(f"T {T} K {K} N {N}: fp32 {t32:7.1f} us (rel {rel(y32):.1e})   x2 {tx:7.1f} us, its GEMM alone {tg:7.1f} us (rel {rel(yx):.1e})   bf16 rel {rel(ybf):.1e}", flush=True)
