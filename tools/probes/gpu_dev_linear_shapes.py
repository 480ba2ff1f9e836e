"""Dev probe: forward Linear(256 -> 256) on ~6e5 rows in bf16 -- which formulation gets a sensible library kernel?"""
import torch, sys
import torch.nn.functional as F
M = int(sys.argv[1]) if len(sys.argv) > 1 else 612345
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for dt in (torch.bfloat16, torch.float32):
    x = torch.randn(M, 256, device="cuda", dtype=dt); W = torch.randn(256, 256, device="cuda", dtype=dt) * 0.05; b = torch.randn(256, device="cuda", dtype=dt)
    fl = 2 * M * 256 * 256
    res = {}
    res["F.linear"] = timeit(lambda: F.linear(x, W, b))
    res["mm + add"] = timeit(lambda: torch.mm(x, W.t()).add_(b))
    res["(W x^T)^T"] = timeit(lambda: (torch.mm(W, x.t()) + b[:, None]).t())
    for C in (16384, 65536, 131072):
        def chunked():
            out = torch.empty(M, 256, device="cuda", dtype=dt)
            for s in range(0, M, C):
                torch.addmm(b, x[s:s + C], W.t(), out=out[s:s + C])
            return out
        res[f"chunks of {C}"] = timeit(chunked)
    Mp = (M + 65535) // 65536 * 65536
    xp = torch.zeros(Mp, 256, device="cuda", dtype=dt); xp[:M] = x
    res[f"padded to {Mp}"] = timeit(lambda: F.linear(xp, W, b))
    print(dt, "M =", M, {k: f"{v:.0f} us ({fl / v / 1e6:.0f} TF/s)" for k, v in res.items()}, flush=True)
