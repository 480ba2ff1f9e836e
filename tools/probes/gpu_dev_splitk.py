"""Dev probe: split-K variants of the weight-gradient GEMMs."""
import torch
T, W = 64 * 513, 1024
def tm(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for (N, K, name) in [(3 * W, W, "qkv"), (W, W, "proj"), (4 * W, W, "fc"), (W, 4 * W, "proj2")]:
    x = torch.randn(T, K, device="cuda", dtype=torch.bfloat16); dy = torch.randn(T, N, device="cuda", dtype=torch.bfloat16)
    base = tm(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))
    ref = torch.mm(dy.t(), x, out_dtype=torch.float32)
    for S in (2, 4, 8, 16):
        if T % S: continue
        def f():
            p = torch.bmm(dy.view(S, T // S, N).transpose(1, 2), x.view(S, T // S, K), out_dtype=torch.float32)
            return p.sum(0)
        try:
            t = tm(f)
            err = float((f() - ref).abs().max() / ref.abs().max())
            print(f"{name} wgrad base {base:.0f} us ; split-K S={S}: {t:.0f} us  ({2*T*N*K/t/1e6:.0f} TF) err {err:.1e}", flush=True)
        except Exception as e:
            print(name, S, "failed", str(e)[:80])
