"""Dev probe: fp32-output weight-gradient GEMMs (not covered by TunableOp) under the two BLAS back ends torch can route to."""
import sys, torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32832
f32 = torch.float32
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, N, K, S in (("c_qkv", 3072, 1024, 4), ("attn.c_proj", 1024, 1024, 8), ("c_fc", 4096, 1024, 4), ("mlp.c_proj", 1024, 4096, 4)):
        S = max(1, min(S, T // 4096))
        while S > 1 and T % S: S //= 2
        dy = torch.randn(T, N, device="cuda").bfloat16(); x = torch.randn(T, K, device="cuda").bfloat16()
        out = torch.empty(N, K, device="cuda")
        a, b = dy.view(S, T // S, N), x.view(S, T // S, K)
        def f():
            if S == 1: torch.mm(dy.t(), x, out_dtype=f32, out=out)
            else: torch.sum(torch.bmm(a.transpose(1, 2), b, out_dtype=f32), dim=0, out=out)
        t = timeit(f)
        print(f"{lib:9s} T={T} {name:12s} S={S}: {t:7.1f} us  {2 * T * N * K / t / 1e6:5.0f} TF/s", flush=True)
