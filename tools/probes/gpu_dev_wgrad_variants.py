"""Dev probe: weight-gradient GEMM formulations (bf16 in, fp32 out) for the four Linear shapes of a block at T = 32832."""
import torch
import sys
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32832
dev = "cuda"
f32 = torch.float32
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, N, K in (("c_qkv", 3072, 1024), ("attn.c_proj", 1024, 1024), ("c_fc", 4096, 1024), ("mlp.c_proj", 1024, 4096)):
    dy = torch.randn(T, N, device=dev).bfloat16(); x = torch.randn(T, K, device=dev).bfloat16()
    fl = 2 * T * N * K
    res = []
    for S in (1, 2, 4, 8):
        if T % S: continue
        a = dy.view(S, T // S, N); b = x.view(S, T // S, K)
        outA = torch.empty(N, K, device=dev); outB = torch.empty(K, N, device=dev)
        def fA():
            if S == 1: torch.mm(dy.t(), x, out_dtype=f32, out=outA)
            else: torch.sum(torch.bmm(a.transpose(1, 2), b, out_dtype=f32), dim=0, out=outA)
        def fB():
            if S == 1: torch.mm(x.t(), dy, out_dtype=f32, out=outB)
            else: torch.sum(torch.bmm(b.transpose(1, 2), a, out_dtype=f32), dim=0, out=outB)
        tA, tB = timeit(fA), timeit(fB)
        res.append(f"S={S}: [N,K] {tA:6.1f} us ({fl / tA / 1e6:5.0f} TF/s)  [K,N] {tB:6.1f} us ({fl / tB / 1e6:5.0f} TF/s)")
    print(f"{name:12s} " + " | ".join(res), flush=True)
