"""Developer probe: hipBLASLt GEMM throughput for the denoiser shapes (via torch)."""
import torch, time, os
T, W = 64 * 513, 1024
dev = "cuda"
def bench(fn, flops, name, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"{name:34s} {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s", flush=True)
    return ms
tot = 0
for (N, K, tag) in [(3 * W, W, "qkv"), (W, W, "proj"), (4 * W, W, "fc"), (W, 4 * W, "proj2")]:
    x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16)
    fl = 2 * T * N * K
    tot += bench(lambda: torch.addmm(b, x, w.t()), fl, f"{tag} fwd  addmm(x, W^T)+b")
    tot += bench(lambda: torch.mm(dy, w), fl, f"{tag} dgrad mm(dy, W)")
    tot += bench(lambda: torch.mm(dy.t(), x), fl, f"{tag} wgrad mm(dy^T, x) bf16 out")
    try:
        o = torch.empty(N, K, device=dev, dtype=torch.float32)
        bench(lambda: torch.mm(dy.t(), x, out_dtype=torch.float32), fl, f"{tag} wgrad out_dtype=f32")
    except Exception as e:
        print("out_dtype fp32 unsupported:", type(e).__name__, str(e)[:100])
    # padded T (multiple of 256)
    Tp = (T + 255) // 256 * 256
    xp = torch.randn(Tp, K, device=dev, dtype=torch.bfloat16)
    bench(lambda: torch.addmm(b, xp, w.t()), 2 * Tp * N * K, f"{tag} fwd padded T={Tp}")
print(f"sum of 12 GEMMs per layer: {tot:.3f} ms -> x24 = {tot*24:.1f} ms")
# elementwise bandwidth sanity
a = torch.randn(T, 4 * W, device=dev, dtype=torch.bfloat16)
bench(lambda: torch.nn.functional.gelu(a), 0, "gelu fwd bf16 [T,4W]")
x32 = torch.randn(T, W, device=dev)
bench(lambda: torch.nn.functional.layer_norm(x32, (W,)), 0, "layer_norm fp32 [T,W]")
