"""Dev probe: hipBLASLt time of the step's forward / dgrad GEMM shapes at M = 32832 (64 x 513 tokens) vs M = 32768 (MS=33024,32768 for another list; TUNE=1 lets TunableOp pick the solution)."""
import sys, os
import torch
if os.environ.get("TUNE"):
    import torch.cuda.tunable as tun
    tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(20)
    tun.set_filename("/tmp/gemm_m_tune.csv")
dev = "cuda"
shapes = [("qkv fwd", 1024, 3072), ("proj fwd", 1024, 1024), ("fc fwd", 1024, 4096), ("proj2 fwd", 4096, 1024)]
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
MS = [int(m) for m in os.environ["MS"].split(",")] if os.environ.get("MS") else ((32832, 32768) if os.environ.get("TUNE") else (32832, 32768, 64))
for M in MS:
    tot = 0.0
    for name, K, N in shapes:
        x = torch.randn(M, K, device=dev).bfloat16(); w = torch.randn(N, K, device=dev).bfloat16(); b = torch.randn(N, device=dev).bfloat16()
        dy = torch.randn(M, N, device=dev).bfloat16()
        t_f = timeit(lambda: torch.addmm(b, x, w.t()))
        t_d = timeit(lambda: torch.mm(dy, w))
        fl = 2 * M * K * N
        print(f"M={M:6d} {name:10s} fwd {t_f:7.1f} us ({fl / t_f / 1e6:6.0f} TF/s)   dgrad {t_d:7.1f} us ({fl / t_d / 1e6:6.0f} TF/s)", flush=True)
        tot += t_f + t_d
    print(f"M={M}: sum {tot:.1f} us per layer", flush=True)
