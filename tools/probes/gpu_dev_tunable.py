"""Developer probe: PyTorch TunableOp (picks the best rocBLAS/hipBLASLt solution per GEMM shape)."""
import os, sys, torch, time
import torch.cuda.tunable as tun
T, W = 64 * 513, 1024
def run(tag):
    tot = 0
    for (N, K, name) in [(3 * W, W, "qkv"), (W, W, "proj"), (4 * W, W, "fc"), (W, 4 * W, "proj2")]:
        x = torch.randn(T, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.02
        b = torch.zeros(N, device="cuda", dtype=torch.bfloat16); dy = torch.randn(T, N, device="cuda", dtype=torch.bfloat16)
        for nm, fn in (("fwd", lambda: torch.addmm(b, x, w.t())), ("dgrad", lambda: torch.mm(dy, w)),
                       ("wgrad", lambda: torch.mm(dy.t(), x, out_dtype=torch.float32))):
            for _ in range(3): fn()
            torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize(); ms = e0.elapsed_time(e1) / 10; tot += ms
            print(f"{tag} {name:6s}{nm:6s} {ms*1e3:7.1f} us {2*T*N*K/ms/1e9:7.1f} TF", flush=True)
    print(f"{tag} total/layer {tot:.3f} ms -> x24 {tot*24:.1f} ms", flush=True)
run("default")
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(20)
tun.set_filename(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gpurun_out", "tunableop_results.csv"))
t0 = time.time(); run("tuning "); print("tuning took", time.time() - t0)
tun.tuning_enable(False); run("tuned  ")
tun.write_file()
