"""Dev probe: the GEMMs of one block's backward at small per-GPU batch (T = B x 513 tokens): weight-gradient split factors and
dgrad || wgrad on two streams.  usage: gpu_dev_b8_gemm.py [B ...]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
import torch.cuda.tunable as tun
tun.enable(True); tun.tuning_enable(False); tun.read_file(os.path.join(R, "profiles", "tunableop_gfx950.csv"))
tun.set_filename("/tmp/npcd_tunableop_unused.csv")
dev = torch.device("cuda", 0)
bf, f32 = torch.bfloat16, torch.float32
W = 1024


def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for B in [int(a) for a in sys.argv[1:]] or (8, 16, 32):
    T = B * 513
    shapes = {"c_qkv": (3 * W, W), "attn.c_proj": (W, W), "c_fc": (4 * W, W), "mlp.c_proj": (W, 4 * W)}
    print(f"--- B={B} T={T}")
    tot = {}
    for name, (N, K) in shapes.items():
        dy = torch.randn(T, N, device=dev).to(bf); x = torch.randn(T, K, device=dev).to(bf); w = torch.randn(N, K, device=dev).to(bf)
        out = torch.empty(N, K, device=dev, dtype=f32)
        res = {}
        for S in (1, 2, 4, 8):
            if T % S: continue
            if S == 1:
                fn = lambda: torch.mm(dy.t(), x, out_dtype=f32, out=out)
            else:
                def fn(S=S):
                    part = torch.bmm(dy.view(S, T // S, -1).transpose(1, 2), x.view(S, T // S, -1), out_dtype=f32)
                    torch.sum(part, dim=0, out=out)
            res[S] = timeit(fn)
        dx = torch.empty(T, K, device=dev, dtype=bf)
        t_d = timeit(lambda: torch.mm(dy, w, out=dx))
        y = torch.empty(T, N, device=dev, dtype=bf); bias = torch.zeros(N, device=dev, dtype=bf)
        t_f = timeit(lambda: torch.addmm(bias, x, w.t(), out=y))
        # dgrad || wgrad on two streams
        side = torch.cuda.Stream()
        bestS = min(res, key=res.get)
        def both():
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                if bestS == 1:
                    torch.mm(dy.t(), x, out_dtype=f32, out=out)
                else:
                    part = torch.bmm(dy.view(bestS, T // bestS, -1).transpose(1, 2), x.view(bestS, T // bestS, -1), out_dtype=f32)
                    torch.sum(part, dim=0, out=out)
            torch.mm(dy, w, out=dx)
            torch.cuda.current_stream().wait_stream(side)
        t_b = timeit(both)
        fl = 2 * T * N * K
        print(f"{name:12s} fwd {t_f:7.1f} us ({fl/t_f/1e6:6.0f} TF) dgrad {t_d:7.1f} ({fl/t_d/1e6:6.0f} TF) wgrad " +
              " ".join(f"S{S}:{v:7.1f}" for S, v in res.items()) + f" | dgrad||wgrad(S{bestS}) {t_b:7.1f} vs serial {t_d + res[bestS]:7.1f}")
