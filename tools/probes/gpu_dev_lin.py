"""Dev probe: the own NT GEMMs (csrc/gemm_nt.hip) against the library on the denoiser's shapes, interleaved timing.
usage: python3 tools/probes/gpu_dev_lin.py [T ...]   (default 32832 4104)"""
import os, sys, statistics
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "neural-point-cloud-diffusion_amd")]
import torch
from npcd.hip import linear as hl, elementwise as ew

Ts = [int(a) for a in sys.argv[1:]] or [32832, 4104]
# the library baseline with the solutions bench.py uses (profiles/tunableop_gfx950.csv, tuning disabled)
tuned = os.path.join(R, "profiles", "tunableop_gfx950.csv")
if os.path.exists(tuned) and not os.environ.get("NPCD_NO_TUNED_GEMM"):
    import torch.cuda.tunable as tun
    tun.enable(True)
    tun.tuning_enable(False)
    tun.set_filename("/tmp/npcd_tunableop_unused_probe.csv")
    print("tuned library solutions loaded:", bool(tun.read_file(tuned)))
W = 1024
dev = "cuda"
torch.manual_seed(0)


def timeit(fns, rounds=7, inner=5):
    res = {k: [] for k in fns}
    for _ in range(2):
        for f in fns.values():
            f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / inner * 1e3)
    return {k: (statistics.median(v), min(v)) for k, v in res.items()}


for T in Ts:
    print(f"==== T = {T}")
    for name, N, K in (("c_qkv", 3 * W, W), ("attn.c_proj", W, W), ("c_fc", 4 * W, W), ("mlp.c_proj", W, 4 * W)):
        x = torch.randn(T, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        out = torch.empty(T, N, device=dev, dtype=torch.bfloat16)
        wt = w.t()
        fl = 2.0 * T * N * K
        fns = {"lib": lambda: torch.addmm(b, x, wt, out=out), "own": lambda: hl.linear_fwd(x, w, b, out=out)}
        if name == "c_fc":
            fns["lib+gelu"] = lambda: ew.gelu_fwd(torch.addmm(b, x, wt, out=out))
            fns["own_gelu"] = lambda: hl.linear_gelu_fwd(x, w, b)
        if name == "mlp.c_proj":
            # its data gradient: dg [T, 4W] = dy [T, W] @ w [W, 4W]; NT form with the transposed weight [4W, W]
            dy = torch.randn(T, N, device=dev).bfloat16()
            h = torch.randn(T, K, device=dev).bfloat16()
            wT = w.t().contiguous()                # [K = 4W, N = W]
            dbias = torch.empty(K, device=dev)
            dgo = torch.empty(T, K, device=dev, dtype=torch.bfloat16)
            fns["lib_dgrad"] = lambda: torch.mm(dy, w, out=dgo)
            fns["lib_dgrad+gelu_bwd"] = lambda: ew.gelu_bwd(torch.mm(dy, w, out=dgo), h, dbias)
            fns["own_dgrad"] = lambda: hl.linear_fwd(dy, wT, None, out=dgo)
            fns["own_dgelu"] = lambda: hl.linear_dgelu_bwd(dy, wT, h)
        r = timeit(fns)
        y_own = hl.linear_fwd(x, w, b).float()
        y_lib = torch.addmm(b, x, wt).float()
        err = float((y_own - y_lib).norm() / y_lib.norm())
        print(f"{name:12s} N {N:5d} K {K:5d}  " + "  ".join(f"{k} {v[0]:7.1f} us ({fl / v[0] / 1e9:5.2f} PF/s; min {v[1]:.1f})" for k, v in r.items())
              + f"  | own vs lib rel {err:.1e}")
