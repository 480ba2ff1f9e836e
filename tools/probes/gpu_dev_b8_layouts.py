"""Dev probe (round 6): the twelve GEMMs of a block at a rank's token count (T = 8 x 513 = 4,104), standalone, in the operand layouts
the backward could use: data gradient against the weight as stored (NN) or against a transposed 16-bit copy (TN, the forward's
layout); weight gradient as dy^T x (both operands token-major), on the own kernel, on token counts with and without the tail of 8
rows, and with pre-transposed activations.  NPCD_PROBE_TUNE=1 tunes the untuned forms on the fly (TunableOp).
usage: gpu_dev_b8_layouts.py [T ...]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
import torch.cuda.tunable as tun
tun.enable(True)
tun.read_file(os.path.join(R, "profiles", "tunableop_gfx950.csv"))
tune = bool(os.environ.get("NPCD_PROBE_TUNE"))
tun.tuning_enable(tune)
if tune:
    tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(20)
tun.set_filename(os.path.join(R, "gpurun_out", "r6_layout_tuned.csv"))
from npcd.hip import elementwise as ew
from npcd.hip import linear as hlin
dev = torch.device("cuda", 0)
bf, f32 = torch.bfloat16, torch.float32
W = 1024


def timeit(fn, n=40):
    for _ in range(6): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for T in [int(a) for a in sys.argv[1:]] or (4104,):
    shapes = {"c_qkv": (3 * W, W), "attn.c_proj": (W, W), "c_fc": (4 * W, W), "mlp.c_proj": (W, 4 * W)}
    print(f"--- T={T}", flush=True)
    tot = {}
    for name, (N, K) in shapes.items():
        dy = torch.randn(T, N, device=dev).to(bf); x = torch.randn(T, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.02).to(bf)
        wT = w.t().contiguous(); dyT = dy.t().contiguous(); xT = x.t().contiguous()
        bias = torch.zeros(N, device=dev, dtype=bf)
        y = torch.empty(T, N, device=dev, dtype=bf); dx = torch.empty(T, K, device=dev, dtype=bf); dw = torch.empty(N, K, device=dev, dtype=f32)
        r = {}
        r["fwd_TN"] = timeit(lambda: torch.addmm(bias, x, w.t(), out=y))
        r["dgrad_NN"] = timeit(lambda: torch.mm(dy, w, out=dx))
        r["dgrad_TN"] = timeit(lambda: torch.mm(dy, wT.t(), out=dx))
        if hlin.supported128(T, K, N):
            r["dgrad_lin128"] = timeit(lambda: hlin.linear128_fwd(dy, wT, None, dx))
        r["wgrad_lib"] = timeit(lambda: torch.mm(dy.t(), x, out_dtype=f32, out=dw))
        Tm = T - T % 256
        if Tm and Tm != T:
            r[f"wgrad_lib_T{Tm}"] = timeit(lambda: torch.mm(dy[:Tm].t(), x[:Tm], out_dtype=f32, out=dw))
        r["wgrad_own"] = timeit(lambda: ew.wgrad(dy, x, dw))
        r["wgrad_preT"] = timeit(lambda: torch.mm(dyT, xT.t(), out_dtype=f32, out=dw))
        dwb = torch.empty(N, K, device=dev, dtype=bf)
        r["wgrad_bf16out"] = timeit(lambda: torch.mm(dy.t(), x, out=dwb))
        r["transpose_w"] = timeit(lambda: hlin.transpose16(w, out=wT))
        fl = 2 * T * N * K
        print(f"{name:12s} " + "  ".join(f"{k} {v:6.1f}us({fl / v / 1e6:4.0f}TF)" if not k.startswith("transpose") else f"{k} {v:5.1f}us" for k, v in r.items()), flush=True)
        for k, v in r.items():
            tot[k] = tot.get(k, 0) + v
    print("block totals: " + "  ".join(f"{k} {v:6.1f}" for k, v in tot.items()), flush=True)
if tune and hasattr(tun, "write_file"):
    tun.write_file(os.path.join(R, "gpurun_out", "r6_layout_tuned.csv"))
