"""Dev probe: the 128 x 128 NT product (npcd_linear128_fwd) against the tuned library at the token counts of one rank (T = 4,104 / 8,208 /
16,416) for the N = 1,024 products of a block: forward and data-gradient shapes.  usage: gpu_dev_lin128.py [T ...]"""
import sys, os
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import torch
import torch.cuda.tunable as tun
csv = os.path.join(R_, "profiles", "tunableop_gfx950.csv")
if os.path.exists(csv):
    tun.enable(True); tun.tuning_enable(False); tun.set_filename(csv); tun.read_file(csv)
from npcd.hip import linear as hl
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
Ts = [int(a) for a in sys.argv[1:]] or [4104, 8208, 16416]
for T in Ts:
    for name, N, K, bias in (("attn.c_proj fwd / dgrad", 1024, 1024, True), ("mlp.c_proj fwd", 1024, 4096, True), ("c_qkv dgrad", 1024, 3072, False),
                             ("c_fc dgrad", 1024, 4096, False), ("c_qkv fwd", 3072, 1024, True), ("c_fc fwd", 4096, 1024, True)):
        x = torch.randn(T, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
        y = torch.empty(T, N, device="cuda", dtype=torch.bfloat16); wt = w.t()
        lib_us = timeit((lambda: torch.addmm(b, x, wt, out=y)) if bias else (lambda: torch.mm(x, wt, out=y)))
        own_us = timeit(lambda: hl.linear128_fwd(x, w, b if bias else None, out=y))
        gf = 2 * T * N * K / 1e9
        print(f"T {T:6d}  {name:24s} N {N:5d} K {K:5d}: library {lib_us:7.1f} us ({gf / lib_us * 1e-3:5.2f} PF/s)   own 128 {own_us:7.1f} us ({gf / own_us * 1e-3:5.2f} PF/s)   x{lib_us / own_us:.2f}")
