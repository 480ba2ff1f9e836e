"""Dev probe: the twelve GEMMs of one transformer block at per-GPU batch 8 (T = 4104 tokens) through the library, one by one:
forward (y = x W^T + b), data gradient (dx = dy W), weight gradient (dW = dy^T x, library and the own kernel).  Prints us and
TFLOP/s per call and the sum per block against the time at 1.25 PFLOP/s (what the same GEMMs reach at per-GPU batch 64)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
import torch.cuda.tunable as tun
if os.path.exists(os.path.join(R, "profiles", "tunableop_gfx950.csv")) and not os.environ.get("NPCD_NO_TUNABLE"):
    tun.enable(True); tun.tuning_enable(False); tun.read_file(os.path.join(R, "profiles", "tunableop_gfx950.csv"))
    tun.set_filename("/tmp/npcd_tunableop_unused.csv")
from npcd.hip import elementwise as ew
dev = torch.device("cuda", 0)
bf, f32 = torch.bfloat16, torch.float32
W = 1024


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in [int(a) for a in sys.argv[1:]] or (8,):
    T = B * 513
    shapes = {"c_qkv": (3 * W, W), "attn.c_proj": (W, W), "c_fc": (4 * W, W), "mlp.c_proj": (W, 4 * W)}
    print(f"--- B={B} T={T}")
    total, ideal = 0.0, 0.0
    for name, (N, K) in shapes.items():
        x = torch.randn(T, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.02).to(bf); b = torch.zeros(N, device=dev, dtype=bf)
        dy = torch.randn(T, N, device=dev).to(bf)
        dw = torch.empty(N, K, device=dev, dtype=f32)
        fl = 2.0 * T * N * K
        t_f = timeit(lambda: torch.addmm(b, x, w.t()))
        t_d = timeit(lambda: torch.mm(dy, w))
        t_w = timeit(lambda: torch.mm(dy.t(), x, out_dtype=f32, out=dw))
        t_o = timeit(lambda: ew.wgrad(dy, x, dw))
        print(f"  {name:12s} N={N:5d} K={K:5d}: fwd {t_f:6.1f} us ({fl / t_f / 1e6:5.0f} TF/s)  dgrad {t_d:6.1f} us ({fl / t_d / 1e6:5.0f})  "
              f"wgrad lib {t_w:6.1f} us ({fl / t_w / 1e6:5.0f})  own {t_o:6.1f} us ({fl / t_o / 1e6:5.0f})")
        total += t_f + t_d + min(t_w, t_o)
        ideal += 3 * fl / 1.25e9
    print(f"  block: {total:.0f} us in GEMMs (x 24 layers = {total * 24 / 1e3:.2f} ms/step); at 1.25 PF/s: {ideal:.0f} us ({ideal * 24 / 1e3:.2f} ms/step)")
