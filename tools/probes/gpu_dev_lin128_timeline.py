"""Dev probe (diagnostic build -DNPCD_L128_TL=<even K-step>): cycle stamps of one step pair of workgroup 0, waves 0 and 4, of the 128 x 128 product."""
import sys, os, ctypes
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import linear as hl, lib
T, N, K = 4104, 1024, 4096
x = torch.randn(T, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
for _ in range(3): hl.linear128_fwd(x, w, None)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 24)(); L = lib(); L.npcd_lin128_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.npcd_lin128_debug_read(ctypes.cast(buf, ctypes.c_void_p), 24)
names = ["start", "loads issued", "frag reads issued", "mfma issued", "stage written", "barrier", "loads issued", "frag reads issued", "mfma issued", "stage written", "barrier"]
for wv in range(2):
    t = list(buf)[12 * wv: 12 * wv + 11]
    print(f"wave {4 * wv}: " + "  ".join(f"{names[i]} +{t[i] - t[i - 1]}" for i in range(1, 11)) + f"   = {t[10] - t[0]} clocks for two K-steps")
