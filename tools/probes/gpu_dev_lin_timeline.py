"""Dev probe: cycle stamps of one K-step of the own NT GEMM (csrc/gemm_nt.hip built with -DNPCD_LIN_TL=<global step>), waves 0 (x loader) and
4 (w loader) of workgroup 0, c_fc shape.  usage: NPCD_HIP_LIB=<timeline build> python3 tools/probes/gpu_dev_lin_timeline.py"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "neural-point-cloud-diffusion_amd")]
import torch
from npcd.hip import linear as hl, lib
T, N, K = 32768, int(os.environ.get("N", 4096)), int(os.environ.get("K", 1024))
torch.manual_seed(0)
x = torch.randn(T, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda").bfloat16()
for _ in range(5):
    hl.linear_fwd(x, w, b)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
L = lib()
L.npcd_lin_debug_read.restype = ctypes.c_int
L.npcd_lin_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.npcd_lin_debug_read(buf, 64) == 0
names = ["top", "wait0", "issue0", "mma0", "dma0", "wait1", "issue1", "mma1", "dma1", "wait2", "issue2", "mma2", "dma2", "wait3", "vmwait", "barrier",
         "dmaW", "issue3", "mma3"]
for wv, tag in ((0, "wave 0 (x loader)"), (1, "wave 4 (w loader)")):
    t = [buf[wv * 32 + i] for i in range(19)]
    print(tag, "step total", t[18] - t[0], "cycles")
    print("   " + "  ".join(f"{names[i]} +{t[i] - t[i - 1]}" for i in range(1, 19)))
print("offset wave4 - wave0 at top:", buf[32] - buf[0])
