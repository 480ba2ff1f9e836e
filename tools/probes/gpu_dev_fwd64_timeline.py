"""Dev probe (diagnostic build -DNPCD_TIMELINE64=<block>): s_memtime stamps of wave 0 of one workgroup of the 64-row forward.
usage: NPCD_HIP_LIB=.../libnpcd_hip_tl64.so python3 tools/probes/gpu_dev_fwd64_timeline.py [n] [B]"""
import sys, os, math, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import attention as A, lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, d = 16, 64
qkv = torch.randn(B, n, H, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
for _ in range(20):
    out, lse = A._fwd(q, k, v, 1 / math.sqrt(d))
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 40)()
L = lib()
L.npcd_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", L.npcd_debug_read(ctypes.cast(buf, ctypes.c_void_p), 40))
t = list(buf)
names = {0: "entry", 1: "prologue loads + DMA issued", 2: "q rows arrived", 3: "seed / vmcnt(0) / barrier done", 20: "loop + flush done", 21: "half sums + barrier",
         22: "stores issued", 23: "stores drained"}
prev = t[0]
for i, x in enumerate(t):
    if x:
        print(f"{i:3d} {names.get(i, 'tile %d done' % (i - 4)):34s} +{x - prev:7d}  (={x - t[0]})")
        prev = x
