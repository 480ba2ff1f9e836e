"""Dev probe (diagnostic build -DNPCD_TIMELINE=<block>): s_memtime stamps of one wave of the dK/dV pass."""
import sys, os, math, ctypes
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import attention as A, lib
B, n, H, d = 64, 513, 16, 64
qkv = torch.randn(B, n, H, 3 * d, device="cuda").bfloat16()
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
dout = torch.randn(B, n, H, d, device="cuda").bfloat16()
scale = 1 / math.sqrt(d)
g = torch.empty_like(qkv)
for _ in range(5):
    out, lse = A._fwd(q, k, v, scale)
    A._bwd(q, k, v, out, dout, lse, g[..., :d], g[..., d:2 * d], g[..., 2 * d:], scale)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 40)()
L = lib()
L.npcd_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", L.npcd_debug_read(ctypes.cast(buf, ctypes.c_void_p), 40))
t = list(buf)
t0 = t[0]
names = {22: "t3 start", 23: "t3 sub0 scores issued", 24: "t3 sub0 accum(prev)", 25: "t3 sub0 softmax", 26: "t3 mid wait+barrier", 27: "t3 prefetch issued", 28: "t3 sub1 scores", 29: "t3 sub1 accum", 30: "t3 sub1 softmax", 0: "entry", 1: "prologue issued (dma x2, kf/vf loads)", 2: "vmcnt(0) done", 3: "barrier done", 20: "loop end", 21: "stores drained"}
prev = t0
for i, x in enumerate(t):
    if x:
        print(f"{i:3d} {names.get(i, 'tile %d done' % (i - 4)):40s} +{x - prev:7d}  (={x - t0})")
        prev = x
