"""Dev probe (diagnostic build -DNPCD_POINTS_TL=<workgroup>): s_memtime stamps (100 MHz counter -> printed in shader-clock-free
ticks) of the first pass of one workgroup of shade_points_kernel, wave 0."""
import sys, os, ctypes
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.utils import synthetic as orr
from npcd.models.pointnerf import PointNeRF
from npcd.hip import lib
coords, feats = orr.ellipsoid_cloud(512, 32, 1, seed=0)
torch.manual_seed(0); model = PointNeRF(1, 32, 512, False).cuda().eval()
extr = orr.look_at_pose(30, 20)[None, None].cuda(); intr = orr.srn_intrinsics()[None, None].cuda()
with torch.no_grad():
    model.render(coords.cuda(), feats.cuda(), extr, intr, 128)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 32)()
L = lib(); L.npcd_points_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.npcd_points_debug_read(ctypes.cast(buf, ctypes.c_void_p), 32)
names = {0: "pass start", 1: "tile in LDS", 2: "agg-4 mfma", 3: "barrier", 4: "store", 5: "barrier", 6: "density mfma + dot"}
for l in range(4):
    names[7 + 4 * l] = f"colour {l} mfma"; names[8 + 4 * l] = "barrier"; names[9 + 4 * l] = "store"; names[10 + 4 * l] = "barrier"
names[21] = "colour dot"; names[20] = "reduce + outputs"
t = list(buf); prev = t[0]
for i in list(range(20)) + [21, 20]:
    if i in names and t[i]:
        print(f"{i:2d} {names[i]:20s} +{t[i] - prev:7d}  (={t[i] - t[0]})"); prev = t[i]
