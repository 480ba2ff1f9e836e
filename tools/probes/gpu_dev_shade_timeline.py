"""Dev probe (diagnostic build -DNPCD_SHADE_TL=<block>): s_memtime stamps of one tile of one workgroup of shade_pairs."""
import sys, os, ctypes
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.utils import synthetic as orr
from npcd.models.pointnerf import PointNeRF
from npcd.hip import lib
coords, feats = orr.ellipsoid_cloud(512, 32, 1, seed=0)
torch.manual_seed(0); model = PointNeRF(1, 32, 512, False).cuda().eval()      # field MLPs: PyTorch default init, as in bench.py
extr = orr.look_at_pose(30, 20)[None, None].cuda(); intr = orr.srn_intrinsics()[None, None].cuda()
with torch.no_grad():
    for _ in range(5): model.render(coords.cuda(), feats.cuda(), extr, intr, 128)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 32)()
L = lib(); L.npcd_shade_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.npcd_shade_debug_read(ctypes.cast(buf, ctypes.c_void_p), 32)
names = ["tile start", "prologue done", "barrier", "L0 mfma", "barrier", "L0 store", "barrier", "L1 mfma", "barrier", "L1 store", "barrier", "L2,L3 done", "aggregation", "barrier"]
t = list(buf); n4 = max(t[14], 1)
print(f"workgroup: {t[15]} tiles ({t[14]} with 4 row blocks, mean {t[17] / max(t[15], 1):.2f} blocks), {t[16]} clocks = {t[16] / max(t[15], 1):.0f} per tile")
print("mean clocks per interval over the 4-block tiles:  " + "  ".join(f"{names[i]} {t[i] / n4:.0f}" for i in range(1, 14)) + f"   = {sum(t[1:14]) / n4:.0f}")
if hasattr(L, "npcd_shade_span_read"):
    import numpy as np
    sp = (ctypes.c_longlong * 2048)(); L.npcd_shade_span_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.npcd_shade_span_read(ctypes.cast(sp, ctypes.c_void_p), 2048)
    sp = np.array(list(sp), dtype=np.int64).reshape(512, 4); t0 = sp[:, 0].min()
    hw, xcc = (sp[:, 2] >> 16) & 0xffffffff, (sp[:, 2] >> 48) & 15
    sp[:, 2] &= 0xffff
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    where = xcc * 10000 + se * 1000 + sh * 100 + cu                  # one number per physical CU
    print("tiles by XCD:", {int(x): int(sp[xcc == x, 2].sum()) for x in np.unique(xcc)}, " workgroups by XCD:", {int(x): int((xcc == x).sum()) for x in np.unique(xcc)})
    per_cu = {}
    for w, tl in zip(where, sp[:, 2]):
        per_cu.setdefault(int(w), []).append(int(tl))
    from collections import Counter
    print("workgroups per CU:", dict(Counter(len(v) for v in per_cu.values())), " CUs used:", len(per_cu))
    print("tiles per CU (sum of its workgroups) min/mean/max:", min(sum(v) for v in per_cu.values()), np.mean([sum(v) for v in per_cu.values()]), max(sum(v) for v in per_cu.values()))
    for k in (1, 2, 3):
        vs = [sum(v) for v in per_cu.values() if len(v) == k]
        if vs: print(f"  CUs with {k} workgroup(s): {len(vs)}, tiles per CU mean {np.mean(vs):.2f}, per workgroup {np.mean(vs) / k:.2f}")
    b, e = (sp[:, 0] - t0) / 100.0, (sp[:, 1] - t0) / 100.0          # microseconds (100-MHz counter)
    print(f"workgroup spans (us): begin min/median/max {b.min():.1f}/{np.median(b):.1f}/{b.max():.1f}   end min/median/max {e.min():.1f}/{np.median(e):.1f}/{e.max():.1f}"
          f"   tiles min/mean/max {sp[:, 2].min()}/{sp[:, 2].mean():.2f}/{sp[:, 2].max()}   clock {np.median(sp[:, 3] / np.maximum(sp[:, 1] - sp[:, 0], 1)) / 10:.3f} GHz")
if hasattr(L, "npcd_shade_rows_debug_read") and not os.environ.get("NPCD_SHADE_TILES"):
    L.npcd_shade_rows_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.npcd_shade_rows_debug_read(ctypes.cast(buf, ctypes.c_void_p), 16)
    names = ["tile start", "window tables", "layer-0 operand", "layer 0", "layer 1", "layer 2", "layer 3", "last epilogue", "aggregation"]
    t = list(buf); prev = t[0]
    print("rows kernel:")
    for i, x in enumerate(t[:9]):
        print(f"{i:2d} {names[i]:16s} +{x - prev:7d}  (={x - t[0]})"); prev = x
    print(f"   aggregation: weights operand +{t[9] - t[7]}, products + staging +{t[10] - t[9]}, row stores +{t[8] - t[10]}")
