"""Dev probe: where the per-iteration spread of the stage-1 (PointNeRF) training step comes from -- per-step GPU time (HIP events) against
the step's shading points P and pairs Q (the rays are redrawn every step), at the bench configuration (fp32)."""
import sys, os
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import torch, numpy as np
from npcd.models import NPCD
from npcd.train import PointNeRFTrainer
from npcd.utils import synthetic as orr
import npcd.models.pointnerf.train_path as tp
import npcd.models.pointnerf.renderer as rd
dev = torch.device("cuda", 0)
B, T, N, F_, res = 8, 50, 512, 32, 128
torch.manual_seed(0)
net = NPCD(n_obj=B, coords_dim=3, feats_dim=F_, num_points=N, use_view_dir=False, width=64, layers=1, heads=1, pointnerf_only=True).to(dev)
coords, _ = orr.ellipsoid_cloud(N, F_, B, seed=0); net.pointnerf.set_all_coords(coords.to(dev))
extr = torch.stack([orr.look_at_pose(7.2 * i, 20 - 0.5 * i) for i in range(T)])[None].expand(B, -1, -1, -1).contiguous().to(dev)
intr = orr.srn_intrinsics()[None, None].expand(B, T, 3, 3).contiguous().to(dev)
sample = {"images": torch.rand(B, T, 3, res, res, device=dev), "intrinsics": intr, "extrinsics": extr, "obj_idx": torch.arange(B, device=dev)}
counts = []
orig = tp.render_train
def wrapped(*a, **k):
    out = orig(*a, **k); counts.append((out["num_shading_points"], out["num_pairs"])); return out
tp.render_train = wrapped
tr = PointNeRFTrainer(net, mlp_dtype=None)
for _ in range(3): tr.step(sample)
torch.cuda.synchronize(); counts.clear()
n = 30
marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
marks[0].record()
for i in range(n):
    tr.step(sample); marks[i + 1].record()
torch.cuda.synchronize()
ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(n)])
P = np.array([c[0] for c in counts], dtype=np.float64); Q = np.array([float(c[1]) for c in counts])
print("per-step ms  min/median/max %.1f / %.1f / %.1f" % (ms.min(), np.median(ms), ms.max()))
print("shading points P min/median/max %d / %d / %d   pairs Q %d / %d / %d" % (P.min(), np.median(P), P.max(), Q.min(), np.median(Q), Q.max()))
print("correlation(ms, P) = %.3f   correlation(ms, Q) = %.3f" % (np.corrcoef(ms, P)[0, 1], np.corrcoef(ms, Q)[0, 1]))
a, b = np.polyfit(Q, ms, 1); print("fit ms = %.2f + %.2f per 10^6 pairs; residual std %.2f ms" % (b, a * 1e6, np.std(ms - (a * Q + b))))
