"""Dev probe: wall time of a 128x128 render, one view per call (NPCD_RENDERS back-to-back calls; NPCD_S = depth samples per ray, default 128;
NPCD_ZERO_DATA=1 zeroes weights and features for the clock check)."""
import sys, os, time
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.utils import synthetic as orr
from npcd.models.pointnerf import PointNeRF
coords, feats = orr.ellipsoid_cloud(512, 32, 1, seed=0)
torch.manual_seed(0); model = PointNeRF(1, 32, 512, False).cuda().eval()      # field MLPs: PyTorch default init, as in bench.py; model.renderer.count_pairs = bool(int(os.environ.get("COUNT_PAIRS", "0")))
extr = orr.look_at_pose(30, 20)[None, None].cuda(); intr = orr.srn_intrinsics()[None, None].cuda()
c, f = coords.cuda(), feats.cuda()
if os.environ.get("NPCD_ZERO_DATA"):      # clock check: all-zero weights and features (same instruction stream, less power)
    with torch.no_grad():
        for p_ in model.field.parameters(): p_.zero_()
    f.zero_()
N_ = int(os.environ.get("NPCD_RENDERS", "10"))
model.renderer.depth_resolution = int(os.environ.get("NPCD_S", "128"))
with torch.no_grad():
    for _ in range(3): out = model.render(c, f, extr, intr, 128)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(N_): out = model.render(c, f, extr, intr, 128)
    torch.cuda.synchronize(); dt = (time.time() - t) / N_
print("128^2 view, S=%d: %.3f ms  %.2f Mrays/s  P=%d Q=%d" % (model.renderer.depth_resolution, dt*1e3, 16384/dt/1e6, out["num_shading_points"], out["num_pairs"]))
