"""Dev probe: occupied 32-row blocks of the pair kernel's tiles (16 points x 8 slots) along the compact list of the bench view."""
import sys, os
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.utils import synthetic as orr
from npcd.models.pointnerf import PointNeRF
from npcd.hip import render as hr
coords, feats = orr.ellipsoid_cloud(512, 32, 1, seed=0)
torch.manual_seed(0); model = PointNeRF(1, 32, 512, False).cuda().eval()
extr = orr.look_at_pose(30, 20)[None, None].cuda(); intr = orr.srn_intrinsics()[None, None].cuda()
rd = model.renderer; agg = rd.field.aggregator; grid = agg.voxel_grid
kp = coords.cuda(); model._set_pointset(kp)
o, d, t0, t1 = hr.ray_gen(extr.flatten(0, 1), intr.flatten(0, 1), 128, rd.cube_scale); t0, t1 = rd.limits(t0, t1)
rays = (o.view(1, -1, 3), d.view(1, -1, 3), t0.view(1, -1), t1.view(1, -1))
counter, ray_base, _, ray_bits, nb, pts = grid.query_compact(agg.k, agg.r, agg.max_shading_pts, rays, rd.depth_resolution, 16384 * agg.max_shading_pts, points=kp)
P = int(counter[0]); cnt = (nb[:P] >= 0).sum(1)
T = (P + 15) // 16; pad = torch.zeros(T * 16, dtype=cnt.dtype, device=cnt.device); pad[:P] = cnt
rows = pad.view(T, 16).sum(1); blk = (rows + 31) // 32
print("P", P, "tiles", T, "pairs", int(cnt.sum()), "mean blocks %.2f" % blk.float().mean().item(), "hist", torch.bincount(blk, minlength=5).tolist())
dec = blk.float().view(-1)[: T // 10 * 10].view(10, -1).mean(1)
print("mean blocks per tenth of the list:", [round(x, 2) for x in dec.tolist()])
