"""Dev probe: per-section timing of one 128x128 render (query / shading / march) with events."""
import sys, os, time
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.utils import synthetic as orr
from npcd.models.pointnerf import PointNeRF
from npcd.hip import render as hr
coords, feats = orr.ellipsoid_cloud(512, 32, 1, seed=0)
torch.manual_seed(0); model = PointNeRF(1, 32, 512, False).cuda().eval()      # field MLPs: PyTorch default init, as in bench.py
extr = orr.look_at_pose(30, 20)[None, None].cuda(); intr = orr.srn_intrinsics()[None, None].cuda()
c, f = coords.cuda(), feats.cuda()
def T(): torch.cuda.synchronize(); return time.time()
agg = model.field.aggregator; grid = agg.voxel_grid
with torch.no_grad():
    for it in range(3):
        t0 = T(); model._set_pointset(c); t1 = T()
        o, d, a, b = hr.ray_gen(extr[0], intr[0], 128); t2 = T()
        rays = (o.view(1, -1, 3), d.view(1, -1, 3), a.view(1, -1), b.view(1, -1))
        counter, ray_base, _, ray_bits, nb, pts = grid.query_compact(8, 2, 50, rays, 128, 204800, points=c); t3 = T()
        w = model.field.packed_weights(c.device); t4 = T()
        sigma, rgb = hr.shade_points(w, 32, nb, pts, c.reshape(-1, 3), f.reshape(-1, 32), n_points=counter[:1]); t5 = T()
        m_, d_, ch = hr.ray_march_compact(sigma, rgb, ray_bits, pts, ray_base, o.view(-1, 3), d.view(-1, 3), b.reshape(-1), 50, True); t6 = T()
        P, ov = counter.tolist(); t7 = T()
        print("setpts %.2f raygen %.2f query %.2f pack %.2f shade %.2f march %.2f tolist %.2f ms  P=%d ov=%d" % tuple([(y - x) * 1e3 for x, y in zip((t0, t1, t2, t3, t4, t5, t6), (t1, t2, t3, t4, t5, t6, t7))] + [P, ov]))
import cProfile, pstats
with torch.no_grad():
    for _ in range(3): model.render(c, f, extr, intr, 128)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(5): model.render(c, f, extr, intr, 128)
    torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
