"""Dev probe: stage-1 step, enqueue (CPU) time vs wall time, and the top CPU functions of one step."""
import sys, os, time, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
from npcd.models import NPCD
from npcd.train import PointNeRFTrainer
from npcd.utils import synthetic as orr
dev = torch.device("cuda", 0)
B, T, N, F_, res = 8, 50, 512, 32, 128
torch.manual_seed(0)
net = NPCD(n_obj=B, coords_dim=3, feats_dim=F_, num_points=N, use_view_dir=False, width=64, layers=1, heads=1, pointnerf_only=True).to(dev)
coords, _ = orr.ellipsoid_cloud(N, F_, B, seed=0)
net.pointnerf.set_all_coords(coords.to(dev))
extr = torch.stack([orr.look_at_pose(7.2 * i, 20 - 0.5 * i) for i in range(T)])[None].expand(B, -1, -1, -1).contiguous().to(dev)
intr = orr.srn_intrinsics()[None, None].expand(B, T, 3, 3).contiguous().to(dev)
sample = {"images": torch.rand(B, T, 3, res, res, device=dev), "intrinsics": intr, "extrinsics": extr, "obj_idx": torch.arange(B, device=dev)}
for dt in (None, torch.bfloat16):
    tr = PointNeRFTrainer(net, mlp_dtype=dt)
    for _ in range(3): tr.step(sample)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5): tr.step(sample)
    t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print(f"mlp_dtype={dt}: enqueue {(t1 - t0) / 5 * 1e3:.1f} ms/step, wall {(t2 - t0) / 5 * 1e3:.1f} ms/step", flush=True)
pr = cProfile.Profile(); pr.enable(); tr.step(sample); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
