"""Dev probe: PointNeRF.render(mlp_dtype=torch.float32) with 8 poses per call (the evaluation loop's render_batch_size), for rocprofv3 kernel stats."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.models.pointnerf import PointNeRF
from npcd.utils import synthetic as orr
dev = torch.device("cuda", 0)
coords, feats = orr.ellipsoid_cloud(512, 32, 1, seed=0)
torch.manual_seed(0)
net = PointNeRF(1, 32, 512, False).to(dev).eval()
poses = torch.stack([orr.look_at_pose(30 + 45 * i, 20 - 5 * i) for i in range(8)])[None].to(dev)
intr8 = orr.srn_intrinsics()[None, None].expand(1, 8, 3, 3).contiguous().to(dev)
c, f = coords.to(dev), feats.to(dev)
with torch.no_grad():
    for _ in range(3): net.render(c, f, poses, intr8, 128, mlp_dtype=torch.float32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = net.render(c, f, poses, intr8, 128, mlp_dtype=torch.float32)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"8 poses per call: {dt * 1e3:.2f} ms per call = {dt / 8 * 1e3:.3f} ms per view, {int(out['num_shading_points'])} shading points")
