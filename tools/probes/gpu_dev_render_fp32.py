"""Dev probe: PointNeRF.render(mlp_dtype=torch.float32) per 128 x 128 view, heads as fp32 library GEMMs or (NPCD_STAGE1_X2_HEADS=1) split-operand GEMMs."""
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
extr = orr.look_at_pose(30, 20)[None, None].to(dev); intr = orr.srn_intrinsics()[None, None].to(dev)
c, f = coords.to(dev), feats.to(dev)
with torch.no_grad():
    for _ in range(5): net.render(c, f, extr, intr, 128, mlp_dtype=torch.float32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): out = net.render(c, f, extr, intr, 128, mlp_dtype=torch.float32)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    ref = net.render(c, f, extr, intr, 128)["channels"]
print(f"X2_HEADS={os.environ.get('NPCD_STAGE1_X2_HEADS')}: {dt * 1e3:.3f} ms per view = {128 * 128 / dt / 1e6:.1f} M rays/s; max |pixel - fp16 render| {float((out['channels'] - ref).abs().max()):.1e}")
