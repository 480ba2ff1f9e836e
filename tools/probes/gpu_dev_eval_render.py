"""Dev probe: the evaluation loop's render calls (8 poses per call, a noise-shaped cloud as in bench.py's sample_and_render leg): wall per
call, shading points per view, and the same calls without the per-call host work of the loop (clamp / round / cat)."""
import sys, os, time
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.eval import load_test_poses
from npcd.models.pointnerf import PointNeRF
torch.manual_seed(0)
model = PointNeRF(1, 32, 512, False).cuda().eval()
poses, intr = load_test_poses("srncars"); poses, intr = poses.cuda().float(), intr.cuda().float()
coords = (torch.rand(1, 512, 3, device="cuda") - 0.5) * float(os.environ.get("SPAN", "1.0")); feats = torch.randn(1, 512, 32, device="cuda")
with torch.no_grad():
    for _ in range(2): out = model.render(coords, feats, poses[None, :8], intr[None, :8], resolution=128)
    torch.cuda.synchronize(); t0 = time.time(); P = 0
    for p0 in range(0, 248, 8):
        out = model.render(coords, feats, poses[None, p0:p0 + 8], intr[None, p0:p0 + 8], resolution=128)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 31
    print("8 views per call: %.2f ms per call = %.3f ms per view; shading points of the last call %d (%.0f per view)" % (dt * 1e3, dt * 1e3 / 8, int(out["num_shading_points"]), int(out["num_shading_points"]) / 8))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for p0 in range(0, 248, 8):
        out = model.render(coords, feats, poses[None, p0:p0 + 8], intr[None, p0:p0 + 8], resolution=128)
    e1.record(); torch.cuda.synchronize()
    print("GPU time between events: %.2f ms per call" % (e0.elapsed_time(e1) / 31))
