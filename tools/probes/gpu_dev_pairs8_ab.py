"""Dev probe: the pair kernel's four-wave and eight-wave forms (NPCD_SHADE_PAIRS8, read per call) in ONE process: are the rendered
outputs the same bits, and the time of a 128 x 128 view in alternating rounds (the pair kernel is ~60 % of it).
usage: python3 tools/probes/gpu_dev_pairs8_ab.py [rounds] [feat_dim] [S]"""
import sys, os
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.utils import synthetic as orr
from npcd.models.pointnerf import PointNeRF
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
F = int(sys.argv[2]) if len(sys.argv) > 2 else 32
coords, feats = orr.ellipsoid_cloud(512, F, 1, seed=0)
torch.manual_seed(0); model = PointNeRF(1, F, 512, False).cuda().eval()
extr = orr.look_at_pose(30, 20)[None, None].cuda(); intr = orr.srn_intrinsics()[None, None].cuda()
c, f = coords.cuda(), feats.cuda()


VAR = os.environ.get("NPCD_AB_VAR", "NPCD_SHADE_PAIRS8")          # which switch to A/B (NPCD_SHADE_PAIRS8 / NPCD_SHADE_PAIRS16)


def render(mode):
    os.environ[VAR] = mode
    with torch.no_grad():
        return model.render(c, f, extr, intr, 128)


outs = {m: render(m) for m in ("0", "1")}
torch.cuda.synchronize()
for k in ("channels", "depth", "mask"):
    a, b = outs["0"][k], outs["1"][k]
    print(k, "equal" if torch.equal(a, b) else f"max diff {float((a - b).abs().max()):.3e}", "finite", bool(torch.isfinite(b).all()))
for r in range(rounds):
    for m in ("0", "1"):
        for _ in range(5): render(m)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): render(m)
        e1.record(); torch.cuda.synchronize()
        print(f"round {r} {VAR}={m}: {e0.elapsed_time(e1) / 40 * 1000:.1f} us per view", flush=True)
