import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/neural-point-cloud-diffusion_amd"); sys.path.insert(0, "/root/repo/tests")
import torch
from oracle import renderer as orr
from test_gpu_render import _overflow_case, _model
for which in ("pairs", "heads", "last_heads"):
    p, coords, feats, nb, pts = _overflow_case(which)
    m = _model(32, 256, p)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    sig, rgb = m.field.shade(nb.cuda(), pts.cuda(), coords.cuda(), feats.cuda(), status=status)
    torch.cuda.synchronize()
    print(which, int(status), bool(torch.isfinite(sig).all()), bool(torch.isfinite(rgb).all()), float(sig.abs().max()), rgb[:2].tolist())
    ref_s, ref_r, _ = orr.shade_points(p, nb.long(), pts, coords, feats)
    print("  oracle max sigma", float(ref_s.abs().max()))
