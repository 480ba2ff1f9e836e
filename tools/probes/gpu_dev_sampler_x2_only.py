"""Dev probe: the fp32-class reverse step alone at batch 16 (for rocprofv3 kernel stats)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
from npcd.models.diffusion import DiffusionModel
CFG = bench.CFG
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = DiffusionModel(3, CFG["feats_dim"], CFG["num_points"], CFG["width"], CFG["layers"], CFG["heads"], True).to(dev).eval()
m.denoiser.backbone.fp32_class = True
B = 16
c = torch.randn(B, 3, CFG["num_points"], device=dev); f = torch.randn(B, CFG["feats_dim"], CFG["num_points"], device=dev)
dp = m.diffusion_process
with torch.no_grad():
    for n in (3, 10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        cc, ff = c, f
        for i in range(999, 999 - n, -1):
            t = torch.full((B,), i, device=dev, dtype=torch.long)
            cc, _, ff, _ = dp.p_sample(m.denoiser, cc, ff, t, None, None)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"{dt * 1e3:.2f} ms per reverse step")
