"""Developer probe: time one denoiser training step (cfg-D) with the simple module path."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.models.diffusion import DiffusionModel
B, N, F_ = int(os.environ.get("B", 64)), 512, 128
m = DiffusionModel(3, F_, N, 1024, 24, 16, True).cuda().train()
torch.nn.init.normal_(m.denoiser.output_proj.weight, std=0.02)
opt = torch.optim.AdamW(m.parameters(), lr=7e-5, weight_decay=0.01, fused=True)
coords = torch.randn(B, 3, N, device="cuda"); feats = torch.rand(B, F_, N, device="cuda") * 2 - 1
def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _, _ = m.compute_loss(coords, feats)
    loss.backward(); opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): l = step()
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f"B={B} step {dt*1e3:.1f} ms  loss {float(l):.4f}  {64.5*B/64/dt/1e3:.1f} TFLOP/s-equivalent  mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
