"""Dev probe: training step at BASELINE configs[4] shape (2048 points x 256-d latents, 8-layer denoiser, width 1024 / 16 heads,
per-GPU batch 32 of the 8-GPU batch 256); NPCD_ATTN_FP8=1 runs the attention forward on the fp8 kernel."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.models.diffusion import DiffusionModel
from npcd.train import DiffusionTrainer
from npcd.hip import attention as A
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = DiffusionModel(3, 256, 2048, 1024, 8, 16, True).cuda()
tr = DiffusionTrainer(m)
c, f = torch.randn(B, 3, 2048, device="cuda"), torch.randn(B, 256, 2048, device="cuda")
for _ in range(3):
    tr.step(c, f)
torch.cuda.synchronize()
A.KERNEL_EVENTS = {t: [] for t in A.KERNEL_TAGS}
t0 = time.time()
n = 8
for _ in range(n):
    loss, _ = tr.step(c, f)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
T, W, L, H, d, s = B * 2049, 1024, 8, 16, 64, 2049
flops = 3 * (2 * T * 12 * W * W * L) + 3.5 * 4 * B * H * s * s * d * L
print(f"cfg5 shape B={B} (attention forward: {'fp8' if A.FWD_FP8 else 'bf16'}): {dt * 1e3:.1f} ms/step, {flops / dt / 1e12:.0f} TFLOP/s, loss {float(loss):.4f}")
fl = 4 * B * H * s * s * d
for tag, mult in (("fwd", 1.0), ("dq", 1.5), ("dkdv", 2.0)):
    ts = sorted(a.elapsed_time(b) for a, b in A.KERNEL_EVENTS[tag])
    if not ts:
        continue
    med = ts[len(ts) // 2]
    print(f"  attention {tag}: {med * 1e3:.0f} us in situ, {fl * mult / med / 1e9:.0f} TFLOP/s")
