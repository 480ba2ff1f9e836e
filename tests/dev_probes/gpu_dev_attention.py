"""Developer probe (not a pytest file): print attention kernel errors for several shapes."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "neural-point-cloud-diffusion_amd"))
import torch
from oracle import denoiser as od
from npcd.hip.attention import attention_qkvpacked

def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())

for (B, n, H) in [(1, 64, 1), (1, 32, 1), (1, 1, 1), (2, 65, 2), (1, 513, 2), (2, 130, 3)]:
    g = torch.Generator().manual_seed(n)
    qkv = torch.randn(B, n, 3 * H * 64, generator=g).bfloat16()
    gout = torch.randn(B, n, H * 64, generator=g).bfloat16()
    x = qkv.cuda().requires_grad_(True)
    out = attention_qkvpacked(x, H)
    out.backward(gout.cuda())
    r = qkv.float().requires_grad_(True)
    ro = od.attention_qkvpacked(r, H)
    (ro * gout.float()).sum().backward()
    d = 64
    gq = x.grad.float().cpu().view(B, n, H, 3, d); rq = r.grad.view(B, n, H, 3, d)
    print(f"B{B} n{n} H{H}: fwd {rel(out, ro):.2e} dq {rel(gq[..., 0, :], rq[..., 0, :]):.2e} "
          f"dk {rel(gq[..., 1, :], rq[..., 1, :]):.2e} dv {rel(gq[..., 2, :], rq[..., 2, :]):.2e}", flush=True)

# timing at BASELINE cfg 2
B, n, H = 64, 513, 16
qkv = torch.randn(B, n, 3 * H * 64, device="cuda").bfloat16().requires_grad_(True)
gout = torch.randn(B, n, H * 64, device="cuda").bfloat16()
for _ in range(3):
    out = attention_qkvpacked(qkv, H); out.backward(gout)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record()
for _ in range(20): out = attention_qkvpacked(qkv, H)
e[1].record()
for _ in range(20):
    out = attention_qkvpacked(qkv, H); out.backward(gout)
e[2].record(); torch.cuda.synchronize()
tf = e[0].elapsed_time(e[1]) / 20; tfb = e[1].elapsed_time(e[2]) / 20
fl = 4 * B * H * n * n * 64
print(f"fwd {tf*1e3:.1f} us  {fl/tf/1e9:.1f} TFLOP/s ; fwd+bwd {tfb*1e3:.1f} us ; bwd {(tfb-tf)*1e3:.1f} us {2.5*fl/(tfb-tf)/1e9:.1f} TFLOP/s(alg)")
