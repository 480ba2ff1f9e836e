"""Dev probe: fp32 attention (forward + backward kernels) at the step's shape against the same math as fp32 library products
(what _AttnF32.backward did before the kernels existed).  argv: B (default 16)."""
import sys, os, math
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip.attention import attention_qkvpacked
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, H, d = 513, 16, 64
torch.manual_seed(0)
qkv = torch.randn(B, n, 3 * H * d, device="cuda", requires_grad=True)
gout = torch.randn(B, n, H * d, device="cuda")


def lib_path(qkv):
    x = qkv.view(B, n, H, 3 * d)
    q, k, v = x[..., :d], x[..., d:2 * d], x[..., 2 * d:]
    p = torch.softmax(torch.einsum("bthd,bshd->bhts", q, k) / math.sqrt(d), dim=-1)
    return torch.einsum("bhts,bshd->bthd", p, v).reshape(B, n, H * d)


def timed(fn, reps=5):
    for _ in range(2):
        qkv.grad = None
        (fn(qkv) * gout).sum().backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        qkv.grad = None
        (fn(qkv) * gout).sum().backward()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, qkv.grad.clone()


t_k, g_k = timed(lambda x: attention_qkvpacked(x, H))
t_l, g_l = timed(lib_path)
fl = 4 * B * H * n * n * d * 3.5
print(f"fp32 attention fwd+bwd, B={B} n={n} H={H}: kernels {t_k:.2f} ms ({fl / t_k / 1e9:.1f} TFLOP/s), library einsum/softmax autograd {t_l:.2f} ms; "
      f"max |grad diff| {float((g_k - g_l).abs().max()):.2e} (max |grad| {float(g_l.abs().max()):.2e})")
