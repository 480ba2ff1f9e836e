"""Dev probe: which GELU does the library's GEMM epilogue compute?  torch._addmm_activation(use_gelu=True) routes to the hipBLASLt
GELU epilogue; compare its result with erf-GELU and tanh-GELU of the same fp32 pre-activation."""
import torch
torch.manual_seed(0)
x = torch.randn(512, 256, device="cuda"); w = torch.randn(256, 384, device="cuda") * 0.1; b = torch.randn(384, device="cuda") * 0.1
pre = torch.addmm(b, x, w)
out = torch._addmm_activation(b, x, w, use_gelu=True)
erf = torch.nn.functional.gelu(pre); tanh = torch.nn.functional.gelu(pre, approximate="tanh")
print(f"fp32: max |epilogue - erf GELU| = {float((out - erf).abs().max()):.3e}, max |epilogue - tanh GELU| = {float((out - tanh).abs().max()):.3e}, "
      f"max |erf - tanh| = {float((erf - tanh).abs().max()):.3e}")
xb, wb, bb = x.bfloat16(), w.bfloat16(), b.bfloat16()
outb = torch._addmm_activation(bb, xb, wb, use_gelu=True).float()
preb = torch.addmm(bb.float(), xb.float(), wb.float())
print(f"bf16: max |epilogue - erf GELU| = {float((outb - torch.nn.functional.gelu(preb)).abs().max()):.3e}, "
      f"max |epilogue - tanh GELU| = {float((outb - torch.nn.functional.gelu(preb, approximate='tanh')).abs().max()):.3e}")
