"""Dev probe: the three attention kernels alone at cfg-D (meant to be run under rocprofv3 --pmc)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip.attention import attention_qkvpacked
B, n, H = int(os.environ.get("NPCD_B", 64)), int(os.environ.get("NPCD_N", 513)), 16
qkv = torch.randn(B, n, 3 * H * 64, device="cuda").bfloat16()
if os.environ.get("NPCD_ZERO_DATA"):
    qkv.zero_()
qkv.requires_grad_(True)
gout = torch.randn(B, n, H * 64, device="cuda").bfloat16()
for _ in range(int(os.environ.get("REPS", 3))):
    out = attention_qkvpacked(qkv, H); out.backward(gout)
torch.cuda.synchronize()
