"""Dev probe for counter passes: the c_fc-shaped product (T = 32768, N = 4096, K = 1024, bf16 + bias) REPS times on the own NT kernel
(csrc/gemm_nt.hip) and REPS times on the library (tuned solution file loaded), nothing else on the device."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, os.path.join(R, "neural-point-cloud-diffusion_amd")]
import torch
from npcd.hip import linear as hl
tuned = os.path.join(R, "profiles", "tunableop_gfx950.csv")
if os.path.exists(tuned):
    import torch.cuda.tunable as tun
    tun.enable(True); tun.tuning_enable(False); tun.set_filename("/tmp/npcd_tunableop_unused_probe.csv"); tun.read_file(tuned)
T, N, K = 32768, int(os.environ.get("N", 4096)), int(os.environ.get("K", 1024))
reps = int(os.environ.get("REPS", "10"))
torch.manual_seed(0)
x = torch.randn(T, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.randn(N, device="cuda").bfloat16()
out = torch.empty(T, N, device="cuda", dtype=torch.bfloat16)
for _ in range(reps):
    hl.linear_fwd(x, w, b, out=out)
torch.cuda.synchronize()
for _ in range(reps):
    torch.addmm(b, x, w.t(), out=out)
torch.cuda.synchronize()
print("done")
