import sys, os, shutil, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(R, "neural-point-cloud-diffusion_amd", "lib", "libnpcd_hip.so")
shutil.copy(lib, "/tmp/orig.so")
code = '''
import sys, os
R = %r
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip.attention import attention_qkvpacked
B, n, H = 64, 513, 16
qkv = torch.randn(B, n, 3 * H * 64, device="cuda").bfloat16()
for _ in range(3): attention_qkvpacked(qkv, H)
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): attention_qkvpacked(qkv, H)
e1.record(); torch.cuda.synchronize(); print("fwd us", e0.elapsed_time(e1) / 20 * 1e3)
''' % R
for a in ("orig", "1", "2", "3"):
    src = "/tmp/orig.so" if a == "orig" else os.path.join(R, "tests", f"libabl{a}.so")
    shutil.copy(src, lib)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print("ABL", a, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
shutil.copy("/tmp/orig.so", lib)
