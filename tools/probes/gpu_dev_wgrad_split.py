"""Dev probe: row-split weight-gradient GEMM (bf16 in, fp32 out) + npcd_sum_slices for S = 1, 2, 4, 8 slices, the four Linear shapes
of a block at T = 32832 tokens (what fused._wgrad runs); each timed right after a data-gradient GEMM on the same operands."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import elementwise as ew
T = int(sys.argv[1]) if len(sys.argv) > 1 else 32832
f32 = torch.float32
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, N, K in (("c_qkv", 3072, 1024), ("attn.c_proj", 1024, 1024), ("c_fc", 4096, 1024), ("mlp.c_proj", 1024, 4096)):
    dy = torch.randn(T, N, device="cuda").bfloat16(); x = torch.randn(T, K, device="cuda").bfloat16()
    out = torch.empty(N, K, device="cuda")
    res = []
    for S in (1, 2, 4, 8):
        if T % S: continue
        def run():
            if S == 1:
                torch.mm(dy.t(), x, out_dtype=f32, out=out)
            else:
                part = torch.bmm(dy.view(S, T // S, N).transpose(1, 2), x.view(S, T // S, K), out_dtype=f32)
                ew.sum_slices(part, out)
        res.append(f"S={S}: {timeit(run):6.1f} us")
    print(f"{name:12s} " + "  ".join(res) + f"   ({2 * T * N * K / 1e9:.0f} GFLOP)", flush=True)
