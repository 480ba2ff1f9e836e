"""Dev probe: the own weight-gradient kernel (csrc/gemm.hip, npcd_wgrad) against the library's row-split form (fused._wgrad's
current path) for the four Linear shapes of a block: error against an fp64 product on a sample, HIP-event time, interleaved.
usage: python3 tools/probes/gpu_dev_wgrad_own.py [T ...]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import elementwise as ew
f32 = torch.float32
Ts = [int(a) for a in sys.argv[1:]] or [32832]
shapes = (("c_qkv", 3072, 1024), ("attn.c_proj", 1024, 1024), ("c_fc", 4096, 1024), ("mlp.c_proj", 1024, 4096))


def lib_wgrad(dy, x, out):
    T = dy.shape[0]
    small = out.numel() <= (1 << 20)
    S = 8 if small else 4
    S = min(S, max(1, T // (2048 if small else 4096)))
    while S > 1 and T % S:
        S //= 2
    if S == 1:
        torch.mm(dy.t(), x, out_dtype=f32, out=out)
        return
    part = torch.bmm(dy.view(S, T // S, -1).transpose(1, 2), x.view(S, T // S, -1), out_dtype=f32)
    if not ew.sum_slices(part, out):
        torch.sum(part, dim=0, out=out)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for T in Ts:
    for name, N, K in shapes:
        torch.manual_seed(0)
        dy = torch.randn(T, N, device="cuda").bfloat16()
        x = torch.randn(T, K, device="cuda").bfloat16()
        o1, o2 = torch.empty(N, K, device="cuda"), torch.empty(N, K, device="cuda")
        assert ew.wgrad(dy, x, o1)
        lib_wgrad(dy, x, o2)
        ref = dy[:, :64].double().t() @ x[:, :96].double()
        e1 = float((o1[:64, :96].double() - ref).abs().max() / ref.abs().max())
        e2 = float((o2[:64, :96].double() - ref).abs().max() / ref.abs().max())
        dmax = float((o1 - o2).abs().max() / o2.abs().max())
        a = [timeit(lambda: ew.wgrad(dy, x, o1)), 0, 0]
        b = [timeit(lambda: lib_wgrad(dy, x, o2)), 0, 0]
        a[1], b[1] = timeit(lambda: ew.wgrad(dy, x, o1)), timeit(lambda: lib_wgrad(dy, x, o2))
        a[2], b[2] = timeit(lambda: ew.wgrad(dy, x, o1)), timeit(lambda: lib_wgrad(dy, x, o2))
        fl = 2 * T * N * K
        print(f"T={T} {name:12s} own {min(a):7.1f} us ({fl / min(a) / 1e6:5.0f} TF/s)  library {min(b):7.1f} us ({fl / min(b) / 1e6:5.0f} TF/s)  "
              f"err own {e1:.1e} lib {e2:.1e}  own-vs-lib {dmax:.1e}  slices {ew.lib().npcd_wgrad_slices(T, N, K)}", flush=True)
