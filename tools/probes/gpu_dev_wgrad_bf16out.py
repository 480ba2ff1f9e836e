"""Dev probe: weight-gradient GEMMs with bf16 OUTPUT (what autocast computes in the reference) -- default heuristic and
TunableOp-tuned -- against the fp32-output split form used now."""
import sys, os, torch
import torch.cuda.tunable as tun
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
shapes = (("c_qkv", 3072, 1024, 4), ("attn.c_proj", 1024, 1024, 8), ("c_fc", 4096, 1024, 4), ("mlp.c_proj", 1024, 4096, 4))
data = {}
for name, N, K, S in shapes:
    data[name] = (torch.randn(T, N, device="cuda").bfloat16(), torch.randn(T, K, device="cuda").bfloat16())
res = {n: {} for n, *_ in shapes}
for name, N, K, S in shapes:
    dy, x = data[name]
    S = max(1, min(S, T // 4096))
    while S > 1 and T % S: S //= 2
    out = torch.empty(N, K, device="cuda")
    a, b = dy.view(S, T // S, N), x.view(S, T // S, K)
    def cur():
        if S == 1: torch.mm(dy.t(), x, out_dtype=f32, out=out)
        else: torch.sum(torch.bmm(a.transpose(1, 2), b, out_dtype=f32), dim=0, out=out)
    res[name]["fp32 split (now)"] = timeit(cur)
    ob = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    def b16():
        torch.mm(dy.t(), x, out=ob); out.copy_(ob)
    res[name]["bf16 out + cast, heuristic"] = timeit(b16)
tun.enable(True); tun.tuning_enable(True); tun.set_max_tuning_duration(30); tun.set_max_tuning_iterations(20)
tun.set_filename("/tmp/tune_wgrad.csv")
for name, N, K, S in shapes:
    dy, x = data[name]
    out = torch.empty(N, K, device="cuda"); ob = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    def b16():
        torch.mm(dy.t(), x, out=ob); out.copy_(ob)
    res[name]["bf16 out + cast, tuned"] = timeit(b16)
for name, N, K, S in shapes:
    fl = 2 * T * N * K
    print(f"{name:12s} " + " | ".join(f"{k}: {v:6.1f} us ({fl / v / 1e6:5.0f} TF/s)" for k, v in res[name].items()), flush=True)
print(open("/tmp/tune_wgrad.csv").read() if os.path.exists("/tmp/tune_wgrad.csv") else "no csv")
