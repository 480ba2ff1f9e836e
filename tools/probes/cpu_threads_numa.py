"""Dev probe: why is the CPU oracle's B = 4 denoiser step 4.4 x SLOWER on 128 threads than on 32?
Times the same step (bench.cpu_baseline's workload) at several thread counts, unpinned and pinned to the cores of one NUMA node /
one socket, and prints the host's topology (lscpu).  usage: python tools/probes/cpu_threads_numa.py"""
import os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
import bench
from oracle import denoiser as od, diffusion as odf
try:
    print(subprocess.run(["lscpu"], capture_output=True, text=True).stdout.split("Vulnerab")[0][-1800:], flush=True)
except Exception as e:       # noqa: BLE001
    print("lscpu:", e)
CFG = bench.CFG
B = 4
params = od.init_params(CFG["coords_dim"], CFG["feats_dim"], CFG["width"], CFG["layers"], CFG["heads"], seed=0)
leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
g = torch.Generator().manual_seed(42)
c0 = torch.randn(B, 3, CFG["num_points"], generator=g)
f0 = torch.rand(B, CFG["feats_dim"], CFG["num_points"], generator=g) * 2 - 1
tab = odf.schedule_tables()
t = torch.randint(0, 1000, (B,), generator=g)
cn, fn = torch.randn(c0.shape, generator=g), torch.randn(f0.shape, generator=g)


def step():
    for v in leaves.values():
        v.grad = None
    loss, _, _ = odf.p_losses(tab, lambda c, f, tt: od.denoiser_forward(leaves, c, f, tt, CFG["heads"]), c0, f0, t, cn, fn)
    loss.backward()


all_cpus = sorted(os.sched_getaffinity(0))
print("usable CPUs:", len(all_cpus), flush=True)
cases = [("32 threads, unpinned", 32, None), ("64 threads, unpinned", 64, None), ("128 threads, unpinned", 128, None),
         ("32 threads, pinned to the first 32 CPUs", 32, all_cpus[:32]), ("64 threads, pinned to the first 64 CPUs", 64, all_cpus[:64]),
         ("128 threads, pinned to the first 128 CPUs", 128, all_cpus[:128])]
for name, th, pin in cases:
    if th > len(all_cpus):
        continue
    os.sched_setaffinity(0, pin if pin else all_cpus)
    torch.set_num_threads(th)
    step()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter(); step(); ts.append(time.perf_counter() - t0)
    print(f"{name}: {min(ts):.2f} s per forward + backward (B = {B})", flush=True)
os.sched_setaffinity(0, all_cpus)
