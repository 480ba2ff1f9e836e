"""Dev probe: where the HOST spends its time inside _RowSplitLinear.backward of the stage-1 step (no device syncs added): wraps the
torch / library calls it makes with wall-clock timers.  NPCD_ROWSPLIT_MIN selects which layers take that path."""
import sys, os, time, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch, bench
from npcd.models.pointnerf import train_path as tp
from npcd.hip import render as hr
acc = collections.defaultdict(float); cnt = collections.Counter()
def wrap(mod, name, tag=None):
    fn = getattr(mod, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[tag or name] += time.perf_counter() - t0; cnt[tag or name] += 1; return r
    setattr(mod, name, w)
for n in ("mm", "addmm", "empty", "sum"):
    wrap(torch, n)
_bmm = torch.bmm
log = []
def bmm_logged(a, b, **k):
    t0 = time.perf_counter(); r = _bmm(a, b, **k); dt = time.perf_counter() - t0
    acc["bmm"] += dt; cnt["bmm"] += 1
    log.append((tuple(a.shape), tuple(b.shape), a.stride(), a.dtype, sorted(k), round(dt * 1e3, 3)))
    return r
torch.bmm = bmm_logged
wrap(hr, "leaky_bwd_colsum")
orig_bwd = tp._RowSplitLinear.backward
def timed_bwd(ctx, dy):
    t0 = time.perf_counter(); r = orig_bwd(ctx, dy); acc["_RowSplitLinear.backward total"] += time.perf_counter() - t0; cnt["_RowSplitLinear.backward total"] += 1; return r
tp._RowSplitLinear.backward = staticmethod(timed_bwd)
r = bench.bench_stage1(torch.device("cuda", 0), mlp_dtype=torch.bfloat16)
print({k: r[k] for k in ("ms_per_step", "ms_per_step_min_median_max")})
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"{k:36s} {v * 1e3 / 23:8.3f} ms/step over {cnt[k] / 23:6.1f} calls/step")
for e in log[:8] + log[-8:]:
    print(e)
