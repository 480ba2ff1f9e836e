"""Dev tool: per-step kernel time of the denoiser training step from a rocprofv3 kernel_stats.csv of
`bench.py --no-render --no-cpu-baseline --no-proxy --steps K --warmup W`, grouped: library GEMMs / npcd kernels / everything else,
with the largest 'everything else' kernels listed.  usage: step_profile_summary.py kernel_stats.csv steps_plus_warmup"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
def cat(nm):
    return "gemm" if "Cijk" in nm else "npcd" if ("npcd" in nm or "attn_bwd_edge" in nm) else "other"
agg = {}
for r in rows:
    agg[cat(r["Name"])] = agg.get(cat(r["Name"]), 0) + float(r["TotalDurationNs"])
print({k: round(v / 1e6 / n, 2) for k, v in agg.items()}, "ms per step")
for c in ("npcd", "other"):
    t = sorted((r for r in rows if cat(r["Name"]) == c), key=lambda r: -float(r["TotalDurationNs"]))
    print("--", c)
    for r in t[:18]:
        print(f"{float(r['TotalDurationNs']) / 1e6 / n:7.3f} ms/step  {float(r['Calls']) / n:7.1f} calls/step  avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:110]}")
