"""Dev tool: GPU idle time inside the denoiser training step from a rocprofv3 kernel trace (*_kernel_trace.csv of
`bench.py --no-render --no-cpu-baseline --no-proxy --steps K --warmup W`): the union of kernel intervals against the span of the last
K steps (found from the K + W optimizer launches), the total idle time per step and the largest gaps with the kernels on either side.
usage: step_gap_analysis.py kernel_trace.csv steps"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
K = int(sys.argv[2])
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
opt = [i for i, e in enumerate(ev) if "adamw_ema_kernel" in e[2]]
first = opt[-K - 1] + 1            # the kernel behind the optimizer pass of the step before the last K
last = opt[-1]
seg = ev[first:last + 1]
span = seg[-1][1] - seg[0][0]
busy, gaps, cur_end, prev = 0, [], seg[0][0], None
for s, e, n in seg:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n))
        busy += e - s
        cur_end = e
    else:
        busy += max(0, e - cur_end)
        cur_end = max(cur_end, e)
    if e >= cur_end:
        prev = n
print(f"{K} steps: span {span / K / 1e6:.3f} ms per step, kernels (union) {busy / K / 1e6:.3f} ms, idle {(span - busy) / K / 1e6:.3f} ms = {100 * (span - busy) / span:.1f} %, "
      f"{len(gaps) / K:.0f} gaps per step, mean {sum(g[0] for g in gaps) / max(len(gaps), 1) / 1e3:.2f} us")
import collections
by = collections.Counter()
for g, a, b in gaps:
    by[(a.split("(")[0][-50:], b.split("(")[0][-50:])] += g
print("largest idle totals by (kernel before -> kernel after), ms per step:")
for (a, b), t in by.most_common(14):
    print(f"  {t / K / 1e6:7.3f}   {a}  ->  {b}")
hist = collections.Counter(min(int(g[0] / 1e3) // 2 * 2, 40) for g in gaps)
print("gap histogram (us bucket: count per step):", {k: round(v / K, 1) for k, v in sorted(hist.items())})
