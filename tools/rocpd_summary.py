"""Dev helper: per-kernel totals from a rocprofv3 rocpd database between the first and last launch of a marker kernel.
usage: rocpd_summary.py results.db marker [skip_first_markers]"""
import sqlite3, collections, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 1
a, b = idx[skip], idx[-1]
n = len(idx) - 1 - skip
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[a:b]:
    agg[r[0][:120]][0] += r[2] - r[1]; agg[r[0][:120]][1] += 1
tot = sum(v[0] for v in agg.values())
print(f"{n} periods; wall {(rows[b][1] - rows[a][1]) / 1e6 / n:.3f} ms/period, kernels busy {tot / 1e6 / n:.3f} ms/period")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{v[0] / 1e3 / n:9.1f} us {v[1] / n:6.1f} calls  {k}")
