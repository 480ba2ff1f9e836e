"""Dev helper: kernels of a rocprofv3 rocpd database between marker launches, sorted by LAUNCH COUNT per period."""
import sqlite3, collections, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 1
a, b = idx[skip], idx[-1]
n = len(idx) - 1 - skip
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[a:b]:
    agg[r[0][:110]][0] += 1; agg[r[0][:110]][1] += r[2] - r[1]
print(f"{n} periods, {sum(v[0] for v in agg.values()) / n:.0f} launches per period")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"{v[0] / n:6.1f} launches {v[1] / 1e3 / n:8.1f} us  {k}")
