"""Dev tool: one step's kernel timeline from a rocprofv3 kernel trace, per HIP stream / queue: for the last full step (between the last
two optimizer launches) the busy time of every queue, the in-queue gaps (end of a kernel to the start of the next on the SAME queue),
the overlap between queues, and -- with `--block i` -- the launch-by-launch timeline of the i-th backward block.
usage: step_timeline.py kernel_trace.csv [--dump N]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")) for r in rows), key=lambda e: e[0])
opt = [i for i, e in enumerate(ev) if "adamw_ema_kernel" in e[2]]
a, b = opt[-2] + 1, opt[-1]
seg = ev[a:b + 1]
t0, t1 = seg[0][0], seg[-1][1]
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(seg)} kernels")
byq = collections.defaultdict(list)
for e in seg:
    byq[(e[3], e[4])].append(e)


def short(n):
    if n.startswith("Cijk") or n.startswith("Custom"):
        import re
        m = re.search(r"(A\w+?_B\w+?)_(\w+?)_.*?(MT\d+x\d+x\d+)", n)
        return "gemm " + (m.group(1) + " " + m.group(2) + " " + m.group(3) if m else n[:40])
    return n.split("(")[0].replace("void ", "").replace("npcd::", "")[:48]


for q, es in sorted(byq.items()):
    busy = sum(e[1] - e[0] for e in es)
    gaps = [es[i + 1][0] - es[i][1] for i in range(len(es) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"queue/stream {q}: {len(es)} kernels, busy {busy / 1e6:.3f} ms, in-queue gaps {sum(pos) / 1e6:.3f} ms over {len(pos)} gaps "
          f"(median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.2f} us), back-to-back overlaps {sum(1 for g in gaps if g <= 0)}")
# union busy
cur, busy = t0, 0
for s, e, *_ in seg:
    if e > cur:
        busy += e - max(s, cur)
        cur = e
print(f"union busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms")
# time with >= 2 kernels running
pts = sorted([(s, 1) for s, e, *_ in seg] + [(e, -1) for s, e, *_ in seg])
depth, last, two = 0, t0, 0
for t, d in pts:
    if depth >= 2:
        two += t - last
    depth += d
    last = t
print(f"time with >= 2 kernels in flight {two / 1e6:.3f} ms")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, *_ in seg:
    agg[short(n)][0] += 1
    agg[short(n)][1] += e - s
print("per kernel in this step (calls, total ms, avg us):")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"  {c:4d} {t / 1e6:7.3f} {t / c / 1e3:8.1f}  {n}")
if "--dump" in sys.argv:
    N = int(sys.argv[sys.argv.index("--dump") + 1])
    mid = len(seg) * 2 // 3
    print("timeline excerpt (start us rel, dur us, queue, kernel):")
    for s, e, n, q, st in seg[mid:mid + N]:
        print(f"  {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q}/s{st}  {short(n)}")
