"""Dev helper: per-kernel means of the counters in a rocprofv3 --pmc output directory (launches after the first three).
usage: python3 tools/probes/pmc_lds_summary.py <dir> [kernel-name substring]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else "shade_p"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sub in k:
        acc[k.split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {n: round(sum(x[3:]) / max(len(x[3:]), 1) / 1e6, 2) for n, x in v.items()})
