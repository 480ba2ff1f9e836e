"""Dev tool: per-kernel mean of rocprofv3 --pmc counters (csv)."""
import csv, collections, sys
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r['Kernel_Name']
        if 'npcd' not in k: continue
        agg[k.split('<')[0].split('(')[0].split('::')[-1]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(k)
        for c, vals in v.items():
            vals = vals[2:] if len(vals) > 3 else vals
            print(f"   {c:28s} {sum(vals)/len(vals):16.0f}  (n={len(vals)})")
