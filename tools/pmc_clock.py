"""Dev tool: effective clock of the attention kernels from a rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace csv run (GUI_ACTIVE / 8 XCDs / duration)."""
import csv, collections, glob, sys
f = glob.glob(sys.argv[1] + "/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "npcd::attn" not in k: continue
    agg[k.split("<")[0].split("::")[-1]].append((float(r["Counter_Value"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
for k, v in agg.items():
    v = v[len(v)//2:]
    clk = sorted(c / 8 / d for c, d in v)
    print(f"  {k}: {clk[len(clk)//2]:.3f} GHz, {sorted(d for _, d in v)[len(v)//2] / 1e3:.1f} us")
