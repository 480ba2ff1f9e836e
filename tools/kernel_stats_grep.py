"""Dev tool: print name / calls / average us of the kernels in a rocprofv3 kernel_stats.csv whose name contains a substring.
usage: kernel_stats_grep.py <kernel_stats.csv> <substring> [...]"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(w in r["Name"] for w in sys.argv[2:]):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us")
