#!/bin/bash
# GPU box: rocprofv3 kernel stats of the step ONE RANK of the 8-GPU job computes (per-GPU batch 8, T = 4,104 tokens; no communication):
# 3 warm-up + 20 timed steps of tools/probes/gpu_dev_b8.py.  -> gpurun_out/<tag>_b8/{kernel_stats.csv, summary.txt}
# usage: tools/run_b8_profile.sh [tag] [per-GPU batch]
TAG=${1:-r5}; B=${2:-8}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_b8; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/tools/probes/gpu_dev_b8.py $B 20 > $O/unprofiled.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b8prof -o b -- python3 $R/tools/probes/gpu_dev_b8.py $B 20 > $O/profiled.txt 2>&1
cp /tmp/b8prof/b_kernel_stats.csv $O/kernel_stats.csv
cd $R; python3 tools/step_profile_summary.py $O/kernel_stats.csv 29 > $O/summary.txt 2>&1
cat $O/unprofiled.txt $O/profiled.txt | grep "B="; cat $O/summary.txt
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
g = sorted((r for r in rows if "Cijk" in r["Name"]), key=lambda r: -float(r["TotalDurationNs"]))
print("-- gemm")
for r in g[:24]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / 29:7.3f} ms/step  {float(r['Calls']) / 29:6.1f} calls/step  avg {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:150]}")
PY
