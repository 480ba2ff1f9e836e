#!/bin/bash
# GPU box: rocprofv3 kernel TRACE (timestamps) of one rank's step at per-GPU batch 8 -> gpurun_out/<tag>_b8trace/{kernel_trace.csv, gaps.txt}
# usage: tools/run_b8_trace.sh [tag] [per-GPU batch] [env assignments ...]
TAG=${1:-r6}; B=${2:-8}; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_b8trace; mkdir -p $O
for kv in "$@"; do export "$kv"; done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/b8trace
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b8trace -o b -- python3 $R/tools/probes/gpu_dev_b8.py $B 12 > $O/profiled.txt 2>&1
cp /tmp/b8trace/b_kernel_trace.csv $O/kernel_trace.csv
cp /tmp/b8trace/b_kernel_stats.csv $O/kernel_stats.csv
cd $R; python3 tools/step_gap_analysis.py $O/kernel_trace.csv 10 > $O/gaps.txt 2>&1
grep "B=" $O/profiled.txt; cat $O/gaps.txt
