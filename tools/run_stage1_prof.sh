#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s1prof; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_s1 -o r -- python3 $R/tools/probes/gpu_dev_stage1_time.py > $O/log.txt 2>&1
cp /tmp/rp_s1/r_kernel_stats.csv $O/kernel_stats.csv
tail -3 $O/log.txt
python3 $R/tools/kernel_stats_grep.py $O/kernel_stats.csv "" | head -28
