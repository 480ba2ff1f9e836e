#!/bin/bash
# GPU box: per-kernel times of a 128 x 128 render with the ray-ordered compact lists and with the atomic one-launch form
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/compact_ab; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export NPCD_RENDERS=50 NPCD_S=${NPCD_S:-128}
for o in 1 0; do
  export NPCD_COMPACT_ORDERED=$o
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$o -o r -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/log_$o.txt 2>&1
  cp /tmp/rp_$o/r_kernel_stats.csv $O/kernel_stats_ordered$o.csv
  echo "== ordered=$o"; python3 $R/tools/kernel_stats_grep.py $O/kernel_stats_ordered$o.csv kernel | head -12
done
