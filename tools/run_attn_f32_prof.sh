#!/bin/bash
# GPU box: per-kernel times of the fp32 attention (forward + backward kernels) at B = 16, n = 513, H = 16
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn_f32; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_f32 -o r -- python3 $R/tools/probes/gpu_dev_attn_f32_bwd.py 16 > $O/log.txt 2>&1
cp /tmp/rp_f32/r_kernel_stats.csv $O/kernel_stats.csv
tail -1 $O/log.txt
python3 $R/tools/kernel_stats_grep.py $O/kernel_stats.csv kernel | head -8
