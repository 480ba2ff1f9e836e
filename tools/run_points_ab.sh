#!/bin/bash
# GPU box: per-kernel time of the point-level shading kernel for diagnostic builds (lib/diag/libnpcd_hip_<tag>.so) against the default
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/points_ab; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export NPCD_RENDERS=50 NPCD_S=${NPCD_S:-128}
for tag in default "$@"; do
  if [ $tag = default ]; then unset NPCD_HIP_LIB; else export NPCD_HIP_LIB=$R/neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_$tag.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$tag -o r -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/log_$tag.txt 2>&1
  cp /tmp/rp_$tag/r_kernel_stats.csv $O/kernel_stats_$tag.csv
  echo "== $tag: $(tail -1 $O/log_$tag.txt)"; python3 $R/tools/kernel_stats_grep.py $O/kernel_stats_$tag.csv shade_ | head -3
done
