#!/bin/bash
# GPU box: the shading kernels in their 16x16x32 (default) and 32x32x16 forms under the SQ / LDS counters (S = 128, 30 views each).
TAG=${1:-r5}
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
export NPCD_RENDERS=30 NPCD_S=128
for M in 1 0; do
  export NPCD_SHADE_PAIRS16=$M NPCD_SHADE_POINTS16=$M
  O=$R/gpurun_out/${TAG}_shape_$M; mkdir -p $O
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s16_$M -o r -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/stats_S128.log 2>&1
  cp /tmp/s16_$M/r_kernel_stats.csv $O/kernel_stats_S128.csv
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_S128 -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/sq_S128.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/mem_S128 -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/mem_S128.log 2>&1
  cd $R; python3 tools/kernel_stats_grep.py $O/kernel_stats_S128.csv kernel | head -4
  python3 tools/make_render_pmc_json.py $O $O/all.json $O/shade.json > /dev/null 2>&1; cd /tmp
done
