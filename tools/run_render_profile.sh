#!/bin/bash
# GPU box: render-only evidence of one round.  One 128 x 128 view per call, bench scene, S = 128 and S = 64
# separately:  (1) rocprofv3 --kernel-trace --stats  -> per-kernel averages;  (2) two --pmc passes (SQ issue / wait counters, then
# LDS / memory-instruction counters) over the same probe.  tools/make_render_pmc_json.py turns the counter CSVs into
# profiles/<tag>_render_sq_pmc.json (+ the shading subset as <tag>_shade_sq_pmc.json); copy the kernel stats next to them.
# usage: tools/run_render_profile.sh [tag]      (results under gpurun_out/<tag>_render/)
TAG=${1:-r4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_render; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export NPCD_RENDERS=30
for S in 128 64; do
  export NPCD_S=$S
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rprof_$S -o r -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/stats_S$S.log 2>&1
  cp /tmp/rprof_$S/r_kernel_stats.csv $O/kernel_stats_S$S.csv
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_S$S -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/sq_S$S.log 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/mem_S$S -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/mem_S$S.log 2>&1
  # HBM traffic: FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md, rocprofv3 PMC slots: they do not fit one pass)
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_S$S -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/fetch_S$S.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_S$S -- python3 $R/tools/probes/gpu_dev_render_time.py > $O/write_S$S.log 2>&1
  tail -1 $O/stats_S$S.log
done
cd $R
python3 tools/make_render_pmc_json.py $O $O/${TAG}_render_sq_pmc.json $O/${TAG}_shade_sq_pmc.json
python3 tools/make_render_traffic_json.py $O $O/${TAG}_render_hbm_traffic_pmc.json
for S in 128 64; do python3 tools/kernel_stats_grep.py $O/kernel_stats_S$S.csv kernel | head -14; done
