#!/bin/bash
# GPU box: the profile set of one round under gpurun_out/<tag>_profiles/ (copy what is to be judged into profiles/):
#   (1) render-only kernel stats + SQ / LDS counter passes, S = 128 and 64   (tools/run_render_profile.sh)
#   (2) attention kernels: SQ counters at the step's shape (B 64, n 513) and at B 16 / n 2049 (the 64-rows-per-wave forward),
#       FETCH_SIZE / WRITE_SIZE passes at the step's shape
#   (3) kernel stats of the bench's timed region alone + its bench line      (tools/run_stats.sh)
# usage: tools/run_round_profiles.sh [tag]
TAG=${1:-r5}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_profiles; mkdir -p $O
bash $R/tools/run_render_profile.sh $TAG > $O/render.log 2>&1; tail -30 $O/render.log
cp $R/gpurun_out/${TAG}_render/${TAG}_render_sq_pmc.json $R/gpurun_out/${TAG}_render/${TAG}_shade_sq_pmc.json $O/ 2>/dev/null
for S in 128 64; do cp $R/gpurun_out/${TAG}_render/kernel_stats_S$S.csv $O/${TAG}_render_kernel_stats_S$S.csv; done
cd /tmp; export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
export REPS=8
timeout 600 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/attn_sq -- python3 $R/tools/probes/gpu_dev_attn_only.py > $O/attn_sq.log 2>&1
export NPCD_B=16 NPCD_N=2049
timeout 600 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/attn_sq_n2049 -- python3 $R/tools/probes/gpu_dev_attn_only.py > $O/attn_sq_n2049.log 2>&1
unset NPCD_B NPCD_N
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/attn_fetch -- python3 $R/tools/probes/gpu_dev_attn_time.py 10 > $O/attn_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/attn_write -- python3 $R/tools/probes/gpu_dev_attn_time.py 10 > $O/attn_write.log 2>&1
# (2b) the elementwise kernels of the step (add + LayerNorm, LayerNorm backward, GELU, GELU backward + column sums): FETCH_SIZE / WRITE_SIZE
export NPCD_EW_ONLY_GELU_COLSUM=1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/ew_fetch -- python3 $R/tools/probes/gpu_dev_ew_time.py > $O/ew_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/ew_write -- python3 $R/tools/probes/gpu_dev_ew_time.py > $O/ew_write.log 2>&1
unset NPCD_EW_ONLY_GELU_COLSUM
cd $R
python3 tools/make_traffic_json.py ew $(ls $O/ew_fetch/*/*counter_collection.csv | head -1) $(ls $O/ew_write/*/*counter_collection.csv | head -1) $O/${TAG}_elementwise_hbm_traffic_pmc.json
python3 tools/make_sq_pmc_json.py $(dirname $(ls $O/attn_sq/*/*counter_collection.csv | head -1)) $O/${TAG}_attention_sq_pmc.json
python3 tools/make_sq_pmc_json.py $(dirname $(ls $O/attn_sq_n2049/*/*counter_collection.csv | head -1)) $O/${TAG}_attention_sq_pmc_n2049.json
python3 tools/make_traffic_json.py attn $(ls $O/attn_fetch/*/*counter_collection.csv | head -1) $(ls $O/attn_write/*/*counter_collection.csv | head -1) $O/${TAG}_attention_hbm_traffic_pmc.json
bash $R/tools/run_stats.sh > $O/stats.log 2>&1; tail -12 $O/stats.log
cp $R/gpurun_out/stats/kernel_stats.csv $O/${TAG}_bench_timed_region_kernel_stats.csv
cp $R/gpurun_out/stats/bench_line.json $O/${TAG}_bench_timed_region.json
ls -la $O | head -40
