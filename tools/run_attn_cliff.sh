#!/bin/bash
# GPU box: the 32-row attention forward in its two instantiations -- ROWX = false (166 registers: three waves per SIMD) and ROWX = true
# (NPCD_ATTN_ROWX32=1: 170+ registers: two) -- under the SQ counters: the occupancy cliff that any form of this kernel carrying one
# more item of state (a persistent loop's next Q fragments, the last query row's partial state) falls over
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn_cliff; mkdir -p $O
export REPS=8
for mode in default rowx; do
  if [ $mode = rowx ]; then export NPCD_ATTN_ROWX32=1; else unset NPCD_ATTN_ROWX32; fi
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq_$mode -- python3 $R/tools/probes/gpu_dev_attn_only.py > $O/sq_$mode.log 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/occ_$mode -- python3 $R/tools/probes/gpu_dev_attn_only.py > $O/occ_$mode.log 2>&1
done
python3 $R/tools/make_attn_cliff_json.py $O $O/r4_attention_occupancy_cliff.json
