#!/bin/bash
# GPU box: counters of the own NT GEMM (lin_kernel) beside the library's kernel on the c_fc shape (tools/probes/gpu_dev_lin_only.py):
# SQ issue / wait counters, LDS counters, L2 hit / miss, HBM-side bytes -- each group in a pass of its own.
# usage: tools/run_lin_pmc.sh [tag]    -> gpurun_out/<tag>_lin_pmc/
TAG=${1:-r4}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_lin_pmc; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export REPS=8
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/sq -- python3 $R/tools/probes/gpu_dev_lin_only.py > $O/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/mem -- python3 $R/tools/probes/gpu_dev_lin_only.py > $O/mem.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/l2 -- python3 $R/tools/probes/gpu_dev_lin_only.py > $O/l2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/tools/probes/gpu_dev_lin_only.py > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/tools/probes/gpu_dev_lin_only.py > $O/write.log 2>&1
cd $R
python3 tools/make_lin_pmc_json.py $O $O/${TAG}_gemm_c_fc_pmc.json
