#!/bin/bash
# GPU box: only the elementwise FETCH_SIZE / WRITE_SIZE passes of tools/run_round_profiles.sh -> gpurun_out/<tag>_ew/<tag>_elementwise_hbm_traffic_pmc.json
TAG=${1:-r5}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}_ew; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export NPCD_EW_ONLY_GELU_COLSUM=1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/ew_fetch -- python3 $R/tools/probes/gpu_dev_ew_time.py > $O/ew_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/ew_write -- python3 $R/tools/probes/gpu_dev_ew_time.py > $O/ew_write.log 2>&1
cd $R
python3 tools/make_traffic_json.py ew $(ls $O/ew_fetch/*/*counter_collection.csv | head -1) $(ls $O/ew_write/*/*counter_collection.csv | head -1) $O/${TAG}_elementwise_hbm_traffic_pmc.json
