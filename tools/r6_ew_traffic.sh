#!/bin/bash
# re-take the elementwise kernels' HBM traffic counters (the source changed after the round's profile set)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_profiles; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rm -rf $O/ew_fetch $O/ew_write
export NPCD_EW_ONLY_GELU_COLSUM=1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/ew_fetch -- python3 $R/tools/probes/gpu_dev_ew_time.py > $O/ew_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/ew_write -- python3 $R/tools/probes/gpu_dev_ew_time.py > $O/ew_write.log 2>&1
unset NPCD_EW_ONLY_GELU_COLSUM
cd $R
python3 tools/make_traffic_json.py ew $(ls $O/ew_fetch/*/*counter_collection.csv | head -1) $(ls $O/ew_write/*/*counter_collection.csv | head -1) $O/r6_elementwise_hbm_traffic_pmc.json
cat $O/r6_elementwise_hbm_traffic_pmc.json | head -40
