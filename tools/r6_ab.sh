#!/bin/bash
# round 6: same-box A/B of the rank step (per-GPU batch 8): grouped weight gradients x per-block join, alternating, twice
O=gpurun_out/r6_ab.txt; : > $O
for i in 1 2; do
for cfg in "NPCD_WGRAD_GROUP=0 NPCD_WGRAD_JOIN_PER_BLOCK=1" "NPCD_WGRAD_GROUP=1 NPCD_WGRAD_JOIN_PER_BLOCK=1" "NPCD_WGRAD_GROUP=0" "NPCD_WGRAD_GROUP=1"; do
echo "== $cfg" >> $O
env $cfg python tools/probes/gpu_dev_b8.py 8 30 2>&1 | grep "B=" >> $O
done
done
echo "== B=16 group max 9000 / default" >> $O
NPCD_WGRAD_GROUP_MAX_T=9000 python tools/probes/gpu_dev_b8.py 16 30 2>&1 | grep "B=" >> $O
python tools/probes/gpu_dev_b8.py 16 30 2>&1 | grep "B=" >> $O
echo "== B=32 group max 17000 / default" >> $O
NPCD_WGRAD_GROUP_MAX_T=17000 python tools/probes/gpu_dev_b8.py 32 20 2>&1 | grep "B=" >> $O
python tools/probes/gpu_dev_b8.py 32 20 2>&1 | grep "B=" >> $O
cat $O
