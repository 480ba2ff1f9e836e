#!/bin/bash
O=gpurun_out/r6_arena.txt; : > $O
python -m pytest tests/test_gpu_fused.py -x -q >> $O 2>&1
for i in 1 2; do
for cfg in "NPCD_STEP_ARENA=0" "NPCD_STEP_ARENA=1"; do
echo "== $cfg" >> $O
env $cfg python tools/probes/gpu_dev_b8.py 8 30 2>&1 | grep "B=" >> $O
done
done
echo "== B=16 arena 0/1" >> $O
NPCD_STEP_ARENA=0 python tools/probes/gpu_dev_b8.py 16 30 2>&1 | grep "B=" >> $O
python tools/probes/gpu_dev_b8.py 16 30 2>&1 | grep "B=" >> $O
grep -v amdgpu.ids $O | tail -30
