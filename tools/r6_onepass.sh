#!/bin/bash
O=gpurun_out/r6_onepass.txt; : > $O
python -m pytest tests/test_gpu_fused.py tests/test_gpu_ddp.py -x -q >> $O 2>&1
for i in 1 2 3; do
for cfg in "NPCD_COLSUM_ONEPASS=0" "NPCD_COLSUM_ONEPASS=1"; do
echo "== $cfg" >> $O
env $cfg python tools/probes/gpu_dev_b8.py 8 30 2>&1 | grep "wall" >> $O
done
done
for cfg in "NPCD_COLSUM_ONEPASS=0" "NPCD_COLSUM_ONEPASS=1"; do
echo "== B=64 $cfg" >> $O
env $cfg python tools/probes/gpu_dev_b8.py 64 10 2>&1 | grep "wall" >> $O
done
grep -v amdgpu.ids $O | tail -24
