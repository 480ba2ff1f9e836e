#!/bin/bash
O=gpurun_out/r6_fusedrays.txt; : > $O
python -m pytest tests/test_gpu_render.py -x -q >> $O 2>&1
for i in 1 2; do
for S in 128 64; do
for f in 1 0; do
echo "== S=$S fused=$f: $(NPCD_S=$S NPCD_RENDERS=100 NPCD_RENDER_FUSED_RAYS=$f python tools/probes/gpu_dev_render_time.py 2>&1 | tail -1)" >> $O
done; done; done
grep -v amdgpu.ids $O | tail -20
