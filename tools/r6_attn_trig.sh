#!/bin/bash
O=gpurun_out/r6_attn_trig.txt; : > $O
python -m pytest tests/test_gpu_attention.py -x -q >> $O 2>&1
for i in 1 2 3; do
echo "== trigger softmax (default build)" >> $O
python tools/probes/gpu_dev_attn_time.py 40 2>&1 | grep "fwd\|rel-L2" >> $O
echo "== per-tile maximum (round-5 form)" >> $O
NPCD_HIP_LIB=$GRAFT_REPO_ROOT/neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_notrig.so python tools/probes/gpu_dev_attn_time.py 40 2>&1 | grep "fwd\|rel-L2" >> $O
done
echo "== B=8" >> $O
python tools/probes/gpu_dev_attn_time.py 40 513 8 2>&1 | grep "fwd" >> $O
NPCD_HIP_LIB=$GRAFT_REPO_ROOT/neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_notrig.so python tools/probes/gpu_dev_attn_time.py 40 513 8 2>&1 | grep "fwd" >> $O
grep -v amdgpu.ids $O | tail -30
