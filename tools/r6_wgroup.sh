#!/bin/bash
# round 6: grouped weight-gradient launch -- parity test, standalone time, the rank step with / without it
O=gpurun_out/r6_wgroup.txt; : > $O
python -m pytest tests/test_gpu_fused.py -x -q -k "grouped_weight or own_weight or side_stream" >> $O 2>&1
python - >> $O 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "neural-point-cloud-diffusion_amd")
import torch
from npcd.hip import elementwise as ew
for T in (4104, 8208):
    trip = []
    for N, K in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]:
        trip.append((torch.randn(T, N, device="cuda").bfloat16(), torch.randn(T, K, device="cuda").bfloat16(), torch.empty(N, K, device="cuda")))
    for _ in range(5): ew.wgrad_group(trip)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ew.wgrad_group(trip)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    fl = sum(2 * T * t[0].shape[1] * t[1].shape[1] for t in trip)
    print(f"T={T}: grouped wgrad launch {us:.1f} us = {fl / us / 1e6:.0f} TF/s (192 tiles)")
PY
for i in 1 2; do
python tools/probes/gpu_dev_b8.py 8 30 >> $O 2>&1
NPCD_WGRAD_GROUP=0 python tools/probes/gpu_dev_b8.py 8 30 >> $O 2>&1
NPCD_B8_GRAPH=1 python tools/probes/gpu_dev_b8.py 8 30 >> $O 2>&1
NPCD_B8_GRAPH=1 NPCD_WGRAD_GROUP=0 python tools/probes/gpu_dev_b8.py 8 30 >> $O 2>&1
done
NPCD_WGRAD_GROUP_MAX_T=9000 python tools/probes/gpu_dev_b8.py 16 30 >> $O 2>&1
python tools/probes/gpu_dev_b8.py 16 30 >> $O 2>&1
grep -v "amdgpu.ids" $O | tail -30
