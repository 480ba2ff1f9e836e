#!/bin/bash
# dev tool: gfx950 ISA listing + resource usage of the attention kernels -> /tmp/asm/attention.s
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -fno-slp-vectorize $NPCD_EXTRA_FLAGS -S --cuda-device-only /root/repo/neural-point-cloud-diffusion_amd/csrc/attention.hip -o /tmp/asm/attention.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -i "error\|Function Name\| VGPRs:\|VGPRs Spill\|ScratchSize\|Occupancy" | grep -v "F16EEE" 
