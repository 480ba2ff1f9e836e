#!/bin/bash
# dev tool: compile csrc/gemm_nt.hip alone and print the per-kernel register / spill report (+ extra flags)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -Rpass-analysis=kernel-resource-usage "$@" -c $(dirname "$0")/../neural-point-cloud-diffusion_amd/csrc/gemm_nt.hip -o /tmp/gemm_nt.o 2>&1 | grep -E "error|warning:|Function Name|VGPRs:|VGPRs Spill|Scratch" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//'
