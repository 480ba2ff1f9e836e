#!/bin/bash
# dev tool: a variant of libnpcd_hip.so in which ONLY csrc/gemm_nt.hip is rebuilt with extra -D flags (all other objects come from the
# regular in-tree build, csrc/build/*.o) -> neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_<tag>.so (git-ignored, travels with gpurun)
# usage: tools/build_lin_variant.sh <tag> -DNPCD_LIN_DIAG=3 ...      (load it with NPCD_HIP_LIB=<path>)
set -e
tag=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/neural-point-cloud-diffusion_amd/csrc
O=/tmp/npcd_lin_$tag; mkdir -p $O $R/neural-point-cloud-diffusion_amd/lib/diag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans "$@" -c $C/gemm_nt.hip -o $O/gemm_nt.o
objs=$(ls $C/build/*.o | grep -v gemm_nt.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_$tag.so $objs $O/gemm_nt.o
echo built lib/diag/libnpcd_hip_$tag.so
