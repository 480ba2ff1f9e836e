#!/bin/bash
# dev tool: build a DIAGNOSTIC variant of libnpcd_hip.so with extra -D flags into neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_<tag>.so (git-ignored, travels with gpurun)
# usage: tools/build_diag_lib.sh <tag> -DNPCD_DIAG_HALF_MFMA ...      (load it with NPCD_HIP_LIB=<path>)
set -e
tag=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/neural-point-cloud-diffusion_amd/csrc
O=/tmp/npcd_diag_$tag; mkdir -p $O $R/neural-point-cloud-diffusion_amd/lib/diag
COMMON="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fno-fast-math"
/opt/rocm/bin/hipcc $COMMON "$@" -c $C/api.hip -o $O/api.o &
/opt/rocm/bin/hipcc $COMMON -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans -fno-slp-vectorize "$@" -c $C/attention.hip -o $O/attention.o &
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off "$@" -c $C/geometry.hip -o $O/geometry.o &
/opt/rocm/bin/hipcc $COMMON -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans "$@" -c $C/shade.hip -o $O/shade.o &
wait
/opt/rocm/bin/hipcc $COMMON "$@" -c $C/elementwise.hip -o $O/elementwise.o &
/opt/rocm/bin/hipcc $COMMON -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans "$@" -c $C/gemm.hip -o $O/gemm.o &
/opt/rocm/bin/hipcc $COMMON -ffp-contract=off "$@" -c $C/pairs.hip -o $O/pairs.o &
/opt/rocm/bin/hipcc $COMMON -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans "$@" -c $C/pairs_mlp.hip -o $O/pairs_mlp.o &
/opt/rocm/bin/hipcc $COMMON -mllvm -amdgpu-mfma-vgpr-form -fno-honor-nans "$@" -c $C/shade_rows.hip -o $O/shade_rows.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_$tag.so $O/*.o
echo built $R/neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_$tag.so
