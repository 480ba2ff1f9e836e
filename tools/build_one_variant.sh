#!/bin/bash
# dev tool: a variant of libnpcd_hip.so in which ONE source of csrc/ is rebuilt with extra -D flags (all other objects come from the
# regular in-tree build, csrc/build/*.o) -> neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_<tag>.so (git-ignored, travels with gpurun)
# usage: tools/build_one_variant.sh <source stem, e.g. attention; or several: shade,shade_rows> <tag> -DNPCD_FWD_WAVES=2 ...      (load it with NPCD_HIP_LIB=<path>)
set -e
stem=$1; tag=$2; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/neural-point-cloud-diffusion_amd/csrc
O=/tmp/npcd_var_$tag; mkdir -p $O $R/neural-point-cloud-diffusion_amd/lib/diag
objs=$(ls $C/build/*.o)
for st in ${stem//,/ }; do
extra=$(python3 - <<EOF
import sys; sys.path.insert(0, "$C")
import build
print(" ".join(build.SOURCES["$st.hip"]))
EOF
)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -fno-fast-math $extra "$@" -c $C/$st.hip -o $O/$st.o
objs="$(echo "$objs" | grep -v "/build/$st.o")
$O/$st.o"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/neural-point-cloud-diffusion_amd/lib/diag/libnpcd_hip_$tag.so $objs
echo built lib/diag/libnpcd_hip_$tag.so
