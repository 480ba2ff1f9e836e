#!/usr/bin/env python3
"""Build libnpcd_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python neural-point-cloud-diffusion_amd/csrc/build.py [--force]

Objects are rebuilt only when their source (or a header) is newer.  The library is written
in-tree to neural-point-cloud-diffusion_amd/lib/ so that it travels to the GPU box with the repo
snapshot (it is git-ignored, not gpurun-ignored).
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(os.path.dirname(HERE), "lib")
OBJ_DIR = os.path.join(HERE, "build")
LIB = os.path.join(LIB_DIR, "libnpcd_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# source -> extra flags.  Geometry kernels must reproduce the oracle's fp32 operation order
# bit-exactly, so they are compiled without FMA contraction.
SOURCES = {
    "api.hip": [],
    # VGPR-form MFMA: accumulators stay in VGPRs (gfx950 has a unified register file), no v_accvgpr moves
    # -fno-honor-nans: no NaN can arise in the softmax math; drops the canonicalising v_max before every fmaxf
    # -fno-slp-vectorize: packed fp32 VALU (v_pk_mul/add_f32) is slower than two scalar ops beside MFMAs and
    # mis-pairs the 16-bit packing (MI355X_MICROARCH.md, per-instruction cycle constants)
    "attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans", "-fno-slp-vectorize"],
    "attention_gen.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "geometry.hip": ["-ffp-contract=off"],
    "shade.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    # the rows kernel runs one wave per SIMD on the whole register file: accumulators (which the epilogues read) in VGPRs, the
    # activation fragments (matrix-instruction operands only) in AGPRs
    "shade_rows.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "elementwise.hip": [],
    "gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "gemm_nt.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "pairs.hip": ["-ffp-contract=off"],
    "pairs_mlp.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-honor-nans"],
    "split.hip": [],
    "points_x2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-fno-gpu-rdc", "-ffast-math" if False else "-fno-fast-math"]


def newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def compile_one(src, extra, force):
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    path = os.path.join(HERE, src)
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "npcd_hip.h"))
    if force or newer(path, obj) or any(newer(h, obj) for h in headers):
        cmd = [HIPCC, *COMMON, *extra, *os.environ.get("NPCD_EXTRA_FLAGS", "").split(), "-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = {s: f for s, f in SOURCES.items() if os.path.exists(os.path.join(HERE, s))}
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda kv: compile_one(kv[0], kv[1], force), srcs.items()))
    if force or any(newer(o, LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB) from {sorted(srcs)}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
