"""sha256 of the kernel sources a counter file belongs to (written into the file as `_meta.source_sha256`; bench.py compares them
with the in-tree sources and marks a counter file whose kernels have changed since as stale)."""
import hashlib, os
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neural-point-cloud-diffusion_amd", "csrc")


def source_hashes(*files):
    return {f: hashlib.sha256(open(os.path.join(CSRC, f), "rb").read()).hexdigest() for f in files if os.path.exists(os.path.join(CSRC, f))}
