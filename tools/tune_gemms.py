"""Record hipBLASLt/rocBLAS solution choices for the GEMM shapes of the denoiser step with PyTorch TunableOp
(run on an MI355X; writes profiles/tunableop_gfx950.csv, which bench.py loads with tuning DISABLED)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
import torch.cuda.tunable as tun
import bench
out = os.path.join(R, "gpurun_out", os.environ.get("NPCD_TUNE_OUT", "tunableop_gfx950.csv"))
# NPCD_TUNE_ROTATE=<MB>: candidates are timed on operands rotated through a buffer of that size (cold caches, as in the step, where a
# product's inputs were just written by another kernel and its weights were last read a layer ago) instead of the same hot buffers
rot = int(os.environ.get("NPCD_TUNE_ROTATE", "0"))
tun.enable(True); tun.tuning_enable(True)
tun.set_max_tuning_duration(int(os.environ.get("NPCD_TUNE_MS", "40"))); tun.set_max_tuning_iterations(int(os.environ.get("NPCD_TUNE_ITERS", "30")))
if rot and hasattr(tun, "set_rotating_buffer_size"):
    tun.set_rotating_buffer_size(rot)
tun.set_filename(out)
dev = torch.device("cuda", 0)
# NPCD_TUNE_BATCHES="8,16": only those per-GPU batches (e.g. to add the shapes of a new switch to an existing file: tools/merge_tuned.py)
for B in tuple(int(b) for b in os.environ.get("NPCD_TUNE_BATCHES", "64,32,16,8").split(",")):      # per-GPU batches of the 1/2/4/8-GPU strong-scaling runs
    tr = bench.build_trainer(dev, B)
    coords, feats = bench.synthetic_batch(64, 0, 64 // B, dev)
    for _ in range(2):
        tr.step(coords, feats)
    torch.cuda.synchronize()
    del tr
    torch.cuda.empty_cache()
    print("tuned per-GPU batch", B, flush=True)
tun.write_file(out) if hasattr(tun, "write_file") else None
print("results ->", out)
