#!/usr/bin/env python3
"""Benchmark of the NPCD hot path on MI355X (contract: see the repo-level task description).

    python bench.py --gpus N --steps K --warmup W          (any N: for N > 1 without WORLD_SIZE it starts its own ranks
                                                            as a torch.distributed.run child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one denoiser training step of BASELINE.json configs[1]: zero_grad + forward (bf16
autocast) + backward + AdamW + EMA on 512 points x 128-d latents, width 1024 / 24 layers / 16 heads,
GLOBAL batch 64 (strong scaling: 64/N samples per GPU, gradients averaged with an RCCL all-reduce).
Rank 0 prints ONE JSON line.  `value` = denoiser train steps/s of the whole job; the renderer half of
the BASELINE metric (rays/s for 128x128 views, k=8, 128 depth samples) is measured in its own timed
region and reported under "render".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "neural-point-cloud-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

CFG = dict(coords_dim=3, feats_dim=128, num_points=512, width=1024, layers=24, heads=16, global_batch=64)
PEAK_BF16_TFLOPS = 2500.0     # dense MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0          # HBM3E, MI355X_MICROARCH.md


def denoiser_flops_per_sample():
    """fwd = L(24 n W^2 + 4 n^2 W) + 2 N 2(3+F) W + 16 W^2, n = N+1; step = 3x."""
    W, L, N, F_ = CFG["width"], CFG["layers"], CFG["num_points"], CFG["feats_dim"]
    n = N + 1
    fwd = L * (24 * n * W * W + 4 * n * n * W) + 2 * N * 2 * (3 + F_) * W + 16 * W * W
    return 3 * fwd


def build_trainer(device, per_rank_batch):
    from npcd.models.diffusion import DiffusionModel
    from npcd.train import DiffusionTrainer
    torch.manual_seed(1234)                       # identical weights on every rank
    model = DiffusionModel(CFG["coords_dim"], CFG["feats_dim"], CFG["num_points"], CFG["width"], CFG["layers"], CFG["heads"], True)
    torch.nn.init.normal_(model.denoiser.output_proj.weight, std=0.02)    # non-zero so grads are non-trivial
    model = model.to(device).train()
    # NPCD_BENCH_NO_SHARD=1: plain bucketed all-reduce instead of reduce-scatter + sharded optimizer + all-gather;
    # NPCD_COMM_BF16=1 (read by the trainer): gradient buckets travel as bf16.  A/B switches for the first multi-GPU runs.
    shard = False if os.environ.get("NPCD_BENCH_NO_SHARD") else None
    trainer = DiffusionTrainer(model, lr=7e-5, weight_decay=0.01, ema_decay=0.9999, dtype=torch.bfloat16, shard_optimizer=shard)
    # the number below is only valid for the native path: fused AdamW/EMA kernel, bf16 shadow, hand-written backbone node
    assert trainer.native, "DiffusionTrainer fell back to the torch optimizer: libnpcd_hip.so path not engaged"
    assert model.denoiser.backbone.fused_engine is not None, "fused HIP backbone (FusedBackboneEngine) not engaged"
    return trainer


def synthetic_batch(global_batch, rank, world, device):
    g = torch.Generator().manual_seed(42)
    coords = torch.randn(global_batch, CFG["coords_dim"], CFG["num_points"], generator=g)
    feats = torch.rand(global_batch, CFG["feats_dim"], CFG["num_points"], generator=g) * 2 - 1
    per = global_batch // world
    sl = slice(rank * per, (rank + 1) * per)
    return coords[sl].to(device), feats[sl].to(device)


def query_roofline(S, query_ms, launches, clock_ghz=None):
    """The neighbour-query kernel (grid_query_wave_kernel, csrc/geometry.hip) is neither HBM- nor MFMA-bound: its inputs are a
    6 KB cloud and a 1 MB voxel table, its work is integer / fp32 vector instructions.  Its bound is VECTOR ISSUE: a SIMD issues at
    most one vector wave-instruction per 2 cycles (MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles on a SIMD-32), 1,024 SIMDs.
    Wave-instruction counts per launch come from a committed rocprofv3 --pmc pass over the same scene (profiles/r*_render_sq_pmc.json:
    SQ_INSTS_VALU and friends; bench.py cannot collect counters); the duration is measured here with HIP events around the C call.
    The counter file records the sha256 of the kernel source it was taken on: when the in-tree csrc/geometry.hip differs (or the file
    predates that record) the counts belong to another binary -- `stale_counters` says so and `achieved` / `frac` are None."""
    import hashlib
    base = {"kernel": "grid_query_wave_kernel", "avg_ms": query_ms, "launches": launches, "bound": "valu-issue", "achieved": None}
    fp = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r6_render_sq_pmc.json", "r5_render_sq_pmc.json", "r4_render_sq_pmc.json", "r3_render_sq_pmc.json")) if os.path.exists(f)), None)
    if fp is None:
        return base
    doc = json.load(open(fp))
    pj = doc.get(f"S{S}", {}).get("grid_query_wave_kernel")
    if not pj:
        return base
    src = os.path.join(ROOT, "neural-point-cloud-diffusion_amd", "csrc", "geometry.hip")
    have = hashlib.sha256(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None
    want = doc.get("source_sha256", {}).get("geometry.hip")
    stale = want is None or have is None or want != have
    clock = clock_ghz or 2.4
    peak = 1024 * clock / 2.0                                      # G vector wave-instructions / s
    ach = None if stale else pj["insts_valu"] / (query_ms * 1e-3) / 1e9
    return {"kernel": "grid_query_wave_kernel", "bound": "valu-issue", "achieved": ach, "peak": peak, "unit": "G wave-instr/s",
            "frac": None if stale else ach / peak, "avg_ms": query_ms, "launches": launches, "stale_counters": stale,
            "vector_wave_instructions_per_launch": pj["insts_valu"], "all_wave_instructions_per_launch": pj.get("insts_all"),
            "counters_source": f"profiles/{os.path.basename(fp)} (rocprofv3 --pmc, same scene / pose / depth samples)",
            "peak_basis": f"1,024 SIMDs x {clock:.2f} GHz ({'sclk sampled during the timed loop' if clock_ghz else 'nominal'}) / 2 cycles per wave64 vector instruction"}


def bench_render(device, n_iters=100, burn_in=5):
    """pointnerf_evaluation.py:217-224 protocol: sync, t0, render, sync, t1; burn-in renders discarded (the reference drops 3
    and then renders 251 views per object: the timed region here is 100 back-to-back renders, i.e. the sustained rate).
    Depth samples per ray: 128 is the reference's code (pointnerf.py:184), 64 is what BASELINE.json configs[2] names: both."""
    from npcd.models.pointnerf import PointNeRF
    from npcd.utils import synthetic as orr
    coords, feats = orr.ellipsoid_cloud(512, 32, 1, seed=0)
    torch.manual_seed(0)
    net = PointNeRF(1, 32, 512, False)            # field MLPs: PyTorch default init (random weights)
    net = net.to(device).eval()
    extr = orr.look_at_pose(30, 20)[None, None].to(device)
    intr = orr.srn_intrinsics()[None, None].to(device)
    c, f = coords.to(device), feats.to(device)
    poses = torch.stack([orr.look_at_pose(30 + 45 * i, 20 - 5 * i) for i in range(8)])[None].to(device)
    intr8 = orr.srn_intrinsics()[None, None].expand(1, 8, 3, 3).contiguous().to(device)

    def timed(fn, iters, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters

    per_s = {}
    with torch.no_grad():
        for S in (128, 64):
            net.renderer.depth_resolution = S
            net.renderer.count_pairs = False
            dt = timed(lambda: net.render(c, f, extr, intr, 128), n_iters, burn_in)
            # throughput form: 8 views of the same object in one call (launch / sync overhead amortised)
            dt8 = timed(lambda: net.render(c, f, poses, intr8, 128), 20, 2)
            net.renderer.count_pairs = True        # workload statistics (FLOP accounting), outside the timed region
            out = net.render(c, f, extr, intr, 128)
            P, Q = int(out["num_shading_points"]), int(out["num_pairs"])
            flops = Q * 2 * (95 * 256 + 3 * 256 * 256) + P * 2 * (256 * 256 * 6 + 256 + 3 * 256)   # as executed (last agg layer on points)
            # the renderer's dominant kernels (shade_pairs_kernel + shade_points_kernel, f16 MFMA) on their own: HIP events on the
            # launch stream around the one C call that enqueues both, 20 renders, against the dense f16 MFMA peak
            from npcd.hip import render as hrender
            hrender.SHADE_EVENTS, hrender.QUERY_EVENTS = [], []
            net.renderer.count_pairs = False
            with ClockSampler(device.index or 0, period_s=0.002) as rclk:
                for _ in range(22):
                    net.render(c, f, extr, intr, 128)
                torch.cuda.synchronize()
            rsum = rclk.summary()
            rclock = rsum["sclk_mhz_min_median_max"][1] / 1e3 if rsum else None
            ev, hrender.SHADE_EVENTS = hrender.SHADE_EVENTS[2:], None
            qev, hrender.QUERY_EVENTS = hrender.QUERY_EVENTS[2:], None
            shade_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))
            query_ms = sum(a.elapsed_time(b) for a, b in qev) / max(1, len(qev))
            per_s[S] = {"rays_per_s": 128 * 128 / dt, "rays_per_s_8_views_per_call": 8 * 128 * 128 / dt8, "ms_per_view": dt * 1e3,
                        "shading_points": P, "pairs": Q, "mlp_tflops_whole_view": flops / dt / 1e12,
                        "roofline_shading": {"kernel": "shade_pairs_kernel + shade_points_kernel (csrc/shade.hip)", "bound": "mfma",
                                             "achieved": flops / (shade_ms * 1e-3) / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                             "frac": flops / (shade_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, "avg_ms": shade_ms,
                                             "algorithmic_flops_per_view": flops, "launches": len(ev),
                                             "note": "f16 operands, fp32 accumulation (same dense peak as bf16); FLOPs as executed: the linear last aggregator layer on points, not pairs"},
                        "roofline_query": query_roofline(S, query_ms, len(qev), rclock)}
    r = dict(per_s[128])
    # the same view with the field MLPs in the reference's numerics class (the evaluation scripts run them in fp32): per-pair layers
    # on the fp32-class matrix-core kernel (two bf16 halves per operand, three products), fifth layer + heads on its point-level sibling.
    # At both depth resolutions: 128 is the reference's code, 64 what BASELINE.json configs[2] names.
    for S in (128, 64):
        try:
            net.renderer.depth_resolution = S
            net.renderer.count_pairs = False
            with torch.no_grad():
                dt32 = timed(lambda: net.render(c, f, extr, intr, 128, mlp_dtype=torch.float32), 30, 3)
                a16 = net.render(c, f, extr, intr, 128)["channels"]
                a32 = net.render(c, f, extr, intr, 128, mlp_dtype=torch.float32)["channels"]
            entry = {"rays_per_s": 128 * 128 / dt32, "ms_per_view": dt32 * 1e3, "depth_samples": S,
                     "max_abs_pixel_difference_to_the_fp16_operand_render": float((a16 - a32).abs().max()),
                     "numerics": "PointNeRF.render(mlp_dtype=torch.float32): EMULATED fp32 -- per-pair layers with every operand as two bf16 halves "
                                 "(16 mantissa bits, fp32's exponent range; three matrix instructions per product, ~4e-6 relative per product; "
                                 "csrc/points_x2.hip pairs_x2_kernel), fifth layer and heads fused in the same numerics (points_x2_kernel); "
                                 "7e-7 max pixel error against the fp32 oracle in the tests; one host read of the point count per view"}
        except Exception as e:                      # noqa: BLE001
            entry = {"error": f"{type(e).__name__}: {e}"}
        if S == 128:
            r["fp32_class_shading"] = entry
        else:
            per_s[64]["fp32_class_shading"] = entry
    # continuity with rounds 1-2, which ran the FINE-grid reading of torch_knnquery (fewer shading points with a neighbour, i.e.
    # less work per view): the same view at grid_level "fine" (DESIGN.md section 3 has the evidence for the default)
    level = net.voxel_grid.grid_level
    other = "fine" if level == "scaled" else "scaled"
    net.voxel_grid.set_grid_level(other)
    net.renderer.depth_resolution = 128
    with torch.no_grad():
        net.renderer.count_pairs = False
        dt_o = timed(lambda: net.render(c, f, extr, intr, 128), 50, burn_in)
        net.renderer.count_pairs = True
        out_o = net.render(c, f, extr, intr, 128)
    net.voxel_grid.set_grid_level(level)
    r["grid_level"] = level
    r[f"grid_level_{other}"] = {"rays_per_s": 128 * 128 / dt_o, "ms_per_view": dt_o * 1e3, "shading_points": int(out_o["num_shading_points"]),
                                "pairs": int(out_o["num_pairs"])}
    r.update({"timed_renders": n_iters, "resolution": 128, "depth_samples": 128, "k": 8,
              "depth_samples_64": per_s[64],
              "note": "mlp_tflops_whole_view = shading FLOPs / whole-view wall time (all kernels of the view), not a per-kernel roofline"})
    return r


def bench_stage1(device, n_iters=20, burn_in=6, mlp_dtype=None):
    """Stage-1 (PointNeRF autodecoder) training step at the reference's configuration (configs/npcd_srncars.yaml:13-16,
    data/srn.py:45: 8 objects x 50 views per step, 112 random rays per view, 128 depth samples, Adam lr 1e-3): secondary
    figure for SURVEY §8(f) rank 2.  Synthetic clouds / poses / target images.  20 timed iterations, per-iteration spread
    reported (HIP events around every step)."""
    from npcd.models import NPCD
    from npcd.train import PointNeRFTrainer
    from npcd.utils import synthetic as orr
    B, T, N, F_, res = 8, 50, 512, 32, 128
    torch.manual_seed(0)
    net = NPCD(n_obj=B, coords_dim=3, feats_dim=F_, num_points=N, use_view_dir=False, width=64, layers=1, heads=1, pointnerf_only=True).to(device)
    coords, _ = orr.ellipsoid_cloud(N, F_, B, seed=0)
    net.pointnerf.set_all_coords(coords.to(device))
    extr = torch.stack([orr.look_at_pose(7.2 * i, 20 - 0.5 * i) for i in range(T)])[None].expand(B, -1, -1, -1).contiguous().to(device)
    intr = orr.srn_intrinsics()[None, None].expand(B, T, 3, 3).contiguous().to(device)
    sample = {"images": torch.rand(B, T, 3, res, res, device=device), "intrinsics": intr, "extrinsics": extr,
              "obj_idx": torch.arange(B, device=device)}
    tr = PointNeRFTrainer(net, mlp_dtype=mlp_dtype)
    # the rays are redrawn every step, so the shading points P and (point, neighbour) pairs Q of a step vary: recorded per step
    # (device scalars, read after the loop) to tell a data-dependent spread from host round trips
    import npcd.models.pointnerf.train_path as tpath
    counts, orig_render = [], tpath.render_train

    def counted(*a, **k):
        out = orig_render(*a, **k)
        counts.append((out["num_shading_points"], out["num_pairs"]))
        return out
    tpath.render_train = counted
    try:
        # (the pair count of a step varies by 2 x with its random rays: a fresh allocator pool + a longer burn-in keep first-time
        # allocations of a larger pair list -- one 156-ms iteration in a round-5 run -- out of the timed region)
        torch.cuda.empty_cache()
        for _ in range(burn_in):
            tr.step(sample)
        torch.cuda.synchronize()
        counts.clear()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_iters + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(n_iters):
            loss, _ = tr.step(sample)
            marks[i + 1].record()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_iters
    finally:
        tpath.render_train = orig_render      # a failing leg must not leave the counter installed for the later legs
    raw = [marks[i].elapsed_time(marks[i + 1]) for i in range(n_iters)]
    per_it = sorted(raw)
    spread = {}
    if len(counts) == n_iters:
        import numpy as np
        q = np.array([float(c[1]) for c in counts])
        ms = np.array(raw)
        a, b = np.polyfit(q, ms, 1)
        spread = {"pairs_min_median_max": [float(q.min()), float(np.median(q)), float(q.max())],
                  "shading_points_min_median_max": [float(min(c[0] for c in counts)), float(np.median([c[0] for c in counts])), float(max(c[0] for c in counts))],
                  "corr_ms_vs_pairs": float(np.corrcoef(ms, q)[0, 1]), "fit_ms": {"fixed": float(b), "per_million_pairs": float(a * 1e6), "residual_std": float(np.std(ms - (a * q + b)))},
                  "note": "the step's random rays decide its pairs: the per-iteration spread follows the pair count, not host round trips"}
    return {"steps_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "ms_per_step_min_median_max": [per_it[0], per_it[n_iters // 2], per_it[-1]], "per_iteration_spread": spread,
            "timed_iterations": n_iters, "objects": B, "views_per_object": T, "rays_per_view": net.pointnerf.renderer.ray_subsamples,
            "loss": float(loss), "differentiable_part": tr.describe()}


class ClockSampler:
    """Shader-clock / board-power datum for the timed region, so that a reader can tell a slow box from slow code: a host thread
    reads the amdgpu sysfs files of the GPU this rank runs on every 20 ms while the timed loop runs (no GPU work, no sync):
    `pp_dpm_sclk` (the level marked '*') and hwmon `power1_average` / `power1_input`.  sysfs reports the DPM state the
    firmware holds, which under an MFMA-dense load reads up to ~10 % above the in-kernel clock (MI355X_MICROARCH.md, 'DVFS
    give-back' item 6): a box-to-box comparison datum, not a cycle count."""

    def __init__(self, local_rank=0, period_s=0.02):
        import glob
        import threading
        self.period = period_s
        self.sclk, self.power = [], []
        self._stop = threading.Event()
        self._thread = None
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))
        self.sclk_file = None
        # the DRM card of THIS rank's device, by PCI address (a box may expose one GPU of several: card0 is then somebody else's)
        try:
            pr = torch.cuda.get_device_properties(local_rank)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for c in cards:
                if os.path.basename(os.path.realpath(os.path.dirname(c))).startswith(want):
                    self.sclk_file = c
        except (AttributeError, RuntimeError, AssertionError):
            pass
        self.matched_by_pci = self.sclk_file is not None
        if self.sclk_file is None and cards:
            self.sclk_file = cards[min(local_rank, len(cards) - 1)]
        self.power_file = None
        if self.sclk_file:
            dev = os.path.dirname(self.sclk_file)
            for nm in ("power1_average", "power1_input"):
                hits = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*", nm)))
                if hits:
                    self.power_file = hits[0]
                    break

    def _read(self):
        try:
            with open(self.sclk_file) as fh:
                for line in fh:
                    if "*" in line:
                        self.sclk.append(float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
                        break
            if self.power_file:
                with open(self.power_file) as fh:
                    self.power.append(float(fh.read().strip()) / 1e6)
        except (OSError, ValueError, IndexError):
            pass

    def _loop(self):
        while not self._stop.is_set():
            self._read()
            self._stop.wait(self.period)

    def __enter__(self):
        import threading
        if self.sclk_file:
            self._thread = threading.Thread(target=self._loop, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=1.0)

    def summary(self):
        if not self.sclk:
            return None
        q = lambda v: [min(v), sorted(v)[len(v) // 2], max(v)]
        out = {"sclk_mhz_min_median_max": q(self.sclk), "samples": len(self.sclk), "source": self.sclk_file,
               "device_matched_by_pci_address": self.matched_by_pci,
               "note": "amdgpu sysfs DPM state sampled every 20 ms by a host thread during the timed loop; reads above the in-kernel clock under MFMA load"}
        if self.power:
            out["board_power_w_min_median_max"] = q(self.power)
        return out


def bench_sample_and_render(device, clouds=4, fp32_class=False):
    """The generate -> render loop of the reference's DiffusionEvaluation (npcd/eval/diffusion_evaluation.py:146-183; FID/KID itself
    needs the Inception assets, which are not in this environment): `clouds` point clouds from the full 1000-step sampler at the
    benchmark's model size (bf16 autocast, the forward-only fused backbone), each rendered at 128 x 128 from the 251 bundled SRN-cars
    test poses, 8 poses per render call.  Random-init weights: the clouds are noise-shaped, the work per cloud is the real one."""
    from npcd.eval import load_test_poses, sample_and_render
    from npcd.models import NPCD
    torch.manual_seed(0)
    net = NPCD(n_obj=1, coords_dim=3, feats_dim=32, num_points=CFG["num_points"], use_view_dir=False, width=CFG["width"], layers=CFG["layers"],
               heads=CFG["heads"]).to(device).eval()
    with torch.no_grad():          # a plausible normaliser: clouds inside the unit cube, so that the rays hit something
        net.diffusion.coords_normalization.min.fill_(-2.5); net.diffusion.coords_normalization.max.fill_(2.5)
        net.diffusion.coords_normalization.scale.fill_(0.2)
        net.diffusion.feats_normalization.min.fill_(-1.0); net.diffusion.feats_normalization.max.fill_(1.0)
    poses, intr = load_test_poses("srncars")
    if fp32_class:
        # the reference's numerics class end to end (--eval-loop-fp32-class; ~10 s): fp32 sampling with split-operand GEMMs in the backbone,
        # fp32-class shading (per-pair layers as two bf16 halves per operand, heads fp32)
        r = sample_and_render(net, poses, intr, num_samples=clouds, generate_batch_size=clouds, render_batch_size=8, resolution=128,
                              dtype="fp32_class", use_graph=False, render_mlp_dtype=torch.float32)
        r["note"] = ("reference protocol in the reference's numerics class: DiffusionModel.generate(dtype='fp32_class') (fp32 everywhere, the backbone's Linear layers as "
                     "split-operand bf16 GEMMs: eps 8e-7 from the fp32 path) and PointNeRF.render(mlp_dtype=torch.float32)")
        return r
    r = sample_and_render(net, poses, intr, num_samples=clouds, generate_batch_size=clouds, render_batch_size=8, resolution=128,
                          dtype=torch.bfloat16, use_graph=True)
    r["note"] = ("reference protocol: generate_batch_size clouds per sampler call, 251 poses per cloud, render_batch_size 8; the sampler's denoiser "
                 "under bf16 autocast (opt-in of DiffusionModel.generate; the reference samples in fp32), its reverse step replayed from a HIP graph "
                 "(at batch 4 a step is ~240 launches: 3.6 ms replayed against 4.3 ms eager); feats_dim 32 (the PointNeRF latent width)")
    return r


def bench_sampler(device, batch=16, steps=8, graph_steps=100):
    """one DDPM reverse step (denoiser forward + posterior update, diffusion_model.py:108-133 /
    gaussian_diffusion.py p_sample) at the benchmark's model size.  Lead figure: the reference's numerics (generate() runs the
    denoiser in fp32 with the einsum attention) on the fp32 matrix-instruction attention kernels; beside it the bf16-autocast
    opt-in through the forward-only fused backbone, eager and replayed from a HIP graph."""
    import contextlib
    from npcd.models.diffusion import DiffusionModel
    torch.manual_seed(0)
    m = DiffusionModel(3, CFG["feats_dim"], CFG["num_points"], CFG["width"], CFG["layers"], CFG["heads"], True).to(device).eval()
    c = torch.randn(batch, 3, CFG["num_points"], device=device)
    f = torch.randn(batch, CFG["feats_dim"], CFG["num_points"], device=device)
    dp = m.diffusion_process

    def run(n, ctx):
        cc, ff = c, f
        with torch.no_grad(), ctx:
            for i in range(dp.num_timesteps - 1, dp.num_timesteps - 1 - n, -1):
                t = torch.full((batch,), i, device=device, dtype=torch.long)
                cc, _, ff, _ = dp.p_sample(m.denoiser, cc, ff, t, None, None)
        return cc

    out = {"batch": batch, "model": "benchmark denoiser (width %d, %d layers)" % (CFG["width"], CFG["layers"])}
    for key, ctx in (("fp32_reference_numerics", contextlib.nullcontext()), ("fp32_class_split_operands", contextlib.nullcontext()),
                     ("bf16_autocast_opt_in", torch.autocast("cuda", dtype=torch.bfloat16))):
        # fp32_class_split_operands (generate(dtype="fp32_class"), round 5): fp32 everywhere except that each Linear layer of the backbone is ONE
        # bf16 library GEMM over the three cross products of split operands (two bf16 halves per operand, fp32 accumulation): 3e-6 relative per
        # product, eps within 2e-5 of the fp32 path (tests/test_gpu_attention.py::test_fp32_class_sampler_forward_matches_the_fp32_path)
        m.denoiser.backbone.fp32_class = key == "fp32_class_split_operands"
        run(2, ctx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run(steps, ctx)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[key] = {"ms_per_reverse_step": dt * 1e3, "clouds_per_s_at_1000_steps": batch / (1000 * dt), "finite": bool(torch.isfinite(res).all())}
    m.denoiser.backbone.fp32_class = False
    saved = dp.num_timesteps
    try:
        def loop(n, graph):
            dp.num_timesteps = n
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                return dp.p_sample_loop(m.denoiser, c, f, (-3.0, 3.0), (-1.0, 1.0), use_graph=graph)
        loop(4, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = loop(graph_steps, True)[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / graph_steps
        out["bf16_fused_hip_graph"] = {"ms_per_reverse_step": dt * 1e3, "steps": graph_steps, "finite": bool(torch.isfinite(res).all()),
                                       "note": "capture amortised over the steps"}
    finally:
        dp.num_timesteps = saved
    return out


def bench_cfg5(device, steps=20, warmup=2, round_steps=5):
    """BASELINE configs[4] as ONE rank of its 8-GPU job computes it (no communication): 2048 points x 256-d latents, 8-layer
    denoiser (width 1024 / 16 heads: BASELINE leaves W / H open, SURVEY 8(d) takes the yaml's), sequence 2049, per-GPU batch
    32 of the global 256 -- the training step with the attention forward on the bf16 kernel and on the fp8 (e4m3, block-scaled
    MFMA) kernel (backward: bf16 kernels both ways).  tests/test_gpu_fused.py::test_stress_config_step_at_per_gpu_batch_32_...
    holds the parity bars of both modes against the fp32 oracle.
    The two legs are INTERLEAVED (rounds of `round_steps` steps, bf16 / fp8 / bf16 / ...; `steps` timed steps per leg) on two
    trainers with identical weights: clock and thermal drift of the box lands on both legs alike (with the legs
    run one after the other the same bf16 dK/dV kernel read 1.51 ms in one leg and 1.11 ms in the other)."""
    from npcd.hip import attention as hattn
    from npcd.models.diffusion import DiffusionModel
    from npcd.train import DiffusionTrainer
    B, N, F_, W, L, H = 32, 2048, 256, 1024, 8, 16
    g = torch.Generator().manual_seed(7)
    c = torch.randn(B, 3, N, generator=g).to(device)
    f = (torch.rand(B, F_, N, generator=g) * 2 - 1).to(device)
    n = N + 1
    flops = 3 * B * (L * (24 * n * W * W + 4 * n * n * W) + 2 * N * 2 * (3 + F_) * W + 16 * W * W)
    out = {"config": f"{N} points x {F_}-d latents, {L} layers, width {W}, {H} heads (seq {n}), per-GPU batch {B} of the global 256, "
                     f"bf16 autocast, one rank's step without communication", "timed_steps": steps,
           "schedule": f"legs interleaved in rounds of {round_steps} steps", "algorithmic_flops_per_step": flops}
    saved = hattn.FWD_FP8
    modes = ("bf16", "fp8")
    try:
        trainers, secs, events, loss = {}, {m: [] for m in modes}, {m: {k: [] for k in hattn.KERNEL_TAGS} for m in modes}, {}
        for mode in modes:
            hattn.FWD_FP8 = mode == "fp8"
            torch.manual_seed(0)
            m = DiffusionModel(3, F_, N, W, L, H, True)
            torch.nn.init.normal_(m.denoiser.output_proj.weight, std=0.02)
            tr = DiffusionTrainer(m.to(device).train())
            assert tr.native and m.denoiser.backbone.fused_engine is not None
            torch.manual_seed(3)
            for _ in range(warmup):
                tr.step(c, f)
            trainers[mode] = tr
        torch.cuda.synchronize()
        for _ in range((steps + round_steps - 1) // round_steps):
            for mode in modes:
                hattn.FWD_FP8 = mode == "fp8"
                hattn.KERNEL_EVENTS = events[mode]
                t0 = time.perf_counter()
                for _ in range(round_steps):
                    loss[mode], _ = trainers[mode].step(c, f)
                torch.cuda.synchronize()
                secs[mode].append((time.perf_counter() - t0) / round_steps)
                hattn.KERNEL_EVENTS = None
        U = 2 * B * H * n * n * 64
        for mode in modes:
            dt = sum(secs[mode]) / len(secs[mode])
            kms = {k: sum(a.elapsed_time(b) for a, b in v) / len(v) for k, v in events[mode].items() if v}
            out[f"attention_forward_{mode}"] = {"ms_per_step": dt * 1e3, "ms_per_step_by_round": [x * 1e3 for x in secs[mode]],
                                                "step_tflops": flops / dt / 1e12, "loss": float(loss[mode]),
                                                "attention_kernel_ms": kms,
                                                "attention_fwd_tflops": 2 * U / (kms["fwd"] * 1e-3) / 1e12 if "fwd" in kms else None}
        out["attention_forward_fp8"]["status"] = ("opt-in (NPCD_ATTN_FP8=1), measured slower than or equal to the bf16 forward in the step at head_dim 64; "
                                                  "forward only, backward on the bf16 kernels; output rel-L2 5e-2 against 2e-3 (DESIGN.md section 8)")
        out["fp8_over_bf16_step_time"] = out["attention_forward_fp8"]["ms_per_step"] / out["attention_forward_bf16"]["ms_per_step"]
        for tr in trainers.values():
            tr.close()
        del trainers
        torch.cuda.empty_cache()
    finally:
        hattn.FWD_FP8, hattn.KERNEL_EVENTS = saved, None
    return out


def host_cpu_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    return model, os.cpu_count() or 1, affinity


def physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo; None when the file does not say."""
    cores, phys = set(), None
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                key = line.split(":", 1)[0].strip().lower()
                if key == "physical id":
                    phys = line.split(":", 1)[1].strip()
                elif key == "core id":
                    cores.add((phys, line.split(":", 1)[1].strip()))
    except OSError:
        return None
    return len(cores) or None


PARITY_PARAMS = ("input_proj.weight", "time_embed.c_fc.weight", "backbone.resblocks.0.attn.c_qkv.weight", "backbone.resblocks.0.ln_1.weight",
                 "backbone.resblocks.11.mlp.c_fc.weight", "backbone.resblocks.23.attn.c_proj.weight", "backbone.resblocks.23.mlp.c_proj.bias",
                 "ln_post.weight", "output_proj.weight")


def cpu_baseline():
    """The CPU oracle (oracle/: an fp32 PyTorch restatement of the reference, pinned by the golden fixtures; `kind: port`) timed on
    this box's host cores on a BOUNDED sample of the benchmark workload: full-width denoiser train steps (W 1024 / L 24 / H 16,
    forward + backward + AdamW) on B = 4 of the 64 samples of a step, at ONE thread count chosen once: 32 threads (or every usable
    CPU when there are fewer; NPCD_CPU_BASELINE_THREADS overrides).  Round 4 timed 32 threads AND one thread per physical core on
    the pool's 2 x 64-core EPYC 9575F hosts: 4.0-6.2 s against 14.5-17.7 s per step (profiles/r4_bench.json
    `cpu_baseline.timed_step_seconds_by_threads`, DESIGN.md section 7 for why) -- the slower count only doubled the bench's wall
    time.  One untimed warm-up step and three timed steps; `value` is the best step scaled by 64 / 4, `cores` the threads used
    -- plus one 128 x 128 render with the voxel-grid semantics for the rays/s half.
    The warm-up step's loss and gradient norms are returned too: bench.py feeds the same samples, timesteps and noise through
    the GPU trainer and reports the differences (`parity_full_width`)."""
    from oracle import denoiser as od, diffusion as odf, renderer as orr
    model_name, logical, affinity = host_cpu_info()
    phys = physical_cores()
    candidates = [max(1, min(affinity, int(os.environ.get("NPCD_CPU_BASELINE_THREADS", "32"))))]
    torch.set_num_threads(candidates[0])
    B = 4
    params = od.init_params(CFG["coords_dim"], CFG["feats_dim"], CFG["width"], CFG["layers"], CFG["heads"], seed=0)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.AdamW(list(leaves.values()), lr=7e-5, weight_decay=0.01)
    g = torch.Generator().manual_seed(42)
    c0 = torch.randn(B, 3, CFG["num_points"], generator=g)
    f0 = torch.rand(B, CFG["feats_dim"], CFG["num_points"], generator=g) * 2 - 1
    tab = odf.schedule_tables()
    t = torch.randint(0, 1000, (B,), generator=g)
    cn, fn = torch.randn(c0.shape, generator=g), torch.randn(f0.shape, generator=g)

    def fwd_bwd():
        opt.zero_grad()
        loss, _, _ = odf.p_losses(tab, lambda c, f, tt: od.denoiser_forward(leaves, c, f, tt, CFG["heads"]), c0, f0, t, cn, fn)
        loss.backward()
        return loss

    loss0 = fwd_bwd()                             # untimed: thread-pool / allocator warm-up; also the parity reference
    ref = {"loss": float(loss0.detach()), "grad_norm": {k: float(leaves[k].grad.norm()) for k in PARITY_PARAMS},
           "inputs": (params, c0, f0, t, cn, fn)}
    opt.step()
    by_threads = {}
    for i, th in enumerate(candidates):
        torch.set_num_threads(th)
        if i:
            fwd_bwd()                             # this thread count's warm-up
            opt.step()
        samples = []
        for _ in range(3):
            t0 = time.perf_counter()
            fwd_bwd()
            opt.step()
            samples.append(time.perf_counter() - t0)
        by_threads[th] = samples
    threads = min(by_threads, key=lambda th: min(by_threads[th]))
    torch.set_num_threads(threads)
    samples = by_threads[threads]
    dt = min(samples)
    steps_per_s = (B / dt) / CFG["global_batch"]
    fp = orr.init_field_params(32, seed=0)
    coords, feats = orr.synthetic_cloud(512, 32, 1, seed=0)
    res = 128
    t0 = time.perf_counter()
    with torch.no_grad():
        orr.render(fp, coords, feats, orr.look_at_pose(30, 20)[None, None], orr.srn_intrinsics()[None, None], res=res)
    dtr = time.perf_counter() - t0
    out = {"value": steps_per_s, "unit": "steps/s", "cores": threads, "kind": "port",
           "cpu_model": model_name, "host_logical_cpus": logical, "affinity_cpus": affinity,
           "sample": f"oracle (fp32 PyTorch CPU restatement of the reference) on {threads} threads of '{model_name}' "
                     f"({logical} logical CPUs, {affinity} usable): full-width denoiser train step (fwd + bwd + AdamW, "
                     f"{CFG['layers']} blocks) at B = {B} of {CFG['global_batch']}: 1 warm-up + 3 timed steps "
                     f"({'; '.join(f'{th} threads: ' + ', '.join(f'{x:.2f}' for x in v) + ' s' for th, v in by_threads.items())}), "
                     f"best step x {CFG['global_batch'] // B} (batch); "
                     f"render: one {res} x {res} view (grid semantics, 128 depth samples) = {dtr:.1f} s",
           "physical_cores": phys, "timed_step_seconds_by_threads": {str(k): v for k, v in by_threads.items()},
           "timed_step_seconds": samples, "render_rays_per_s": res * res / dtr}
    return out, ref


def parity_full_width(device, ref):
    """The full-width model (W 1024 / L 24 / H 16, 310.8 M parameters) on the GPU trainer against the CPU oracle on the SAME
    weights, samples, timesteps and noise (those of the cpu_baseline leg): loss and a few parameter-gradient norms.  The
    oracle is the checker here, the timed region is long over.  Bars: loss within 2e-2 relative, gradient norms within 5e-2
    (bf16 GEMM operands and attention against fp32)."""
    from npcd.models.diffusion import DiffusionModel
    from npcd.train import DiffusionTrainer
    params, c0, f0, t, cn, fn = ref["inputs"]
    model = DiffusionModel(CFG["coords_dim"], CFG["feats_dim"], CFG["num_points"], CFG["width"], CFG["layers"], CFG["heads"], True)
    model.denoiser.load_state_dict(params)
    model = model.to(device).train()
    tr = DiffusionTrainer(model, dtype=torch.bfloat16)
    assert tr.native and model.denoiser.backbone.fused_engine is not None
    tr.flat.zero_grad()
    tr.reducer.start_step()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _, _ = model.compute_loss(c0.to(device), f0.to(device), t=t.to(device), coords_noise=cn.to(device), feats_noise=fn.to(device))
    loss.backward()
    torch.cuda.synchronize()
    named = dict(model.denoiser.named_parameters())
    gn = {k: float(named[k].grad.norm()) for k in PARITY_PARAMS}
    rel = {k: abs(gn[k] - ref["grad_norm"][k]) / max(ref["grad_norm"][k], 1e-30) for k in PARITY_PARAMS}
    loss_gpu = float(loss.detach())
    loss_rel = abs(loss_gpu - ref["loss"]) / abs(ref["loss"])
    return {"loss_gpu": loss_gpu, "loss_oracle": ref["loss"], "loss_rel_diff": loss_rel,
            "grad_norm_gpu": gn, "grad_norm_oracle": ref["grad_norm"], "grad_norm_rel_diff": rel,
            "bars": {"loss_rel": 2e-2, "grad_norm_rel": 5e-2},
            "ok": bool(loss_rel < 2e-2 and max(rel.values()) < 5e-2),
            "config": "W 1024 / L 24 / H 16, B = 4, same weights / samples / t / noise on both sides; GPU = fused engine under bf16 autocast"}


def strong_scaling_proxy(trainer, coords, feats, steps=8, warmup=3):
    """One-GPU proxy of the strong-scaling curve (global batch 64 fixed): the step at per-GPU batch 64 / 32 / 16 / 8, i.e. what
    one rank of a 1 / 2 / 4 / 8-GPU job computes, WITHOUT communication.  The optimizer pass is timed on its own: with the
    sharded optimizer a rank runs 1/ranks of it, so `ms_per_rank_step` = step - optimizer * (1 - 1/ranks).  Upper bound of the
    speed-up = ms(64) / ms_per_rank_step(b)."""
    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        trainer._adamw_range(0, trainer.flat.numel, zero_grad=False)
    e1.record()
    torch.cuda.synchronize()
    opt_ms = e0.elapsed_time(e1) / 5
    trainer.flat.grad.zero_()
    base = base_in = None
    for b in (64, 32, 16, 8):
        c, f = coords[:b], feats[:b]
        if c.shape[0] < b:
            continue
        for _ in range(warmup):
            trainer.step(c, f)
        torch.cuda.synchronize()
        # the optimizer pass INSIDE the step, between HIP events (it runs on fresher caches than the stand-alone passes above)
        opt_events, orig_adamw = [], trainer._adamw_range

        def timed_adamw(*a, **k):
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            orig_adamw(*a, **k)
            ev[1].record()
            opt_events.append(ev)
        trainer._adamw_range = timed_adamw
        try:
            t0 = time.perf_counter()
            for _ in range(steps):
                trainer.step(c, f)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
        finally:
            del trainer._adamw_range          # (back to the class's method)
        opt_in_step = sum(a.elapsed_time(e) for a, e in opt_events) / max(1, len(opt_events))
        ranks = 64 // b
        per_rank = ms - opt_ms * (1 - 1 / ranks)
        base = base or per_rank
        # the HOST's share: two steps submitted into an empty queue (no back-pressure from a full ring), best of three -- what the
        # step costs the submitting thread; the step is GPU-bound where this is below ms_per_step_one_gpu
        host = []
        for _ in range(3):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            trainer.step(c, f)
            trainer.step(c, f)
            host.append((time.perf_counter() - h0) / 2 * 1e3)
        torch.cuda.synchronize()
        per_rank_in = ms - opt_in_step * (1 - 1 / ranks)
        base_in = base_in or per_rank_in
        out[str(b)] = {"ranks": ranks, "ms_per_step_one_gpu": ms, "ms_per_rank_step": per_rank, "speedup_bound": base / per_rank,
                       "host_submit_ms_per_step": min(host),
                       # the same bound with the optimizer pass as timed INSIDE this step (HIP events) instead of stand-alone: the
                       # stand-alone pass varies 2.2-2.6 ms by box and is slower than the in-step one, which flatters the bound
                       "optimizer_pass_in_step_ms": opt_in_step, "ms_per_rank_step_in_step_basis": per_rank_in,
                       "speedup_bound_in_step_basis": base_in / per_rank_in}
    out["optimizer_pass_ms"] = opt_ms
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child process and relay its output: stdout
    lines pass through unchanged (exactly one JSON line, printed by rank 0), stderr is inherited.  Called BEFORE any GPU call
    of this process; returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts (RCCL needs it across processes)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench.py: no WORLD_SIZE in the environment, starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    try:
        for line in child.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
        return child.wait()
    except BaseException:
        child.terminate()
        try:
            child.wait(timeout=30)
        except Exception:           # noqa: BLE001
            child.kill()
        raise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-render", action="store_true")
    ap.add_argument("--no-proxy", action="store_true", help="skip the one-GPU strong-scaling proxy (per-GPU batch 64/32/16/8)")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the BASELINE configs[4] step (2048 points x 256-d, per-GPU batch 32)")
    ap.add_argument("--no-sampler", action="store_true", help="skip the DDPM reverse-step timing (SURVEY 8(f) rank 1)")
    ap.add_argument("--eval-loop-fp32-class", action="store_true", help="also run the generate -> render evaluation loop in the reference's numerics class (~10 s more)")
    ap.add_argument("--rccl-algo", default=None, help="NCCL_ALGO for RCCL (e.g. Ring, Tree); default: RCCL's own choice")
    ap.add_argument("--rccl-proto", default=None, help="NCCL_PROTO for RCCL (e.g. Simple, LL, LL128); default: RCCL's own choice")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (a CHILD process, never an exec; nothing in this
        # process has touched the GPU yet), relay rank 0's JSON line and leave with the launcher's return code
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or unset WORLD_SIZE and bench.py starts its own ranks)")
    # NPCD_BENCH_DRYRUN_ONE_GPU=1 (tests only): all ranks share cuda:0 and talk through gloo, so that the multi-rank code path
    # of this script can be exercised on a one-GPU box (RCCL refuses two ranks on one device)
    dryrun = bool(os.environ.get("NPCD_BENCH_DRYRUN_ONE_GPU"))
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if args.rccl_algo:
        os.environ["NCCL_ALGO"] = args.rccl_algo            # (read by RCCL at communicator creation)
    if args.rccl_proto:
        os.environ["NCCL_PROTO"] = args.rccl_proto
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dryrun:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)     # "nccl" is RCCL on ROCm
    if CFG["global_batch"] % world:
        raise SystemExit("global batch 64 must be divisible by the number of GPUs")
    per = CFG["global_batch"] // world

    # hipBLASLt / rocBLAS solution choices recorded once with PyTorch TunableOp (tools/tune_gemms.py) for the GEMM shapes
    # of this step; loaded with tuning DISABLED (shapes not in the file use the library default).  NPCD_NO_TUNED_GEMM=1 skips it.
    tuned = os.environ.get("NPCD_TUNED_CSV") or os.path.join(ROOT, "profiles", "tunableop_gfx950.csv")      # (the variable: A/B of tuning recipes)
    use_tuned = os.path.exists(tuned) and not os.environ.get("NPCD_NO_TUNED_GEMM")
    if use_tuned:
        import torch.cuda.tunable as tun
        tun.enable(True)
        tun.tuning_enable(False)
        tun.record_untuned_enable(False) if hasattr(tun, "record_untuned_enable") else None
        tun.set_filename(os.path.join("/tmp", f"npcd_tunableop_unused_{rank}.csv"))    # never overwrite the committed file
        use_tuned = bool(tun.read_file(tuned))

    from npcd.hip import attention as hattn
    from npcd.hip import elementwise as hew
    trainer = build_trainer(device, per)
    coords, feats = synthetic_batch(CFG["global_batch"], rank, world, device)
    # weights are identical on every rank (seed above); the per-step draws (timesteps, noise) are not: seed + rank, as a
    # DistributedSampler-style job would
    torch.manual_seed(42 + rank)
    torch.cuda.manual_seed(42 + rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(coords, feats)
    barrier()
    hattn.KERNEL_EVENTS = {k: [] for k in hattn.KERNEL_TAGS}
    hew.KERNEL_EVENTS = {"add_ln_fwd": [], "ln_bwd": [], "gelu_fwd": [], "gelu_bwd": []}
    with ClockSampler(local_rank) as clocks:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, _ = trainer.step(coords, feats)
        barrier()
        elapsed = time.perf_counter() - t0
    events, hattn.KERNEL_EVENTS = hattn.KERNEL_EVENTS, None
    ew_events, hew.KERNEL_EVENTS = hew.KERNEL_EVENTS, None
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    assert torch.isfinite(loss), "training diverged"

    # ---- attention roofline (MFMA-bound), on the ALGORITHMIC FLOPs of SURVEY 8(d) ---------------------------------------
    # unit U = one B x H x n x n x d product = 2 B H n^2 d FLOP.  Forward = 2 U (QK^T, PV).  Backward = 5 U (S, dP, dV, dK, dQ)
    # however many products the kernels execute: a two-pass backward (dq pass: S, dP, dQ; dk/dv pass: S, dP, dV, dK)
    # EXECUTES 7 U -- the two recomputed products are not credited.  "Launch" of the backward = the kernel(s) of one
    # attention backward; durations are HIP-event averages over the timed region, on the launch stream.
    kern_ms = {k: (sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in events.items() if v}
    n = CFG["num_points"] + 1
    U = 2 * per * CFG["heads"] * n * n * 64
    bwd_tags = [k for k in kern_ms if k != "fwd"]
    bwd_ms = sum(kern_ms[k] for k in bwd_tags)
    executed = {"fwd": 2 * U, "dq": 3 * U, "dkdv": 4 * U, "bwd": 5 * U}
    tf = lambda fl, ms: fl / (ms * 1e-3) / 1e12
    bwd_tf, fwd_tf, all_tf = tf(5 * U, bwd_ms), tf(2 * U, kern_ms["fwd"]), tf(7 * U, bwd_ms + kern_ms["fwd"])
    knames = {"fwd": "attn_fwd_kernel", "dq": "attn_bwd_dq_kernel", "dkdv": "attn_bwd_dkdv_kernel", "bwd": "attn_bwd_kernel"}
    # HBM traffic and MFMA-busy: rocprofv3 PMC passes on these exact kernels and this shape (tools/make_traffic_json.py,
    # tools/pmc_summary.py); bench.py itself cannot collect PMC counters
    def newest(*names, sources=()):
        """The newest committed counter file of the given names + whether it is STALE: the file records the sha256 of the kernel
        sources it was taken on (`_meta.source_sha256`, tools/source_hashes.py); when a source differs in the tree -- or the file
        predates that record -- its numbers belong to another binary and are reported as such, not as this build's."""
        import hashlib
        for nm in names:
            fp = os.path.join(ROOT, "profiles", nm)
            if os.path.exists(fp):
                doc = json.load(open(fp))
                want = doc.get("_meta", {}).get("source_sha256", {})
                stale = not want
                for f in sources:
                    src = os.path.join(ROOT, "neural-point-cloud-diffusion_amd", "csrc", f)
                    have = hashlib.sha256(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None
                    stale = stale or want.get(f) != have
                return nm, doc, stale
        return None, {}, True
    attn_src = ("attention.hip", "common.h")
    tname, tjson, tstale = newest("r6_attention_hbm_traffic_pmc.json", "r5_attention_hbm_traffic_pmc.json", "r4_attention_hbm_traffic_pmc.json", "r3_attention_hbm_traffic_pmc.json", sources=attn_src)
    sname, sjson, sstale = newest("r6_attention_sq_pmc.json", "r5_attention_sq_pmc.json", "r4_attention_sq_pmc.json", "r3_attention_sq_pmc.json", sources=attn_src)
    traffic = None
    # sequences of 128 j + 1 tokens: the last token's three gradient rows are finished by a small third kernel, launched (and
    # timed here) with the dK/dV pass
    edge = "dkdv" in bwd_tags and (n & 127) == 1 and n > 128
    if per == 64 and tjson and not tstale and all(knames[k] in tjson for k in bwd_tags):
        traffic = sum(tjson[knames[k]]["hbm_bytes"] for k in bwd_tags) + (tjson.get("attn_bwd_edge_kernel", {}).get("hbm_bytes", 0) if edge else 0)
    roofline = {
        "kernel": " + ".join(knames[k] for k in bwd_tags) + (" + attn_bwd_edge_kernel" if edge else "") + " (one attention backward)",
        "bound": "mfma", "achieved": bwd_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": bwd_tf / PEAK_BF16_TFLOPS,
        "traffic": traffic, "traffic_source": f"profiles/{tname} (rocprofv3 --pmc, bytes per launch, summed over the kernels)" if traffic else None,
        "stale_counters": {"traffic": bool(tstale), "mfma_busy": bool(sstale),
                           "note": "true = the committed counter file was taken on other kernel sources than this tree's (or records none): not reported"},
        "basis": "ALGORITHMIC FLOPs, SURVEY 8(d): backward = 5 products = 5 U, U = 2 B H n^2 d; recomputed products are not credited; "
                 "the timed kernels also produce the c_qkv bias gradient (column sums of dq / dk / dv from their row stores, ~6 us of avg_ms, "
                 "instead of a separate 46 us pass over dqkv)",
        "algorithmic_flops_per_launch": 5 * U, "avg_ms": bwd_ms, "launches": len(events[bwd_tags[0]]),
        "executed_flops_per_launch": sum(executed[k] for k in bwd_tags),
        "achieved_on_executed_flops": tf(sum(executed[k] for k in bwd_tags), bwd_ms),
        "forward": {"kernel": knames["fwd"], "achieved": fwd_tf, "frac": fwd_tf / PEAK_BF16_TFLOPS, "algorithmic_flops_per_launch": 2 * U,
                    "avg_ms": kern_ms["fwd"]},
        "forward_plus_backward": {"achieved": all_tf, "frac": all_tf / PEAK_BF16_TFLOPS, "algorithmic_flops": 7 * U, "avg_ms": bwd_ms + kern_ms["fwd"]},
        "per_kernel_ms": kern_ms,
        "per_kernel_tflops_executed": {k: tf(executed[k], kern_ms[k]) for k in kern_ms},
        "mfma_busy": None if sstale else ({k: sjson[knames[k]] for k in kern_ms if knames[k] in sjson} or None),
        "mfma_busy_source": (f"profiles/{sname} (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES etc., rocprofv3 --pmc on the same kernels; taken on "
                             f"another box than this run: a property of the kernels, not of this run's clock)") if sjson and not sstale else None,
    }

    # the HBM-bound kernels of the step (residual stream fp32, activations bf16; T tokens x W / 4W columns), priced at their
    # algorithmic bytes: every operand read once, every result written once
    T, Wd = per * n, CFG["width"]
    ew_bytes = {"add_ln_fwd": T * Wd * (4 + 2 + 4 + 2), "ln_bwd": T * Wd * (2 + 4 + 4 + 4 + 2),
                "gelu_fwd": T * 4 * Wd * (2 + 2), "gelu_bwd": T * 4 * Wd * (2 + 2 + 2)}
    ew_ms = {k: (sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in ew_events.items() if v}
    hbm = None
    if ew_ms:
        dom = max(ew_ms, key=lambda k: ew_ms[k] * len(ew_events[k]))
        gbs = {k: ew_bytes[k] / (ew_ms[k] * 1e-3) / 1e9 for k in ew_ms}
        hname = {"add_ln_fwd": "add_ln_fwd_kernel", "ln_bwd": "ln_bwd_kernel", "gelu_fwd": "gelu_fwd_kernel", "gelu_bwd": "colsum_kernel<true>"}[dom]
        hfile, hjson, hstale = newest("r6_elementwise_hbm_traffic_pmc.json", "r5_elementwise_hbm_traffic_pmc.json", "r2_elementwise_hbm_traffic_pmc.json", sources=("elementwise.hip", "common.h"))
        htraffic = hjson.get(hname, {}).get("hbm_bytes") if per == 64 and not hstale else None   # PMC-measured bytes per launch at exactly this shape
        hbm = {"kernel": hname,
               "bound": "hbm", "achieved": gbs[dom], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs[dom] / PEAK_HBM_GBS,
               "traffic": htraffic, "traffic_source": f"profiles/{hfile}" if htraffic else None, "stale_counters": bool(hstale),
               "algorithmic_bytes_per_launch": ew_bytes[dom], "avg_ms": ew_ms[dom], "launches": len(ew_events[dom]),
               "all_elementwise_kernels_ms": ew_ms, "all_elementwise_kernels_gbs": gbs}

    result = {
        "metric": "denoiser train steps/sec (+ rendered rays/sec under 'render'), SRN-Cars 512pt x 128d",
        "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: denoiser fwd/bwd + AdamW + EMA, 512 points x 128-d latents, "
                               "width 1024 / 24 layers / 16 heads (seq 513), bf16 autocast",
                   "global_batch": CFG["global_batch"], "per_gpu_batch": per, "parallelism": f"dp{world}",
                   "rng": "weights seed 1234 on every rank; timestep / noise draws seed 42 + rank"},
        "native_path": {"fused_backbone_engine": True, "fused_adamw_ema": True, "sharded_optimizer": bool(trainer.reducer.shard)},
        "step_tflops": denoiser_flops_per_sample() * CFG["global_batch"] / (elapsed / args.steps) / 1e12,
        "step_frac_of_bf16_mfma_peak": denoiser_flops_per_sample() * CFG["global_batch"] / (elapsed / args.steps) / 1e12
                                       / (PEAK_BF16_TFLOPS * world),
        "roofline": roofline,
        "roofline_hbm": hbm,
        "loss": float(loss),
        "tuned_gemm_file": (os.path.relpath(tuned, ROOT) if os.path.abspath(tuned).startswith(ROOT + os.sep) else tuned) if use_tuned else None,
        "clock": clocks.summary(),
    }
    if world > 1:
        # what went over the wire in the last step, and proof that the ranks still hold the same model
        trainer.wait_params()
        # elementwise: rank 0's flat parameter buffer is broadcast and every rank reports max |p - p_rank0| (0.0 = bit-identical
        # up to the sign of zero); the maximum over ranks goes into the line
        p0 = trainer.flat.flat.detach().clone()
        dist.broadcast(p0, src=0)
        dev_max = (trainer.flat.flat.detach() - p0).abs().max().reshape(1).double()
        dev_max = torch.nan_to_num(dev_max, nan=float("inf"))
        dist.all_reduce(dev_max, op=dist.ReduceOp.MAX)
        del p0
        comm = trainer.comm_stats()
        comm["max_abs_parameter_difference_to_rank0"] = float(dev_max)
        comm["parameters_identical_across_ranks"] = bool(float(dev_max) == 0.0)
        comm["backend"] = dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else " (one-GPU dry run)")
        comm["world_size_seen_by_backend"] = dist.get_world_size()
        seen = torch.ones(1, device=device)
        dist.all_reduce(seen)                       # every rank of the communicator adds 1: counts the ranks the collective reached
        comm["ranks_counted_by_all_reduce"] = int(seen.item())
        comm["launched_by"] = os.environ.get("TORCHELASTIC_RUN_ID") and "torch.distributed.run" or "external launcher"
        comm["rccl_env"] = {k: os.environ[k] for k in ("NCCL_ALGO", "NCCL_PROTO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS") if k in os.environ}
        result["comm"] = comm
    if world == 1 and not args.no_proxy:
        try:
            result["strong_scaling_proxy"] = strong_scaling_proxy(trainer, coords, feats)
        except Exception as e:                      # noqa: BLE001
            result["strong_scaling_proxy"] = {"error": f"{type(e).__name__}: {e}"}
    # the secondary measurements must never take the headline line down with them
    if not args.no_render:
        try:
            r = bench_render(device)
        except Exception as e:                      # noqa: BLE001
            r = {"error": f"{type(e).__name__}: {e}", "rays_per_s": 0.0}
        if world > 1:       # every rank renders its own views: aggregate rays/s (replicas, no collective on the path)
            tt = torch.tensor([r["rays_per_s"]], device=device, dtype=torch.float64)
            dist.all_reduce(tt)
            r["rays_per_s_all_gpus"] = float(tt)
        result["render"] = r
        try:
            # lead figure: the reference's numerics -- stage 1 trains in TRUE fp32 (train_pointnerf.py has no autocast), which is what
            # PointNeRFTrainer(mlp_dtype=None) runs: fp32 operands and accumulation on library GEMMs for every Linear layer (round 6, ADVICE
            # r5: the split-operand mode had been the default for one round).  Beside it the two explicit opt-ins: "fp32_class" (per-pair
            # layers + the point-level forward on the matrix cores with every operand as two bf16 halves: ~1e-5 relative per product,
            # tests/test_gpu_train_render.py::test_fused_pair_mlp_fp32_class_mode) and bf16 operands (narrower still).
            keys = ("steps_per_s", "ms_per_step", "ms_per_step_min_median_max", "loss", "differentiable_part")
            s1 = bench_stage1(device)
            s1["numerics"] = ("fp32 operands and accumulation on library GEMMs for every Linear layer, losses and Adam in fp32 "
                              "(PointNeRFTrainer(mlp_dtype=None)): the reference's arithmetic")
            x2 = bench_stage1(device, mlp_dtype="fp32_class")
            s1["fp32_class_opt_in"] = {k: x2[k] for k in keys}
            s1["fp32_class_opt_in"]["numerics"] = ("EMULATED fp32 (opt-in, PointNeRFTrainer(mlp_dtype='fp32_class')): every operand as two bf16 halves (hi + lo), "
                                                   "three matrix instructions per product, fp32 accumulation and weight gradients -- 16 mantissa bits, ~1e-5 relative per "
                                                   "product where fp32 has 6e-8; per-pair layers forward + backward (csrc/pairs_mlp.hip precision 1) and the point-level "
                                                   "layers' forward (csrc/points_x2.hip); the text of `differentiable_part` is derived from the predicates the forward uses")
            opt = bench_stage1(device, mlp_dtype=torch.bfloat16)
            s1["bf16_operands_opt_in"] = {k: opt[k] for k in keys}
            s1["bf16_operands_opt_in"]["numerics"] = ("bf16 operands, fp32 accumulation, fp32 weight gradients and optimizer: NARROWER than the "
                                                      "reference's fp32 -- not the creditable figure")
            result["stage1_pointnerf_training"] = s1
        except Exception as e:                      # noqa: BLE001
            result["stage1_pointnerf_training"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_cfg5:
        try:
            result["cfg5_stress_step"] = bench_cfg5(device)
        except Exception as e:                      # noqa: BLE001
            result["cfg5_stress_step"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_sampler:
        try:
            result["sampler_reverse_step"] = bench_sampler(device)
        except Exception as e:                      # noqa: BLE001
            result["sampler_reverse_step"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_sampler and not args.no_render:
        try:
            result["sample_and_render"] = bench_sample_and_render(device)
        except Exception as e:                      # noqa: BLE001
            result["sample_and_render"] = {"error": f"{type(e).__name__}: {e}"}
        if args.eval_loop_fp32_class:
            try:
                result["sample_and_render_fp32_class"] = bench_sample_and_render(device, fp32_class=True)
            except Exception as e:                  # noqa: BLE001
                result["sample_and_render_fp32_class"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"], ref = cpu_baseline()
                result["gpu_over_cpu"] = {"denoiser_steps": result["value"] / result["cpu_baseline"]["value"]}
                if result.get("render", {}).get("rays_per_s"):
                    result["gpu_over_cpu"]["render_rays"] = result["render"]["rays_per_s"] / result["cpu_baseline"]["render_rays_per_s"]
            except Exception as e:                  # noqa: BLE001
                result["cpu_baseline"], ref = {"error": f"{type(e).__name__}: {e}"}, None
            if ref is not None:
                try:
                    del trainer
                    torch.cuda.empty_cache()
                    result["parity_full_width"] = parity_full_width(device, ref)
                except Exception as e:              # noqa: BLE001
                    result["parity_full_width"] = {"error": f"{type(e).__name__}: {e}", "ok": False}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
