#!/usr/bin/env python3
"""Benchmark of the NPCD hot path on MI355X (contract: see the repo-level task description).

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

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
PEAK_HBM_GBS = 8000.0


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
    return DiffusionTrainer(model, lr=7e-5, weight_decay=0.01, ema_decay=0.9999, dtype=torch.bfloat16)


def synthetic_batch(global_batch, rank, world, device):
    g = torch.Generator().manual_seed(42)
    coords = torch.randn(global_batch, CFG["coords_dim"], CFG["num_points"], generator=g)
    feats = torch.rand(global_batch, CFG["feats_dim"], CFG["num_points"], generator=g) * 2 - 1
    per = global_batch // world
    sl = slice(rank * per, (rank + 1) * per)
    return coords[sl].to(device), feats[sl].to(device)


def bench_render(device, n_iters=100, burn_in=5):
    """pointnerf_evaluation.py:217-224 protocol: sync, t0, render, sync, t1; burn-in renders discarded (the reference drops 3
    and then renders 251 views per object: the timed region here is 100 back-to-back renders, i.e. the sustained rate)."""
    from npcd.models.pointnerf import PointNeRF
    from npcd.utils import synthetic as orr
    coords, feats = orr.ellipsoid_cloud(512, 32, 1, seed=0)
    torch.manual_seed(0)
    net = PointNeRF(1, 32, 512, False)            # field MLPs: PyTorch default init (random weights)
    net = net.to(device).eval()
    extr = orr.look_at_pose(30, 20)[None, None].to(device)
    intr = orr.srn_intrinsics()[None, None].to(device)
    c, f = coords.to(device), feats.to(device)
    with torch.no_grad():
        for _ in range(burn_in):
            out = net.render(c, f, extr, intr, 128)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_iters):
            out = net.render(c, f, extr, intr, 128)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_iters
        # throughput form: 8 views of the same object in one call (launch / sync overhead amortised)
        poses = torch.stack([orr.look_at_pose(30 + 45 * i, 20 - 5 * i) for i in range(8)])[None].to(device)
        intr8 = orr.srn_intrinsics()[None, None].expand(1, 8, 3, 3).contiguous().to(device)
        for _ in range(2):
            net.render(c, f, poses, intr8, 128)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            net.render(c, f, poses, intr8, 128)
        torch.cuda.synchronize()
        dt8 = (time.perf_counter() - t0) / 20
        net.renderer.count_pairs = True            # workload statistics (FLOP accounting), outside the timed region
        out = net.render(c, f, extr, intr, 128)
    P, Q = int(out["num_shading_points"]), int(out["num_pairs"])
    flops = Q * 2 * (95 * 256 + 3 * 256 * 256) + P * 2 * (256 * 256 * 6 + 256 + 3 * 256)   # as executed (last agg layer on points)
    return {"rays_per_s": 128 * 128 / dt, "rays_per_s_8_views_per_call": 8 * 128 * 128 / dt8, "ms_per_view": dt * 1e3, "timed_renders": n_iters, "resolution": 128, "depth_samples": 128, "k": 8,
            "shading_points": P, "pairs": Q, "mlp_tflops": flops / dt / 1e12,
            "mlp_frac_of_f16_mfma_peak": flops / dt / 1e12 / PEAK_BF16_TFLOPS}


def bench_stage1(device, n_iters=5, burn_in=2, mlp_dtype=None):
    """Stage-1 (PointNeRF autodecoder) training step at the reference's configuration (configs/npcd_srncars.yaml:13-16,
    data/srn.py:45: 8 objects x 50 views per step, 112 random rays per view, 128 depth samples, Adam lr 1e-3): secondary
    figure for SURVEY §8(f) rank 2.  Synthetic clouds / poses / target images."""
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
    for _ in range(burn_in):
        tr.step(sample)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_iters):
        loss, _ = tr.step(sample)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n_iters
    return {"steps_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "objects": B, "views_per_object": T, "rays_per_view": net.pointnerf.renderer.ray_subsamples,
            "loss": float(loss), "differentiable_part": "HIP: ray generation, both neighbour queries, pair inputs, aggregation and ray march (forward + backward); library GEMMs (row-split weight gradients) and torch autograd for the MLP layers"}


def cpu_baseline():
    """The CPU oracle (a restatement of the reference, `kind: port`) timed on this box's host cores:
    one fp32 denoiser train step (fwd + bwd + AdamW) at B=2 of the same architecture, scaled to the
    B=64 step; plus one 64x64 render (grid semantics) for the rays/s half."""
    from oracle import denoiser as od, diffusion as odf, renderer as orr
    cores = min(os.cpu_count() or 1, 32)          # more threads only add contention for these op sizes
    torch.set_num_threads(cores)
    B, L_sample = 8, CFG["layers"]                # bounded sample: 8 of the 64 samples of the step, all 24 blocks
    params = od.init_params(CFG["coords_dim"], CFG["feats_dim"], CFG["width"], L_sample, CFG["heads"], seed=0)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.AdamW(list(leaves.values()), lr=7e-5, weight_decay=0.01)
    g = torch.Generator().manual_seed(42)
    c0 = torch.randn(B, 3, CFG["num_points"], generator=g)
    f0 = torch.rand(B, CFG["feats_dim"], CFG["num_points"], generator=g) * 2 - 1
    tab = odf.schedule_tables()
    t = torch.randint(0, 1000, (B,), generator=g)
    cn, fn = torch.randn_like(c0), torch.randn_like(f0)

    def step():
        opt.zero_grad()
        loss, _, _ = odf.p_losses(tab, lambda c, f, tt: od.denoiser_forward(leaves, c, f, tt, CFG["heads"]), c0, f0, t, cn, fn)
        loss.backward()
        opt.step()

    step()                                        # untimed: thread-pool / allocator warm-up
    t0 = time.perf_counter(); step(); dt_sample = time.perf_counter() - t0
    dt = dt_sample * CFG["layers"] / L_sample     # transformer blocks are > 99 % of the step
    steps_per_s = (B / dt) / CFG["global_batch"]
    fp = orr.init_field_params(32, seed=0)
    coords, feats = orr.synthetic_cloud(512, 32, 1, seed=0)
    res = 128
    K = orr.srn_intrinsics().clone(); K[0, 0] = K[1, 1] = 131.25 * res / 128; K[0, 2] = K[1, 2] = res / 2
    t0 = time.perf_counter()
    with torch.no_grad():
        orr.render(fp, coords, feats, orr.look_at_pose(30, 20)[None, None], K[None, None], res=res)
    dtr = time.perf_counter() - t0
    return {"value": steps_per_s, "unit": "steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle (fp32 PyTorch CPU restatement of the reference), {cores} threads: 1 denoiser train step (fwd+bwd+AdamW) "
                      f"at B={B} of {CFG['global_batch']} with {L_sample} of {CFG['layers']} blocks = {dt_sample:.1f} s, scaled x{CFG['layers'] // L_sample} "
                      f"(blocks) and x{CFG['global_batch'] // B} (batch) ; render: one {res}x{res} view = {dtr:.1f} s",
            "render_rays_per_s": res * res / dtr}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-render", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    # NPCD_BENCH_DRYRUN_ONE_GPU=1 (tests only): all ranks share cuda:0 and talk through gloo, so that the multi-rank code path
    # of this script can be exercised on a one-GPU box (RCCL refuses two ranks on one device)
    dryrun = bool(os.environ.get("NPCD_BENCH_DRYRUN_ONE_GPU"))
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
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
    tuned = os.path.join(ROOT, "profiles", "tunableop_gfx950.csv")
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

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(coords, feats)
    barrier()
    hattn.KERNEL_EVENTS = {"fwd": [], "dq": [], "dkdv": []}
    hew.KERNEL_EVENTS = {"add_ln_fwd": [], "ln_bwd": [], "gelu_fwd": [], "gelu_bwd": []}
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

    kern_ms = {k: (sum(a.elapsed_time(b) for a, b in v) / max(1, len(v))) for k, v in events.items()}
    n = CFG["num_points"] + 1
    unit_flops = 2 * per * CFG["heads"] * n * n * 64          # one B x H x n x n x d product
    alg = {"fwd": 2 * unit_flops, "dq": 3 * unit_flops, "dkdv": 4 * unit_flops}
    dominant = max(kern_ms, key=lambda k: kern_ms[k])
    achieved = alg[dominant] / (kern_ms[dominant] * 1e-3) / 1e12

    # HBM traffic of the dominant kernel: measured with rocprofv3 PMC counters (FETCH_SIZE/WRITE_SIZE, separate passes,
    # gfx950 x2 read correction) on this exact kernel and shape -- bench.py itself cannot collect PMC counters
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "r1_attention_hbm_traffic_pmc.json")
    kname = {"fwd": "attn_fwd_kernel", "dq": "attn_bwd_dq_kernel", "dkdv": "attn_bwd_dkdv_kernel"}[dominant]
    if per == 64 and os.path.exists(tfile):
        traffic = json.load(open(tfile)).get(kname, {}).get("hbm_bytes")

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
        htraffic, hfile = None, os.path.join(ROOT, "profiles", "r1_elementwise_hbm_traffic_pmc.json")
        if per == 64 and os.path.exists(hfile):         # PMC-measured bytes per launch at exactly this shape (rocprofv3 --pmc passes)
            htraffic = json.load(open(hfile)).get(hname, {}).get("hbm_bytes")
        hbm = {"kernel": hname,
               "bound": "hbm", "achieved": gbs[dom], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs[dom] / PEAK_HBM_GBS,
               "traffic": htraffic, "traffic_source": "profiles/r1_elementwise_hbm_traffic_pmc.json" if htraffic else None,
               "algorithmic_bytes_per_launch": ew_bytes[dom], "avg_ms": ew_ms[dom], "launches": len(ew_events[dom]),
               "all_elementwise_kernels_ms": ew_ms, "all_elementwise_kernels_gbs": gbs}

    result = {
        "metric": "denoiser train steps/sec (+ rendered rays/sec under 'render'), SRN-Cars 512pt x 128d",
        "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: denoiser fwd/bwd + AdamW + EMA, 512 points x 128-d latents, "
                               "width 1024 / 24 layers / 16 heads (seq 513), bf16 autocast",
                   "global_batch": CFG["global_batch"], "per_gpu_batch": per, "parallelism": f"dp{world}"},
        "step_tflops": denoiser_flops_per_sample() * CFG["global_batch"] / (elapsed / args.steps) / 1e12,
        "step_frac_of_bf16_mfma_peak": denoiser_flops_per_sample() * CFG["global_batch"] / (elapsed / args.steps) / 1e12
                                       / (PEAK_BF16_TFLOPS * world),
        "roofline": {"kernel": kname,
                     "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic,
                     "traffic_source": "profiles/r1_attention_hbm_traffic_pmc.json (rocprofv3 --pmc, bytes per launch)" if traffic else None,
                     "algorithmic_flops_per_launch": alg[dominant],
                     "avg_ms": kern_ms[dominant], "launches": len(events[dominant]),
                     "all_attention_kernels_ms": kern_ms,
                     "all_attention_kernels_tflops": {k: alg[k] / (kern_ms[k] * 1e-3) / 1e12 for k in kern_ms}},
        "roofline_hbm": hbm,
        "loss": float(loss),
        "tuned_gemm_file": "profiles/tunableop_gfx950.csv" if use_tuned else None,
    }
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
            result["stage1_pointnerf_training"] = bench_stage1(device)
            opt_in = bench_stage1(device, mlp_dtype=torch.bfloat16)
            result["stage1_pointnerf_training"]["opt_in_bf16_mlp"] = {k: opt_in[k] for k in ("steps_per_s", "ms_per_step", "loss")}
        except Exception as e:                      # noqa: BLE001
            result["stage1_pointnerf_training"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline()
                result["gpu_over_cpu"] = {"denoiser_steps": result["value"] / result["cpu_baseline"]["value"]}
                if result.get("render", {}).get("rays_per_s"):
                    result["gpu_over_cpu"]["render_rays"] = result["render"]["rays_per_s"] / result["cpu_baseline"]["render_rays_per_s"]
            except Exception as e:                  # noqa: BLE001
                result["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
