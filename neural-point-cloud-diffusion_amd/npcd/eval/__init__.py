"""Evaluation protocol of the renderer (reference npcd/eval/pointnerf_evaluation.py): per view, `--eval_batch_size` views at a
time, synchronise - time - render - synchronise (:215-233; the first `burn_in_samples` objects give no timing, :224), PSNR of every
rendered view against its ground-truth image with data range 1 (skimage.metrics.peak_signal_noise_ratio, :242-257), mean over
all views of all objects (:288).  Only the measurement loop is here: datasets, writers, qualitative dumps are out of scope
(SURVEY section 8)."""
import math
import time
from typing import Dict, Iterable, List

import torch


def psnr(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> float:
    """skimage.metrics.peak_signal_noise_ratio in float64 (:254): 10 log10(data_range^2 / mse)."""
    mse = float(((pred.double() - target.double()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * math.log10(data_range ** 2 / mse)


def unflatten_pred(channels: torch.Tensor) -> torch.Tensor:
    """[..., R, C] -> [..., C, res, res] (utils/util.py:199-203; ray r = i * res + j is pixel (i, j))."""
    x = channels.transpose(-1, -2)
    side = round(x.shape[-1] ** 0.5)
    return x.reshape(*x.shape[:-1], side, side)


@torch.no_grad()
def evaluate_pointnerf(pointnerf, samples: Iterable[Dict[str, torch.Tensor]], eval_batch_size: int = 1, burn_in_samples: int = 3) -> Dict:
    """samples: dicts with obj_idx [1], intrinsics [1,V,3,3], extrinsics [1,V,4,4], images [1,V,3,H,W] (one object each, like the
    reference's dataloader).  Returns {'psnr': mean over all views, 'runtime_model_in_msec': mean over the timed calls (NaN-free),
    'views': per-view records}."""
    records: List[Dict] = []
    levels = set()
    dev = next(pointnerf.parameters()).device
    for num, sample in enumerate(samples):
        sample = {k: v.to(dev) for k, v in sample.items()}
        V = sample["extrinsics"].shape[1]
        for v0 in range(0, V, eval_batch_size):
            sl = slice(v0, min(V, v0 + eval_batch_size))
            torch.cuda.synchronize()
            t0 = time.time()
            pred, _ = pointnerf(sample["obj_idx"], sample["intrinsics"][:, sl], sample["extrinsics"][:, sl], sample_rays=False)
            torch.cuda.synchronize()
            dt = time.time() - t0
            timed = eval_batch_size == 1 and num >= burn_in_samples
            levels.add(pred.get("grid_level"))
            imgs = unflatten_pred(pred["channels"].contiguous()[0])
            for j, img in enumerate(imgs):
                records.append({"sample": num, "view": v0 + j, "psnr": psnr(img, sample["images"][0, v0 + j]),
                                "runtime_model_in_msec": 1000 * dt if timed else float("nan")})
    times = [r["runtime_model_in_msec"] for r in records if not math.isnan(r["runtime_model_in_msec"])]
    return {"psnr": sum(r["psnr"] for r in records) / max(1, len(records)),
            "runtime_model_in_msec": sum(times) / len(times) if times else float("nan"), "views": records,
            "grid_level": sorted(str(l) for l in levels)}


# ---- generate -> render: the loop of the reference's DiffusionEvaluation (npcd/eval/diffusion_evaluation.py:146-183) ---------------
def load_test_poses(name: str = "srncars"):
    """The 251 test poses / intrinsics the reference evaluates every generated cloud on (its data/<name>_test_poses.npy,
    data/<name>_test_intrinsics.npy, diffusion_evaluation.py:41-43): bundled under npcd/data/ (data, not code)."""
    import os
    import numpy as np
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
    poses = torch.from_numpy(np.load(os.path.join(d, f"{name}_test_poses.npy"))).float()
    intr = torch.from_numpy(np.load(os.path.join(d, f"{name}_test_intrinsics.npy"))).float()
    return poses, intr


@torch.no_grad()
def sample_and_render(model, poses: torch.Tensor, intrinsics: torch.Tensor, num_samples: int = 4, generate_batch_size: int = 4,
                      render_batch_size: int = 8, resolution: int = 128, dtype=None, use_graph: bool = False, feed=None,
                      max_shading_points=None, render_mlp_dtype=None) -> Dict:
    """The measurement loop of the reference's DiffusionEvaluation._evaluate (:146-183): `generate_batch_size` clouds at a time from
    model.diffusion.generate (:152), every cloud rendered from all poses, `render_batch_size` poses per PointNeRF.render call
    (:163-169), images clipped to [0, 1] and rounded to 8 bits (:172-173).  `feed(images [n_poses, 3, H, W] in [0, 1])`, when given,
    receives every cloud's images (the reference hands them to FID/KID as images * 2 - 1, :179; the Inception network is outside this
    package).  `dtype`: run the sampler's denoiser under autocast, or "fp32_class" (DiffusionModel.generate);
    `render_mlp_dtype=torch.float32`: shade in the reference's numerics class (PointNeRF.render).  Returns timings:
    seconds in generate / render, clouds per second, rendered views and rays per second (device-synchronised walls)."""
    model.eval()
    dev = next(model.pointnerf.parameters()).device
    poses, intrinsics = poses.to(dev).float(), intrinsics.to(dev).float()
    n_poses = poses.shape[0]
    t_gen = t_ren = 0.0
    n_views = 0
    shapes = None
    for s0 in range(0, num_samples, generate_batch_size):
        n = min(generate_batch_size, num_samples - s0)
        torch.cuda.synchronize()
        t0 = time.time()
        coords_b, feats_b = model.diffusion.generate(num=n, batch_size=n, progress=False, dtype=dtype, use_graph=use_graph)
        torch.cuda.synchronize()
        t_gen += time.time() - t0
        for coords, feats in zip(coords_b, feats_b):
            coords = coords.permute(1, 0)[None].contiguous()               # the format PointNeRF takes (:156-157)
            feats = feats.permute(1, 0)[None].contiguous()
            torch.cuda.synchronize()
            t0 = time.time()
            imgs = []
            for p0 in range(0, n_poses, render_batch_size):
                out = model.pointnerf.render(coords, feats, poses[None, p0:p0 + render_batch_size], intrinsics[None, p0:p0 + render_batch_size],
                                             resolution=resolution, max_shading_points=max_shading_points, mlp_dtype=render_mlp_dtype)
                im = unflatten_pred(out["channels"])[0].clamp(0.0, 1.0)
                imgs.append(torch.round(im * 255.0) / 255.0)
            images = torch.cat(imgs)
            torch.cuda.synchronize()
            t_ren += time.time() - t0
            n_views += n_poses
            shapes = tuple(images.shape)
            if feed is not None:
                feed(images)
    return {"clouds": num_samples, "poses_per_cloud": n_poses, "resolution": resolution, "image_batch_shape": shapes,
            "generate_seconds": t_gen, "render_seconds": t_ren, "clouds_per_s_generate": num_samples / t_gen if t_gen else None,
            "views_per_s": n_views / t_ren if t_ren else None, "rays_per_s": n_views * resolution * resolution / t_ren if t_ren else None,
            "clouds_per_s_end_to_end": num_samples / (t_gen + t_ren)}
