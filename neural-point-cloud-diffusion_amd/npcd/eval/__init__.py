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
