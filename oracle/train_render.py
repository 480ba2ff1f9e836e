"""Oracle (CPU, fp32 torch + numpy) restatement of the TRAINING-mode renderer path and the stage-1 losses
(SURVEY.md §8(f) rank 2).  TEST INFRASTRUCTURE.  Pinned by tests/golden/{train_render,losses,variational_embedding}.npz
(tests/golden/make_golden_train.py imports the reference and records its random draws).

The reference draws random numbers inline; here every draw is an explicit argument:
  ray_perm    torch.randperm(num_rays)                    renderer.py:233
  jitter      torch.rand_like(depths)  [B*T, R', S, 1]    renderer.py:74-76
  valid_perm  torch.randperm(total_valid_rays)            aggregator.py:98
  eps         torch.randn_like(std)                       variational_embedding.py:53
"""
import numpy as np
import torch

from .renderer import (DEFAULT_GRID, Params, camera_rays, depths_from_points, ray_box_limits, ray_march, shade_points)
from .voxel_grid import VoxelGridOracle, brute_force_query


def subsample_rays(o, d, ray_perm, n_sub):
    """renderer.py:232-238: the SAME n_sub rays (first n_sub of one permutation) for every (object, view) instance.
    o, d [B,T,R,3] -> o, d [B,T,n_sub,3], ray_idx [n_sub]."""
    idx = ray_perm[:n_sub].long()
    return o[:, :, idx], d[:, :, idx], idx


def jittered_depths(start, end, S, jitter):
    """renderer.py:49-77 with randomize_depth_samples: linspace + U(0,1) * (end-start)/(S-1).  start/end [...,1],
    jitter [...,S] -> [...,S]."""
    steps = torch.arange(S, dtype=torch.float32) / (S - 1)
    dep = start + steps * (end - start)
    return dep + jitter * ((end - start) / (S - 1))


def subsample_valid_rays(slot_mask, valid_perm, ray_subsamples, stable=True):
    """aggregator.py:78-119.  slot_mask [I, R', M] bool (I = B*T instances) -> ray_sample_mask [I, R'] bool with the same
    number of rays selected per instance: min(min_i #valid_i, ray_subsamples), taken from a random shuffle of each
    instance's valid rays.  The reference regroups the shuffled list by instance with torch.argsort, whose order among
    equal keys is unspecified: `stable=False` calls the same primitive (reproduces the fixture bit for bit on CPU torch),
    `stable=True` is the specification the HIP build follows (shuffled order kept inside every instance) -- both are
    uniformly random subsets of the valid rays."""
    valid = slot_mask.any(dim=-1)                                   # [I, R']
    pairs = torch.nonzero(valid)                                    # [total, 2] row-major
    pairs = pairs[valid_perm.long()]
    order = torch.sort(pairs[:, 0], stable=True).indices if stable else torch.argsort(pairs[:, 0])
    rays = pairs[order, 1]
    num_valid = valid.sum(dim=-1)
    n = int(min(int(num_valid.min()), ray_subsamples))
    start = torch.cumsum(num_valid, 0) - num_valid
    sel = rays[(start[:, None] + torch.arange(n)[None, :]).reshape(-1)].reshape(valid.shape[0], n)
    out = torch.zeros_like(valid)
    out.scatter_(1, sel, True)
    return out, n


def render_train(p: Params, coords, feats, extr, intr, res, S, M, k, r, mode, renderer_ray_subsamples,
                 aggregator_ray_subsamples, ray_perm, jitter, valid_perm, grid_cfg=None, cube_scale=1.0, white_back=True,
                 stable_regroup=True):
    """Renderer.forward(sample=True) in train mode (renderer.py:202-268, volume_renderer.py:41-92, field.py:77-152,
    aggregators/mlp.py:36-100).  Differentiable w.r.t. feats and the field weights p (torch autograd on CPU).
    -> dict(mask [B,T,n,1], depth [B,T,n,1], channels [B,T,n,3], ray_idx [B,T,n,1])."""
    B, T = extr.shape[:2]
    o, d = camera_rays(extr.flatten(0, 1).float(), intr.flatten(0, 1).float(), res)
    R = o.shape[1]
    o, d = o.reshape(B, T, R, 3), d.reshape(B, T, R, 3)
    if renderer_ray_subsamples:
        o, d, ray_ids = subsample_rays(o, d, ray_perm, renderer_ray_subsamples)
    else:
        ray_ids = torch.arange(R)
    Rs = o.shape[2]
    start, end = ray_box_limits(o.reshape(B, T * Rs, 3), d.reshape(B, T * Rs, 3), cube_scale)   # global min/max over ALL instances
    start, end = start.reshape(B, T, Rs, 1), end.reshape(B, T, Rs, 1)
    dep = jittered_depths(start, end, S, jitter.reshape(B, T, Rs, S))
    x = o[..., None, :] + dep[..., None] * d[..., None, :]                                      # [B,T,Rs,S,3]
    xn = x.detach().reshape(B, T * Rs, S, 3).numpy()
    cn = coords.detach().numpy()
    if mode == "grid":
        grid = VoxelGridOracle(**(grid_cfg or DEFAULT_GRID))
        grid.set_pointset(cn, np.full((B,), coords.shape[1], dtype=np.int32))
        idx, loc, _, _ = grid.query_dense(xn, k, r, M)
    else:
        idx, loc, _ = brute_force_query(xn, cn, k, r, M)
    idx_t = torch.from_numpy(idx.astype(np.int64)).reshape(B * T, Rs, M, k)
    loc_t = torch.from_numpy(loc).reshape(B * T, Rs, M, 3)
    slot_valid = (idx_t >= 0).any(dim=-1)                                                       # [I,Rs,M]
    ray_sel, n = subsample_valid_rays(slot_valid, valid_perm, aggregator_ray_subsamples, stable_regroup)      # [I,Rs]
    idx_s = idx_t[ray_sel].reshape(B * T * n, M, k)                                             # ascending ray order
    loc_s = loc_t[ray_sel].reshape(B * T * n, M, 3)
    valid_s = slot_valid[ray_sel].reshape(B * T * n, M)
    nb, pts = idx_s[valid_s], loc_s[valid_s]
    sigma_c, rgb_c, _ = shade_points(p, nb, pts, coords, feats)
    rows = torch.nonzero(valid_s)
    sigma = torch.zeros(B * T * n, M).index_put((rows[:, 0], rows[:, 1]), sigma_c[:, 0])
    rgb = torch.zeros(B * T * n, M, 3).index_put((rows[:, 0], rows[:, 1]), rgb_c)
    dense_pts = torch.zeros(B * T * n, M, 3).index_put((rows[:, 0], rows[:, 1]), pts)
    o_s = o.reshape(B * T, Rs, 3)[ray_sel]
    d_s = d.reshape(B * T, Rs, 3)[ray_sel]
    end_s = end.reshape(B * T, Rs, 1)[ray_sel]
    depths = depths_from_points(dense_pts, valid_s, o_s, d_s, end_s)
    total, cdepth, chan = ray_march(sigma, depths, rgb, valid_s, white_back)
    ray_idx = ray_ids[None, :].expand(B * T, Rs)[ray_sel].reshape(B, T, n, 1)
    return {"mask": total.reshape(B, T, n, 1), "depth": cdepth.reshape(B, T, n, 1), "channels": chan.reshape(B, T, n, 3),
            "ray_idx": ray_idx, "num_rays": n, "num_shading_points": int(nb.shape[0]), "num_pairs": int((nb >= 0).sum())}


# ---------------------------------------------------------------------------------------------- embeddings / losses
def variational_embedding(table, idx, n_kp, out_dim, eps=None):
    """variational_embedding.py:36-58: rows [n_obj, n_kp*2*out_dim] -> mean (+ exp(log_var/2) * eps in train mode)."""
    emb = table[idx].reshape(-1, n_kp, 2 * out_dim)
    mean, log_var = emb[..., :out_dim], emb[..., out_dim:]
    return mean if eps is None else mean + torch.exp(0.5 * log_var) * eps


def kl_loss(feats_mean, feats_log_var, weight=1.0):
    """neural_point_cloud_kl_loss.py:29-44 -> (scalar, pointwise [B,N])."""
    kld = -0.5 * torch.sum(1 + feats_log_var - feats_mean.pow(2) - feats_log_var.exp(), dim=-1) * weight
    return kld.mean(), kld


def subsample_gt(gt_map, ray_idx):
    """utils/util.py:188-196: images [B,T,C,H,W] -> [B,T,R,C] (gathered at ray_idx [B,T,n,1] if given)."""
    samples = gt_map.flatten(-2, -1).transpose(-1, -2)
    if ray_idx is not None:
        samples = samples.gather(dim=-2, index=ray_idx.expand(*ray_idx.shape[:-1], samples.shape[-1]))
    return samples


def image_loss(images, pred_channels, ray_idx=None, weight=1.0):
    """image_reconstruction_loss.py:28-40."""
    return ((pred_channels - subsample_gt(images, ray_idx)) ** 2).mean() * weight


def tv_loss(coords, feats, k, r, weight=1.0, mode="brute", grid_cfg=None):
    """neural_point_cloud_tv_loss.py:29-83: inverse-distance weighted L1 feature variation over each point's k nearest
    neighbours inside the radius (the point itself is dropped when it has another neighbour) -> (scalar, pointwise [B,N])."""
    B, N = coords.shape[:2]
    x = coords.detach().reshape(B, N, 1, 3).numpy()
    if mode == "grid":
        grid = VoxelGridOracle(**(grid_cfg or DEFAULT_GRID))
        grid.set_pointset(coords.detach().numpy(), np.full((B,), N, dtype=np.int32))
        idx, _, _, _ = grid.query_dense(x, k, r, 1)
    else:
        idx, _, _ = brute_force_query(x, coords.detach().numpy(), k, r, 1)
    nb = torch.from_numpy(idx.astype(np.int64)).reshape(B, N, k)                     # global indices, -1 pad
    found = (nb >= 0).any(dim=-1, keepdim=True)
    own = torch.arange(N)[None, :, None] + (torch.arange(B) * N)[:, None, None]
    padded = torch.full((B, N, k), -1, dtype=torch.long)
    padded[..., :1] = own                                                            # lost points keep themselves (:48-50)
    nb = torch.where(found, nb, padded)
    identity = nb == own
    enough = (nb >= 0).sum(dim=-1, keepdim=True) > 1
    nb = torch.where(identity & enough, torch.full_like(nb, -1), nb).reshape(B * N, k)
    valid = nb >= 0
    owner = torch.arange(B * N)[:, None].expand(B * N, k)[valid]
    cflat, fflat = coords.detach().reshape(B * N, 3), feats.reshape(B * N, -1)
    w = 1.0 / (torch.linalg.norm(cflat[nb[valid]] - cflat[owner], dim=-1) + 1e-5)
    dist = torch.linalg.norm(fflat[nb[valid]] - fflat[owner], ord=1, dim=-1)
    tv = torch.zeros(B * N).index_add_(0, owner, w * dist).reshape(B, N) * weight
    return tv.mean(), tv
