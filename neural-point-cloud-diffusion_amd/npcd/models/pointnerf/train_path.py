"""Training-mode rendering (stage-1 autodecoder training): a few hundred randomly chosen rays
per view with jittered depth samples, differentiable w.r.t. the neural-point features and the field weights.

First version of this row.  Geometry -- ray generation and the neighbour query -- runs on the HIP kernels
(npcd.hip.render); the differentiable part (per-pair MLP, aggregation, heads, ray march: ~1e6 (point, neighbour) pairs per
step at the reference's 8 objects x 50 views x 112 rays) is written with torch operators on the device so that autograd
provides the backward; its GEMMs are library calls.  Nothing here runs on the CPU: the neighbour query fails loudly
without the HIP library.

Reference: renderers/renderer.py:49-77,96-110,120-185,202-268; renderers/volume_renderer.py:23-92; fields/field.py:77-152;
fields/aggregators/aggregator.py:78-119; fields/aggregators/mlp.py:36-125; fields/positional_encoder.py:7-23;
renderers/math_utils.py:46-97.  The reference draws its random numbers inline; every draw can be injected (`rng`
dictionary: ray_perm, jitter, valid_perm) so that tests can replay the reference's numbers.
"""
import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from ...hip import render as hr
from ...utils import AttrDict


# ------------------------------------------------------------------------------------------------- rays
def box_limits(o: torch.Tensor, d: torch.Tensor, half: float):
    """Slab test of rays against the cube [-half, half]^3.  o, d [..., 3] -> start, end [..., 1]; rays that miss get the
    smallest start / largest end among the rays that hit (one global reduction, as the reference does)."""
    inv = 1.0 / d
    near = (torch.where(inv < 0, half, -half) - o) * inv
    far = (torch.where(inv < 0, -half, half) - o) * inv
    t0, t1 = near[..., 0], far[..., 0]
    ok = torch.ones_like(t0, dtype=torch.bool)
    for a in (1, 2):
        ok = ok & ~((t0 > far[..., a]) | (near[..., a] > t1))
        t0, t1 = torch.maximum(t0, near[..., a]), torch.minimum(t1, far[..., a])
    t0 = torch.where(ok, t0, torch.full_like(t0, -1.0))
    t1 = torch.where(ok, t1, torch.full_like(t1, -2.0))
    hit = t1 > t0
    if bool(hit.any()):
        t0 = torch.where(hit, t0, t0[hit].min())
        t1 = torch.where(hit, t1, t1[hit].max())
    return t0.unsqueeze(-1), t1.unsqueeze(-1)


def jittered_depths(start: torch.Tensor, end: torch.Tensor, S: int, jitter: Optional[torch.Tensor]):
    """start/end [..., 1] -> depths [..., S]: S evenly spaced samples, each moved forward by U(0,1) of one spacing."""
    steps = torch.arange(S, dtype=torch.float32, device=start.device) / (S - 1)
    dep = start + steps * (end - start)
    if jitter is not None:
        dep = dep + jitter * ((end - start) / (S - 1))
    return dep


# ------------------------------------------------------------------------------------------------- field (autograd)
def positional_encoding(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """[x, sin(x_c 2^i pi) (i < n), cos(x_c 2^i pi) (i < n) for c in x, y, z] -> 3 + 6 n columns."""
    bands = (2 ** torch.arange(n_freqs, device=x.device)) * torch.pi
    spec = x[..., None] * bands
    return torch.cat((x, torch.cat((spec.sin(), spec.cos()), dim=-1).flatten(start_dim=-2)), dim=-1)


def shade_autograd(field, nb_idx: torch.Tensor, pts: torch.Tensor, kp_pos: torch.Tensor, kp_feat: torch.Tensor):
    """Compact shading points -> sigma [P] (softplus(x - 1)), rgb [P, 3] (sigmoid), differentiable w.r.t. kp_feat and the
    field's parameters.  nb_idx [P, k] global point indices (-1 pad), pts [P, 3]; positions are constants (the point
    coordinates are frozen in stage 1: pointnerf.py:24,68, aggregators/mlp.py:58-59)."""
    agg = field.aggregator
    P, k = nb_idx.shape
    valid = nb_idx >= 0
    owner = torch.arange(P, device=pts.device)[:, None].expand(P, k)[valid]
    flat = nb_idx[valid]
    pos = kp_pos.detach().reshape(-1, 3)[flat]
    feat = kp_feat.reshape(-1, kp_feat.shape[-1])[flat]
    rel = pts[owner] - pos
    w = 1.0 / (torch.linalg.norm(rel, dim=-1) + 1e-5)
    local = agg.local_field(torch.cat((feat, positional_encoding(rel, agg.n_freqs)), dim=-1))
    w = w / torch.zeros(P, device=pts.device, dtype=w.dtype).index_add_(0, owner, w)[owner]
    agg_feat = torch.zeros(P, local.shape[1], device=pts.device, dtype=local.dtype).index_add_(0, owner, w[:, None] * local)
    sigma = F.softplus(field.shape_net(agg_feat) - 1.0)[:, 0]
    rgb = torch.sigmoid(field.channel_net(agg_feat))
    return sigma, rgb


# ------------------------------------------------------------------------------------------------- ray march (autograd)
def depths_from_points(pts, mask, o, d, ray_end):
    """pts [Nr, M, 3] (zeros at invalid slots), mask [Nr, M], o / d [Nr, 3], ray_end [Nr, 1] -> depth per slot [Nr, M]:
    invalid slots repeat the last valid depth before them, leading invalid slots take the ray end."""
    dep = torch.nanmean((pts - o[:, None, :]) / d[:, None, :], dim=-1)
    dep = torch.where(mask, dep, torch.full_like(dep, -math.inf))
    dep = torch.cummax(dep, dim=1).values
    return torch.where(dep == -math.inf, ray_end.expand_as(dep), dep)


def ray_march(sigma, depths, rgb, mask, white_back: bool):
    """sigma / depths / mask [Nr, M] dense, rgb [Nr, M, 3] -> opacity [Nr, 1], expected depth [Nr, 1], colour [Nr, 3]."""
    delta = torch.cat((depths[:, 1:] - depths[:, :-1], torch.zeros_like(depths[:, :1])), dim=1)
    alpha = 1.0 - torch.exp(-(sigma * delta))
    trans = torch.cumprod(torch.cat((torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10), dim=1), dim=1)[:, :-1]
    w = alpha * trans
    total = w.sum(dim=1, keepdim=True)
    depth = torch.nan_to_num((w * depths).sum(dim=1, keepdim=True) / total, float("inf"))
    if depth.numel() > 0:
        depth = torch.clamp(depth, depths.min(), depths.max())
    chan = ((w * mask)[..., None] * rgb).sum(dim=1)
    if white_back:
        chan = chan + 1.0 - total
    return total, depth, chan


# ------------------------------------------------------------------------------------------------- the training-mode forward
def render_train(renderer, kp_pos, kp_feat, extr, intr, resolution: int, sample: bool, rng: Optional[Dict] = None,
                 knn_mode: int = 0) -> AttrDict:
    """VolumeRenderer.forward for sample=True and / or jittered depths.  kp_pos [B,N,3], kp_feat [B,N,F] (may require grad),
    extr [B,T,4,4], intr [B,T,3,3] -> AttrDict(mask [B,T,n,1], depth [B,T,n,1], channels [B,T,n,3], ray_idx [B,T,n,1])."""
    rng = rng or {}
    field, agg = renderer.field, renderer.field.aggregator
    B, T = extr.shape[:2]
    dev = kp_pos.device
    R = resolution * resolution
    if renderer.ray_subsamples and sample:          # the same random rays for every (object, view) instance
        perm = rng["ray_perm"].to(dev) if "ray_perm" in rng else torch.randperm(R, device=dev)
        ray_ids = perm[:renderer.ray_subsamples].long()
        o, d, _, _ = hr.ray_gen(extr.flatten(0, 1), intr.flatten(0, 1), resolution, renderer.cube_scale, pixel_ids=ray_ids)
    else:
        ray_ids = torch.arange(R, device=dev)
        o, d, _, _ = hr.ray_gen(extr.flatten(0, 1), intr.flatten(0, 1), resolution, renderer.cube_scale)
    o, d = o.view(B, T, -1, 3), d.view(B, T, -1, 3)
    Rs = o.shape[2]
    start, end = box_limits(o, d, renderer.cube_scale)
    S = renderer.depth_resolution
    jitter = None
    if renderer.randomize_depth_samples:
        jitter = rng["jitter"].to(dev).reshape(B, T, Rs, S) if "jitter" in rng else torch.rand(B, T, Rs, S, device=dev)
    dep = jittered_depths(start, end, S, jitter)
    x = o[..., None, :] + dep[..., None] * d[..., None, :]                       # [B,T,Rs,S,3]
    M = agg.max_shading_pts
    grid = agg.voxel_grid
    idx, loc, _, _ = grid.query_dense(agg.k, agg.scaled_r if knn_mode else agg.r, M, x=x.reshape(B, T * Rs, S, 3).contiguous(),
                                      mode=knn_mode, points=kp_pos.detach())
    idx = idx.view(B * T, Rs, M, agg.k).long()
    loc = loc.view(B * T, Rs, M, 3)
    slot_valid = idx[..., 0] >= 0                                                # lists are sorted: valid iff first entry valid
    if sample:
        ray_sel = agg.select_valid_rays(slot_valid, rng.get("valid_perm"))      # [B*T, Rs] bool, same count per instance
    else:
        ray_sel = torch.ones(B * T, Rs, dtype=torch.bool, device=dev)
    n = int(ray_sel[0].sum())
    idx_s, loc_s, valid_s = idx[ray_sel], loc[ray_sel], slot_valid[ray_sel]      # [B*T*n, M, ...] in ascending ray order
    nb, pts = idx_s[valid_s], loc_s[valid_s]
    sigma_c, rgb_c = shade_autograd(field, nb, pts, kp_pos, kp_feat)
    rows = torch.nonzero(valid_s, as_tuple=True)
    sigma = torch.zeros(valid_s.shape, device=dev, dtype=sigma_c.dtype).index_put(rows, sigma_c)
    rgb = torch.zeros(valid_s.shape + (3,), device=dev, dtype=rgb_c.dtype).index_put(rows, rgb_c)
    dense_pts = torch.zeros(valid_s.shape + (3,), device=dev).index_put(rows, pts)
    o_s, d_s, end_s = o.reshape(B * T, Rs, 3)[ray_sel], d.reshape(B * T, Rs, 3)[ray_sel], end.reshape(B * T, Rs, 1)[ray_sel]
    depths = depths_from_points(dense_pts, valid_s, o_s, d_s, end_s)
    total, cdepth, chan = ray_march(sigma, depths, rgb, valid_s, renderer.white_back)
    out = AttrDict(mask=total.view(B, T, n, 1), depth=cdepth.view(B, T, n, 1), channels=chan.view(B, T, n, 3))
    if sample:
        out["ray_idx"] = ray_ids[None, :].expand(B * T, Rs)[ray_sel].view(B, T, n, 1)
    out["num_shading_points"] = int(nb.shape[0])
    out["num_pairs"] = int((nb >= 0).sum())
    return out
