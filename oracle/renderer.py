"""Oracle (CPU, fp32 torch + numpy) restatement of the PointNeRF volume renderer.
TEST INFRASTRUCTURE.  Pinned by tests/golden/render_*.npz (brute-force neighbour branch of the
reference); the voxel-grid neighbour branch is "parity unpinned" (see oracle/voxel_grid.py).

Field weights are a flat dict with the reference's ``state_dict`` keys relative to the Field
module (pointnerf.py:27): ``aggregator.local_field.{0,2,4,6,8}.{weight,bias}``,
``shape_net.{0,2}.*``, ``channel_net.{0,2,4,6,8}.*``.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .voxel_grid import VoxelGridOracle, brute_force_query

Params = Dict[str, torch.Tensor]

DEFAULT_GRID = dict(voxel_size=(0.04, 0.04, 0.04), voxel_scale=(2, 2, 2), kernel_size=(3, 3, 3),
                    max_points_per_voxel=4, max_occ_voxels_per_example=5000,
                    ranges=(-1.0, -1.0, -1.0, 1.0, 1.0, 1.0),          # pointnerf.py:147-153
                    grid_level="scaled")                                # the reading of the absent torch_knnquery source (voxel_grid.py)


# ------------------------------------------------------------------ rays ---------------------
def camera_rays(extr: torch.Tensor, intr: torch.Tensor, res: int):
    """ray_sampler.py:10-49.  extr [V,4,4] world2cam, intr [V,3,3] -> origins, dirs [V,res*res,3];
    ray r = i*res + j looks through pixel centre (x=j+0.5, y=i+0.5)."""
    V = extr.shape[0]
    fx, fy = intr[:, 0, 0:1], intr[:, 1, 1:2]
    cx, cy = intr[:, 0, 2:3], intr[:, 1, 2:3]
    sk = intr[:, 0, 1:2]
    pix = torch.arange(res, dtype=torch.float32, device=intr.device) + 0.5
    y = pix.repeat_interleave(res)[None].expand(V, -1)       # row i
    x = pix.repeat(res)[None].expand(V, -1)                  # column j
    z = torch.ones_like(x)
    xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    yl = (y - cy) / fy * z
    rot_t = extr[:, :3, :3].transpose(1, 2)                  # cam2world rotation
    centre = -torch.matmul(rot_t, extr[:, :3, 3:])           # [V,3,1]
    cam2world = extr.clone()
    cam2world[:, :3, :3] = rot_t
    cam2world[:, :3, 3:] = centre
    hom = torch.stack((xl, yl, z, torch.ones_like(z)), dim=-1)           # [V,R,4]
    world = torch.bmm(cam2world, hom.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
    loc = cam2world[:, :3, 3]
    dirs = F.normalize(world - loc[:, None, :], dim=2)
    return loc[:, None, :].expand(-1, dirs.shape[1], -1).contiguous(), dirs.contiguous()


def ray_box_limits(o: torch.Tensor, d: torch.Tensor, box: float = 1.0):
    """math_utils.py:46-97 + renderer.py:36-43.  o, d [...,3] -> start, end [...,1].
    Rays that miss the cube get (global min start, global max end) of the rays that hit."""
    shape = o.shape[:-1]
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    inv = 1.0 / d
    neg = inv < 0
    lo = torch.where(neg, torch.full_like(o, box), torch.full_like(o, -box))
    hi = torch.where(neg, torch.full_like(o, -box), torch.full_like(o, box))
    t0 = (lo - o) * inv
    t1 = (hi - o) * inv
    ok = torch.ones(o.shape[0], dtype=torch.bool, device=o.device)
    tmin, tmax = t0[:, 0], t1[:, 0]
    ok &= ~((tmin > t1[:, 1]) | (t0[:, 1] > tmax))
    tmin, tmax = torch.max(tmin, t0[:, 1]), torch.min(tmax, t1[:, 1])
    ok &= ~((tmin > t1[:, 2]) | (t0[:, 2] > tmax))
    tmin, tmax = torch.max(tmin, t0[:, 2]), torch.min(tmax, t1[:, 2])
    tmin = torch.where(ok, tmin, torch.full_like(tmin, -1.0))
    tmax = torch.where(ok, tmax, torch.full_like(tmax, -2.0))
    hit = tmax > tmin
    if bool(hit.any()):
        tmin = torch.where(hit, tmin, tmin[hit].min())
        tmax = torch.where(hit, tmax, tmax[hit].max())
    return tmin.reshape(*shape, 1), tmax.reshape(*shape, 1)


def depth_samples(start: torch.Tensor, end: torch.Tensor, S: int) -> torch.Tensor:
    """renderer.py:49-77 (eval mode) + math_utils.py:100-117.  start/end [...,1] -> [...,S]."""
    steps = torch.arange(S, dtype=torch.float32, device=start.device) / (S - 1)
    return start + steps * (end - start)


# ------------------------------------------------------------------ field --------------------
def positional_encoding(x: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """positional_encoder.py:7-23: cat(x, [sin(x_c f_0..f_{n-1}), cos(x_c f_0..)] for c in xyz)."""
    bands = (2 ** torch.arange(n_freqs, device=x.device)) * torch.pi          # fp32
    spec = x[..., None] * bands
    enc = torch.cat((spec.sin(), spec.cos()), dim=-1).flatten(start_dim=-2)
    return torch.cat((x, enc), dim=-1)


def run_mlp(p: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """utils/model.py:22-36 with act=LeakyReLU(0.01), layer_norm=False: Linear layers live at
    even indices; every Linear but the last is followed by the activation."""
    ids = sorted(int(key[len(prefix) + 1:].split(".")[0]) for key in p
                 if key.startswith(prefix + ".") and key.endswith(".weight"))
    for n, i in enumerate(ids):
        x = F.linear(x, p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"])
        if n + 1 < len(ids):
            x = F.leaky_relu(x, 0.01)
    return x


def shade_points(p: Params, nb_idx: torch.Tensor, pts: torch.Tensor, kp_pos: torch.Tensor,
                 kp_feat: torch.Tensor, n_freqs: int = 10):
    """aggregators/mlp.py:36-125 + field.py:113-141 for compact shading points.
    nb_idx [P,k] int64 global indices (-1 pad), pts [P,3], kp_pos [B,N,3], kp_feat [B,N,F]
    -> sigma [P,1] (after softplus(x-1)), rgb [P,3] (after sigmoid), feat [P,256]."""
    P, k = nb_idx.shape
    table = torch.cat((kp_pos, kp_feat), dim=-1).reshape(-1, 3 + kp_feat.shape[-1])
    valid = nb_idx >= 0
    owner = torch.arange(P, device=pts.device)[:, None].expand(P, k)[valid]        # aggregator.py:147-156
    rows = table[nb_idx[valid]]
    rel = pts[owner] - rows[:, :3]
    w = 1.0 / (torch.linalg.norm(rel, dim=-1) + 1e-5)
    local = run_mlp(p, "aggregator.local_field", torch.cat((rows[:, 3:], positional_encoding(rel, n_freqs)), dim=-1))
    wsum = torch.zeros(P, device=pts.device).index_add_(0, owner, w)
    w = w / wsum[owner]
    feat = torch.zeros(P, local.shape[1], device=pts.device).index_add_(0, owner, w[:, None] * local)
    sigma = F.softplus(run_mlp(p, "shape_net", feat) - 1.0)
    rgb = torch.sigmoid(run_mlp(p, "channel_net", feat))
    return sigma, rgb, feat


# ------------------------------------------------------------------ ray march ----------------
def depths_from_points(pts: torch.Tensor, mask: torch.Tensor, o: torch.Tensor, d: torch.Tensor,
                       ray_end: torch.Tensor) -> torch.Tensor:
    """renderer.py:96-110.  pts [Nr,M,3], mask [Nr,M] bool, o/d [Nr,3], ray_end [Nr,1] -> [Nr,M]."""
    dep = torch.nanmean((pts - o[:, None, :]) / d[:, None, :], dim=-1)
    dep = torch.where(mask, dep, torch.full_like(dep, -math.inf))
    dep = torch.cummax(dep, dim=1).values
    return torch.where(dep == -math.inf, ray_end.expand_as(dep), dep)


def ray_march(sigma: torch.Tensor, depths: torch.Tensor, rgb: torch.Tensor, mask: torch.Tensor,
              white_back: bool = True):
    """volume_renderer.py:23-39 + renderer.py:120-185.  sigma/depths/mask [Nr,M] dense, rgb [Nr,M,3]
    dense (zeros at invalid slots) -> mask [Nr,1], depth [Nr,1], channels [Nr,3]."""
    delta = torch.cat((depths[:, 1:] - depths[:, :-1], torch.zeros_like(depths[:, :1])), dim=1)
    alpha = 1.0 - torch.exp(-(sigma * delta))
    trans = torch.cumprod(torch.cat((torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10), dim=1), dim=1)[:, :-1]
    w = alpha * trans
    total = w.sum(dim=1, keepdim=True)
    depth = (w * depths).sum(dim=1, keepdim=True) / total
    depth = torch.nan_to_num(depth, float("inf"))
    if depth.numel() > 0:
        depth = torch.clamp(depth, depths.min(), depths.max())
    chan = ((w * mask)[..., None] * rgb).sum(dim=1)
    if white_back:
        chan = chan + 1.0 - total
    return total, depth, chan


# ------------------------------------------------------------------ end to end ---------------
def render(p: Params, coords: torch.Tensor, feats: torch.Tensor, extr: torch.Tensor, intr: torch.Tensor,
           res: int = 128, S: int = 128, M: int = 50, k: int = 8, r: float = 2.0, mode: str = "grid",
           grid_cfg: Optional[dict] = None, cube_scale: float = 1.0, white_back: bool = True,
           return_aux: bool = False):
    """PointNeRF.render (pointnerf.py:107-131) -> Renderer.forward (renderer.py:202-268) ->
    VolumeRenderer.render (volume_renderer.py:41-92), eval mode (no ray / depth randomisation).
    coords [B,N,3], feats [B,N,F], extr [B,T,4,4], intr [B,T,3,3].
    mode "grid": neighbour search per the VoxelGrid spec, radius = r*max(voxel_size);
    mode "brute": the reference's voxel_grid=None branch, radius = r."""
    B, T = extr.shape[:2]
    o, d = camera_rays(extr.flatten(0, 1).float(), intr.flatten(0, 1).float(), res)
    R = o.shape[1]
    o, d = o.reshape(B, T * R, 3), d.reshape(B, T * R, 3)
    start, end = ray_box_limits(o, d, cube_scale)
    depth = depth_samples(start, end, S)                                  # [B,TR,S]
    x = o[:, :, None, :] + depth[..., None] * d[:, :, None, :]            # [B,TR,S,3]
    xn = x.numpy()
    if mode == "grid":
        grid = VoxelGridOracle(**(grid_cfg or DEFAULT_GRID))
        grid.set_pointset(coords.numpy(), np.full((B,), coords.shape[1], dtype=np.int32))
        idx, loc, nsel, _ = grid.query_dense(xn, k, r, M)
    elif mode == "brute":
        idx, loc, nsel = brute_force_query(xn, coords.numpy(), k, r, M)
    else:
        raise ValueError(mode)
    idx_t = torch.from_numpy(idx.astype(np.int64)).reshape(B * T * R, M, k)
    loc_t = torch.from_numpy(loc).reshape(B * T * R, M, 3)
    slot_valid = (idx_t >= 0).any(dim=-1)                                 # aggregator.py:66-70
    nb = idx_t[slot_valid]                                                # row-major compact order
    pts = loc_t[slot_valid]
    sigma_c, rgb_c, _ = shade_points(p, nb, pts, coords, feats)
    sigma = torch.zeros(B * T * R, M)
    sigma[slot_valid] = sigma_c[:, 0]                                     # field.py:62-68
    rgb = torch.zeros(B * T * R, M, 3)
    rgb[slot_valid] = rgb_c
    dense_pts = torch.zeros(B * T * R, M, 3)
    dense_pts[slot_valid] = pts                                           # field.py:70-75,143
    dep = depths_from_points(dense_pts, slot_valid, o.reshape(-1, 3), d.reshape(-1, 3), end.reshape(-1, 1))
    total, cdepth, chan = ray_march(sigma, dep, rgb, slot_valid, white_back)
    out = {"mask": total.reshape(B, T, R, 1), "depth": cdepth.reshape(B, T, R, 1),
           "channels": chan.reshape(B, T, R, 3)}
    if return_aux:
        out["aux"] = {"rays_o": o, "rays_d": d, "start": start, "end": end, "idx": idx, "loc": loc,
                      "nsel": nsel, "slot_valid": slot_valid, "P": int(slot_valid.sum()),
                      "Q": int((idx_t >= 0).sum()), "sigma": sigma, "rgb": rgb, "depths": dep}
    return out


def unflatten_image(channels: torch.Tensor) -> torch.Tensor:
    """utils/util.py:199-203: [..., R, C] -> [..., C, res, res]."""
    x = channels.transpose(-1, -2)
    side = round(x.shape[-1] ** 0.5)
    return x.reshape(*x.shape[:-1], side, side)


def psnr(pred: torch.Tensor, target: torch.Tensor, data_range: float = 1.0) -> float:
    """skimage.metrics.peak_signal_noise_ratio as used at pointnerf_evaluation.py:254 (float64)."""
    mse = float(((pred.double() - target.double()) ** 2).mean())
    return float("inf") if mse == 0 else 10.0 * math.log10(data_range ** 2 / mse)


def init_field_params(feats_dim: int = 32, seed: int = 0, n_freqs: int = 10, hidden: int = 256, dir_dim: int = 0) -> Params:
    """Synthetic PointNeRF field weights with PyTorch's default nn.Linear init (the reference builds
    its MLPs with utils/model.py:22-36 and never re-initialises them).  dir_dim: extra input columns of the first colour
    layer (use_view_dir: 51 = 3 (1 + 2 x 8 frequencies), fields/mlp.py:33-35)."""
    g = torch.Generator().manual_seed(seed)
    p: Params = {}

    def lin(name, out_f, in_f):
        b = 1.0 / math.sqrt(in_f)
        p[name + ".weight"] = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * b
        p[name + ".bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * b

    d_in = feats_dim + 3 * (1 + 2 * n_freqs)
    dims = [d_in, hidden, hidden, hidden, hidden, hidden]                # pointnerf.py:174-179
    for n in range(5):
        lin(f"aggregator.local_field.{2 * n}", dims[n + 1], dims[n])
    lin("shape_net.0", hidden, hidden)                                    # pointnerf.py:162
    lin("shape_net.2", 1, hidden)
    for n in range(4):                                                    # pointnerf.py:161
        lin(f"channel_net.{2 * n}", hidden, hidden + (dir_dim if n == 0 else 0))
    lin("channel_net.8", 3, hidden)
    return p


def synthetic_cloud(n_points: int = 512, feats_dim: int = 32, batch: int = 1, seed: int = 0):
    """points on an ellipsoid (0.45, 0.20, 0.15), feats ~ N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(batch, n_points, 3, generator=g)
    u = u / u.norm(dim=-1, keepdim=True)
    coords = u * torch.tensor([0.45, 0.20, 0.15])
    feats = torch.randn(batch, n_points, feats_dim, generator=g)
    return coords, feats


def look_at_pose(azim_deg: float, elev_deg: float, radius: float = 1.3) -> torch.Tensor:
    """A world2cam matrix on the radius-1.3 sphere of data/srncars_test_poses.npy (camera looks at
    the origin, +z forward)."""
    az, el = math.radians(azim_deg), math.radians(elev_deg)
    c = torch.tensor([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az),
                      radius * math.sin(el)])
    fwd = -c / c.norm()
    up = torch.tensor([0.0, 0.0, 1.0])
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    rot = torch.stack((right, down, fwd))                                 # rows: camera axes in world
    m = torch.eye(4)
    m[:3, :3] = rot
    m[:3, 3] = -rot @ c
    return m


def srn_intrinsics() -> torch.Tensor:
    """data/srncars_test_intrinsics.npy row 0 (fx=fy=131.25, cx=cy=64; K[2,2]=0 in the file)."""
    return torch.tensor([[131.25, 0.0, 64.0], [0.0, 131.25, 64.0], [0.0, 0.0, 0.0]])
