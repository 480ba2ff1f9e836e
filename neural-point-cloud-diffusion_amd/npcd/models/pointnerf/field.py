"""PointNeRF field: neighbour aggregator + density / colour heads, evaluated by the fused HIP
shading kernels.  Parameter names follow the reference:

    field.aggregator.local_field.{0,2,4,6,8}   Linear(F+63,256), 3x Linear(256,256), Linear(256,256)
    field.shape_net.{0,2}                      Linear(256,256), Linear(256,1)
    field.channel_net.{0,2,4,6,8}              4x Linear(256,256), Linear(256,3)

(reference: fields/field.py, fields/mlp.py, fields/aggregators/{aggregator,mlp}.py, utils/model.py:22-36)."""
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from ...hip import render as hr


def define_mlp(dims, d_in, d_out=None, act="LeakyReLU"):
    """Linear layers at even indices, activations at odd ones (utils/model.py:22-36, layer_norm=False)."""
    mods, cur = [], d_in
    for dim in dims:
        mods += [nn.Linear(cur, dim), getattr(nn, act)(inplace=True)]
        cur = dim
    if d_out is not None:
        mods.append(nn.Linear(cur, d_out))
    return nn.Sequential(*mods)


class Aggregator(nn.Module):
    """Holds the per-pair MLP and the neighbour-query settings (aggregator.py:12-23, aggregators/mlp.py:13-34)."""

    def __init__(self, in_dim, voxel_grid, k, r, max_shading_pts, ray_subsamples, out_dim, n_freqs, layers,
                 activation="LeakyReLU", layer_norm=False, freq_mult=1, **unused):
        super().__init__()
        assert k > 0, "k for kNN has to be greater than zero"
        if layer_norm or freq_mult != 1:
            raise NotImplementedError("the HIP shading kernels implement the reference configuration "
                                      "(layer_norm=False, freq_mult=1; pointnerf.py:174-179)")
        self.in_dim, self.voxel_grid = in_dim, voxel_grid
        self.k, self.r, self.max_shading_pts = k, r, max_shading_pts
        self.scaled_r = r if voxel_grid is None else r * max(voxel_grid.vsize_tup)
        self.ray_subsamples, self.out_dim, self.n_freqs = ray_subsamples, out_dim, n_freqs
        self.local_field = define_mlp(layers, in_dim + 3 * (1 + 2 * n_freqs), out_dim, activation)

    # ---- reference helper surface (used by neural_point_cloud_tv_loss.py:44,62,64) ----------------
    def query_keypoints(self, x: torch.Tensor, kp_pos: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """x [B,T,R,S,3], kp_pos [B,N,3] -> neighbor_idx [P,k] int64 (-1 pad), shading_pts [P,3],
        mask [B,T,R,M,1] bool (aggregator.py:25-76).  Requires a preceding voxel_grid.set_pointset."""
        B, T, R, S = x.shape[:4]
        M = self.max_shading_pts
        if self.voxel_grid is None:
            grid = hr.HipVoxelGrid((0.04,) * 3, (2,) * 3, (3,) * 3, 4, 5000, (-1, -1, -1, 1, 1, 1))
            idx, loc, _, nsel = grid.query_dense(self.k, self.r, M, x=x.flatten(1, 2), mode=1, points=kp_pos)
        else:
            idx, loc, _, nsel = self.voxel_grid.query_dense(self.k, self.r, M, x=x.flatten(1, 2), mode=0)
        valid = idx[..., 0] >= 0                                  # lists are sorted: slot valid iff first entry valid
        return idx[valid].long(), loc[valid], valid.view(B, T, R, M, 1)

    def select_valid_rays(self, slot_valid: torch.Tensor, valid_perm: Optional[torch.Tensor] = None) -> torch.Tensor:
        """slot_valid [I, R, M] bool -> ray_sample_mask [I, R] bool: the same number min(min_i #valid_i, ray_subsamples) of
        rays per instance, drawn from a random shuffle of each instance's rays that have a valid shading slot
        (aggregator.py:88-108).  `valid_perm` replays a given shuffle of the row-major (instance, ray) list."""
        valid = slot_valid.any(dim=-1)
        pairs = torch.nonzero(valid)
        perm = torch.randperm(pairs.shape[0], device=valid.device) if valid_perm is None else valid_perm.to(valid.device).long()
        pairs = pairs[perm]
        rays = pairs[torch.sort(pairs[:, 0], stable=True).indices, 1]          # regrouped by instance, shuffled order kept
        num_valid = valid.sum(dim=-1)
        n = int(min(int(num_valid.min()), self.ray_subsamples))
        start = torch.cumsum(num_valid, 0) - num_valid
        sel = rays[(start[:, None] + torch.arange(n, device=valid.device)[None, :]).reshape(-1)].reshape(valid.shape[0], n)
        return torch.zeros_like(valid).scatter_(1, sel, True)

    def subsample_valid_rays(self, neighbor_idx, shading_pts, mask, valid_perm=None):
        """Reference signature (aggregator.py:78-119): neighbor_idx [P,k], shading_pts [P,3], mask [B,T,R,M,1] ->
        (neighbor_idx_samples, shading_pts_samples, sampled_mask [B,T,n,M,1], ray_sample_mask [B,T,R])."""
        B, T, R, M = mask.shape[:4]
        flat = mask.reshape(B * T, R, M)
        ray_sel = self.select_valid_rays(flat, valid_perm)
        pts_sel = ray_sel[:, :, None].expand_as(flat)[flat]                     # per valid shading point: is its ray kept?
        n = int(ray_sel[0].sum())
        return neighbor_idx[pts_sel], shading_pts[pts_sel], flat[ray_sel].view(B, T, n, M, 1), ray_sel.view(B, T, R)

    @staticmethod
    def get_keypoint_data(neighbor_idx, mask, kp_pos=None, kp_feat=None) -> Dict[str, torch.Tensor]:
        data = torch.cat([t for t in (kp_pos, kp_feat) if t is not None], dim=-1)
        rows = data.reshape(-1, data.shape[-1])[neighbor_idx.clamp_min(0)][mask]
        out = {}
        if kp_pos is not None:
            out["pos"], rows = rows[:, :3], rows[:, 3:]
        if kp_feat is not None:
            out["feat"] = rows
        return out

    @staticmethod
    def mask_to_batch_ray_idx(valid_neighbor_mask):
        n = valid_neighbor_mask.shape[0]
        return torch.arange(n, device=valid_neighbor_mask.device)[:, None].expand_as(valid_neighbor_mask)[valid_neighbor_mask]


def encode_dir(d: torch.Tensor, n_freqs: int) -> torch.Tensor:
    """PositionalEncoder1D of the ray directions (utils/positional_encoder.py:16-20): [d, per coordinate sin(d_c 2^i pi) (i < n),
    cos(d_c 2^i pi) (i < n)] -> 3 (1 + 2 n) columns; n = 0: the directions as they are (fields/mlp.py:30)."""
    if n_freqs <= 0:
        return d
    spec = d[..., None] * ((2 ** torch.arange(n_freqs, device=d.device)) * torch.pi)
    return torch.cat((d, torch.cat((spec.sin(), spec.cos()), dim=-1).flatten(start_dim=-2)), dim=-1)


class Field(nn.Module):
    def __init__(self, in_dim: int, voxel_grid, aggregator: dict, feat_freqs=0, dir_freqs=8, channel_layers=(256,) * 4,
                 shape_layers=(256,), activation="LeakyReLU", layer_norm=False, nerf=True, use_dir=False, **unused):
        super().__init__()
        if feat_freqs or layer_norm or not nerf:
            raise NotImplementedError("HIP shading implements feat_freqs=0, layer_norm=False, nerf=True (pointnerf.py:155-165)")
        self.aggregator = Aggregator(in_dim, voxel_grid, **aggregator["kwargs"])
        self.hid_dim = self.aggregator.out_dim
        self.nerf, self.use_dir, self.dir_freqs = nerf, use_dir, dir_freqs
        # use_view_dir (fields/mlp.py:30-36,67-70): the colour head sees [feat | enc(ray direction)]
        self.dir_dim = (3 * (1 + 2 * dir_freqs) if dir_freqs > 0 else 3) if use_dir else 0
        self.channel_net = define_mlp(list(channel_layers), self.hid_dim + self.dir_dim, 3, activation)
        self.shape_net = define_mlp(list(shape_layers), self.hid_dim, 1, activation)
        self._pack: Optional[torch.Tensor] = None
        self._pack_key = None

    def packed_weights(self, device) -> torch.Tensor:
        """fp16 fragment-ordered copy of all 12 Linear layers; rebuilt when a parameter changed.  With use_dir the first colour
        layer is packed without its direction columns (those enter through dir_bias)."""
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._pack is None or key != self._pack_key:
            state = self.state_dict()
            if self.use_dir:
                state = dict(state)
                state["channel_net.0.weight"] = state["channel_net.0.weight"][:, :self.hid_dim]
            self._pack = hr.pack_field_weights(state, self.aggregator.in_dim, device, self.aggregator.n_freqs, self.hid_dim)
            self._pack_key = key
        return self._pack

    def dir_bias(self, rays_d: torch.Tensor) -> Optional[torch.Tensor]:
        """rays_d [n_rays, 3] -> [n_rays, hid] fp32: the view-direction part of the first colour layer's pre-activation, the same
        for every shading point of a ray (enc(d) . W[:, hid:]^T); None without use_dir."""
        if not self.use_dir:
            return None
        w = self.channel_net[0].weight[:, self.hid_dim:]
        return (encode_dir(rays_d.float(), self.dir_freqs) @ w.detach().float().t()).contiguous()

    def fp32_class_ok(self) -> bool:
        """Can shade_fp32 run this field?  (the published aggregator network: pointnerf.py:174-179)"""
        from .train_path import FP32_CLASS, fused_pair_mlp_precision
        return fused_pair_mlp_precision(self, FP32_CLASS) == hr.PAIR_MLP_X2

    @torch.no_grad()
    def shade_fp32(self, nb_idx, pts, kp_pos, kp_feat, point_dir=None):
        """The same shading in the reference's numerics class (eval_pointnerf.py / eval_diffusion.py run the field in fp32): the four
        non-linear per-pair layers + the weighted mean on the fp32-class matrix-core kernel (csrc/pairs_mlp.hip precision 1: every
        operand as two bf16 halves, ~1e-5 relative per product, fp32's exponent range -- nothing overflows that fp32 holds), the
        linear fifth layer and both heads on the P points as fp32 library GEMMs.  nb_idx [P, k] (-1 pad, valid entries first),
        pts [P, 3]; point_dir [P, 3] with use_dir.  -> sigma [P], rgb [P, 3].  Roughly 4 x the time of the fp16 kernels."""
        import torch.nn.functional as F
        agg, lf = self.aggregator, self.aggregator.local_field
        if nb_idx.shape[0] == 0:          # no shading point at all (every ray misses the cloud)
            return torch.zeros(0, dtype=torch.float32, device=pts.device), torch.zeros((0, 3), dtype=torch.float32, device=pts.device)
        import os
        fused_pairs = not os.environ.get("NPCD_FP32_PAIRS_TRAINING_KERNEL") and agg.in_dim in (32, 128) and nb_idx.shape[1] <= 8
        if fused_pairs:
            # forward-only kernel of the same numerics on the compact lists as they are (csrc/points_x2.hip, round 5);
            # NPCD_FP32_PAIRS_TRAINING_KERNEL=1 = the training kernel's forward below (csrc/pairs_mlp.hip, precision 1)
            keyp = (str(pts.device),) + tuple((lf[i].weight.data_ptr(), lf[i].weight._version, lf[i].bias._version) for i in (0, 2, 4, 6))
            if getattr(self, "_packp2_key", None) != keyp:
                self._packp2 = hr.pairs_x2_pack({k: v for k, v in self.state_dict().items()}, agg.in_dim, pts.device)
                self._packp2_key = keyp
            G = hr.pairs_x2(self._packp2, agg.in_dim, nb_idx, pts, kp_pos.reshape(-1, 3), kp_feat.reshape(-1, kp_feat.shape[-1]))
        else:
            nb = nb_idx.long()
            valid = nb >= 0
            cnt = valid.sum(dim=1)
            off = torch.cumsum(cnt, 0) - cnt
            key = (str(pts.device),) + tuple((lf[i].weight.data_ptr(), lf[i].weight._version, lf[i].bias._version) for i in (0, 2, 4, 6))
            if getattr(self, "_pack32_key", None) != key:
                self._pack32 = hr.pair_mlp_pack([lf[i].weight for i in (0, 2, 4, 6)], [lf[i].bias for i in (0, 2, 4, 6)], agg.in_dim, hr.PAIR_MLP_X2,
                                                pts.device)
                self._pack32_key = key
            G = hr.pair_mlp_forward_raw(kp_feat.reshape(-1, kp_feat.shape[-1]), None, None, nb, pts, kp_pos.reshape(-1, 3), off, 0,
                                        hr.PAIR_MLP_X2, save=False, wpack=self._pack32)[0]
        from .train_path import x2_linear_forward

        import os
        _X2_HEADS = bool(os.environ.get("NPCD_STAGE1_X2_HEADS"))      # (opt-in, measured slower in training: train_path._X2Linear)

        def mlp(seq, x):      # fp32 library GEMMs; with the switch the wide layers as three cross products of split bf16 operands
            for m in seq:
                if _X2_HEADS and isinstance(m, nn.Linear) and m.out_features >= 16 and m.in_features % 8 == 0 and x.shape[0] >= 4096:
                    x = x2_linear_forward(x, m.weight, m.bias, save=False)[0]
                else:
                    x = m(x)
            return x
        if not os.environ.get("NPCD_FP32_HEADS_LIBRARY"):
            # the fifth layer and both heads fused, same numerics class as the per-pair layers (csrc/points_x2.hip; round 5: ~0.2 ms
            # against ~0.8 ms of fp32 library GEMMs per 128 x 128 view).  NPCD_FP32_HEADS_LIBRARY=1 = the fp32 GEMMs below.
            pw = [p for m in (lf[8], *self.shape_net, *self.channel_net) if isinstance(m, nn.Linear) for p in (m.weight, m.bias)]
            key2 = (str(pts.device),) + tuple((p.data_ptr(), p._version) for p in pw)
            if getattr(self, "_packx2_key", None) != key2:
                self._packx2 = hr.points_x2_pack({k: v for k, v in self.state_dict().items()}, pts.device)
                self._packx2_key = key2
            db = ray = None
            if self.use_dir:          # the direction columns of the first colour layer, per point, as fp32 bias rows
                w0 = self.channel_net[0].weight
                db = encode_dir(point_dir.float(), self.dir_freqs) @ w0[:, self.hid_dim:].float().t()
                ray = torch.arange(G.shape[0], dtype=torch.int32, device=G.device)
            return hr.points_x2(self._packx2, G, db, ray)
        with torch.autocast("cuda", enabled=False):
            feat = mlp([lf[8]], G)
            # (a point without neighbours aggregates to zero and sees the biases only, like the fused kernels and aggregators/mlp.py:60-62)
            chan_in = feat
            if self.use_dir:          # (the first colour layer then has 256 + 51 input columns and stays a plain fp32 GEMM)
                chan_in = torch.cat((feat, encode_dir(point_dir.float(), self.dir_freqs)), dim=-1)
            sigma = F.softplus(mlp(self.shape_net, feat) - 1.0)[:, 0]
            rgb = torch.sigmoid(mlp(self.channel_net, chan_in))
        return sigma.contiguous(), rgb.contiguous()

    def shade(self, nb_idx, pts, kp_pos, kp_feat, dir_bias=None, point_ray=None, status=None):
        """compact shading points -> sigma [P] (softplus(x-1) applied), rgb [P,3] (sigmoid applied).  With use_dir: dir_bias
        [n_rays, hid] (Field.dir_bias) and point_ray [P] int32, the ray of every compact point."""
        return hr.shade_points(self.packed_weights(pts.device), self.aggregator.in_dim, nb_idx, pts,
                               kp_pos.reshape(-1, 3), kp_feat.reshape(-1, kp_feat.shape[-1]),
                               n_freqs=self.aggregator.n_freqs, hidden=self.hid_dim, dir_bias=dir_bias, point_ray=point_ray, status=status)
