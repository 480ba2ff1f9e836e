"""Volume renderer (reference renderers/{renderer,volume_renderer,ray_sampler,math_utils}.py) as one
device-side pipeline: ray generation -> neighbour query -> fused shading -> ray march."""
import torch
import torch.nn as nn

from ...hip import render as hr
from ...utils import AttrDict


class VolumeRenderer(nn.Module):
    def __init__(self, field, cube_scale: float, depth_resolution: int, ray_limits=None, ray_subsamples: int = 0,
                 disparity_space_sampling: bool = False, white_back: bool = False):
        super().__init__()
        if ray_limits is not None or disparity_space_sampling:
            raise NotImplementedError("HIP renderer implements the published configuration (pointnerf.py:181-190)")
        self.field = field                     # registered twice like the reference (renderer.py:26) -> duplicated keys
        self.cube_scale, self.depth_resolution = cube_scale, depth_resolution
        self.ray_subsamples, self.white_back = ray_subsamples, white_back
        self.randomize_depth_samples = False
        self.capacity_fraction = 0.25      # compact shading-point buffers: fraction of rays*slots reserved up front ...
        self.sync_free_points = 1 << 23    # ... unless the worst case is at most this many points (eight 128^2 views: 6.6 M):
        #                                    then the buffers take the worst case (44 B of lists + 512 B of workspace per point)
        self.count_pairs = False           # also report the number of (point, neighbour) pairs (an extra reduction + sync, ~8 % of a view)

    def forward(self, kp_pos, kp_feat, extr, intr, resolution: int, sample: bool, return_channels: bool = True,
                return_kp_weights: bool = False, knn_mode: int = 0, rng=None):
        """kp_pos [B,N,3], kp_feat [B,N,F], extr [B,T,4,4] world2cam, intr [B,T,3,3] ->
        AttrDict(mask [B,T,R,1], depth [B,T,R,1], channels [B,T,R,3])   (renderer.py:202-268)."""
        if sample or self.randomize_depth_samples:
            # training mode (random ray subset, jittered depths, gradients): npcd.models.pointnerf.train_path
            if return_kp_weights or not return_channels:
                raise NotImplementedError("training-mode rendering returns channels and no key-point weights")
            from .train_path import render_train
            return render_train(self, kp_pos, kp_feat, extr, intr, resolution, sample, rng=rng, knn_mode=knn_mode)
        if return_kp_weights:
            raise NotImplementedError("return_kp_weights")
        B, T = extr.shape[:2]
        agg = self.field.aggregator
        grid = agg.voxel_grid
        o, d, t0, t1 = hr.ray_gen(extr.flatten(0, 1), intr.flatten(0, 1), resolution, self.cube_scale)
        R = o.shape[1]
        rays = (o.view(B, T * R, 3), d.view(B, T * R, 3), t0.view(B, T * R), t1.view(B, T * R))
        M = agg.max_shading_pts
        if knn_mode == 0:
            # fused path: compact shading-point lists are produced on the device; the shading kernels read the
            # point count from device memory, so nothing round-trips through the host until the result is used
            worst = B * T * R * M                              # every slot of every ray valid
            sync_free = worst <= self.sync_free_points
            capacity = worst if sync_free else max(4096, int(worst * self.capacity_fraction))
            while True:
                counter, ray_base, _, ray_bits, nb, pts = grid.query_compact(agg.k, agg.r, M, rays, self.depth_resolution, capacity,
                                                                           points=kp_pos)
                sigma, rgb = hr.shade_points(self.field.packed_weights(kp_pos.device), agg.in_dim, nb, pts, kp_pos.reshape(-1, 3),
                                             kp_feat.reshape(-1, kp_feat.shape[-1]), n_points=counter[:1], n_freqs=agg.n_freqs,
                                             hidden=self.field.hid_dim)
                mask, depth, chan = hr.ray_march_compact(sigma, rgb, ray_bits, pts, ray_base, o.view(-1, 3), d.view(-1, 3), t1.reshape(-1),
                                                         M, self.white_back)
                if sync_free:
                    # buffers sized for the worst case cannot overflow: nothing to check, the call returns without a host
                    # round trip (the point count stays a device scalar) and the next call's launches overlap this one
                    P = counter[0]
                    break
                P, overflow = counter.tolist()
                if not overflow:
                    break
                capacity = worst
            n_pairs = int((nb[:int(P)] >= 0).sum()) if self.count_pairs else -1
        else:
            idx, loc, _, _ = grid.query_dense(agg.k, agg.scaled_r, M, rays=rays, S=self.depth_resolution, mode=knn_mode, points=kp_pos)
            valid = (idx[..., 0] >= 0).view(B * T * R, M)
            per_ray = valid.sum(dim=1, dtype=torch.int32)
            base = torch.cumsum(per_ray, dim=0, dtype=torch.int32) - per_ray
            nb = idx.view(B * T * R, M, agg.k)[valid]                     # compact, row-major over [ray, slot]
            pts = loc.view(B * T * R, M, 3)[valid]
            sigma, rgb = self.field.shade(nb, pts, kp_pos, kp_feat)
            mask, depth, chan = hr.ray_march(sigma, rgb, valid, loc.view(B * T * R, M, 3), base, o.view(-1, 3), d.view(-1, 3),
                                             t1.reshape(-1), self.white_back)
            P, n_pairs = int(nb.shape[0]), int((nb >= 0).sum())
        out = AttrDict(mask=mask.view(B, T, R, 1), depth=depth.view(B, T, R, 1))
        if return_channels:
            out["channels"] = chan.view(B, T, R, 3)
        out["num_shading_points"] = P
        out["num_pairs"] = n_pairs
        return out
