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
        self.field = field                     # registered twice like the reference (renderer.py:26) -> duplicated keys
        self.cube_scale, self.depth_resolution = cube_scale, depth_resolution
        self.ray_limits = None if ray_limits is None else (float(ray_limits[0]), float(ray_limits[1]))    # renderer.py:44-46
        self.disparity_space_sampling = bool(disparity_space_sampling)                                    # renderer.py:60-68
        self.ray_subsamples, self.white_back = ray_subsamples, white_back
        self.randomize_depth_samples = False
        self.capacity_fraction = 0.25      # compact shading-point buffers: fraction of rays*slots reserved up front ...
        self.sync_free_points = 1 << 23    # ... unless the worst case is at most this many points (eight 128^2 views: 6.6 M):
        #                                    then the buffers take the worst case (44 B of lists + 512 B of workspace per point)
        self.count_pairs = False           # also report the number of (point, neighbour) pairs (an extra reduction + sync, ~8 % of a view)
        # fp16 range guard of the fused shading kernels (include/npcd_hip.h): the kernels raise bits in a device word when an
        # activation left the fp16 range (the reference shades in fp32).  The word comes back as out["shading_status"] -- a host
        # int where the call reads the point count from the device anyway, a device scalar on the sync-free path -- and
        # `check_shading_status` turns a raised bit into a FloatingPointError ("raise"), a RuntimeWarning ("warn") or nothing ("off").
        # "promote" (round 6): the call reads the word (one host read per call, like the point count) and, if a bit is raised,
        # shades the SAME compact lists again in the fp32-class kernels on the GPU (csrc/points_x2.hip: fp32's exponent range,
        # nothing overflows) and marches again -- the caller gets reference-class pixels plus out["shading_promoted"] = True; no CPU
        # path is involved.  Needs the published field architecture (Field.fp32_class_ok()).
        self.range_guard = "warn"
        # None: the fused fp16-operand shading kernels (csrc/shade.hip).  torch.float32: the reference's numerics class -- per-pair
        # layers on the fp32-class matrix-core kernel, heads on fp32 library GEMMs (Field.shade_fp32); one host read of the point
        # count per call, about 4 x the shading time.  PointNeRF.render(mlp_dtype=...) sets it per call.
        self.shade_dtype = None

    def limits(self, t0: torch.Tensor, t1: torch.Tensor):
        """Box limits from the ray kernel, or the fixed (near, far) of `ray_limits` for every ray (renderer.py:36-47)."""
        if self.ray_limits is None:
            return t0, t1
        return torch.full_like(t0, self.ray_limits[0]), torch.full_like(t1, self.ray_limits[1])

    def disparity_depths(self, t0: torch.Tensor, t1: torch.Tensor, jitter=None):
        """renderer.py:60-68: S samples evenly spaced in 1 / depth between the limits, each moved by U(0, 1) of one spacing -- the
        reference jitters this mode always, also in evaluation.  t0 / t1 [...] -> depths [..., S]; `jitter` [..., S] replays
        given draws."""
        S = self.depth_resolution
        u = torch.arange(S, dtype=torch.float32, device=t0.device) / (S - 1)
        u = u + (torch.rand(t0.shape + (S,), device=t0.device) if jitter is None else jitter.to(t0.device)) * (1.0 / (S - 1))
        return 1.0 / (1.0 / t0[..., None] * (1.0 - u) + 1.0 / t1[..., None] * u)

    def forward(self, kp_pos, kp_feat, extr, intr, resolution: int, sample: bool, return_channels: bool = True,
                return_kp_weights: bool = False, knn_mode: int = 0, rng=None):
        """kp_pos [B,N,3], kp_feat [B,N,F], extr [B,T,4,4] world2cam, intr [B,T,3,3] ->
        AttrDict(mask [B,T,R,1], depth [B,T,R,1], channels [B,T,R,3], optional kp_weights [B,T,R,N])   (renderer.py:202-268).
        rng: dictionary of random draws to replay (tests): ray_perm / jitter / valid_perm in training mode (train_path.py),
        `jitter_disp` [B,T,R,S] for disparity-space sampling."""
        if sample or self.randomize_depth_samples:
            # training mode (random ray subset, jittered depths, gradients): npcd.models.pointnerf.train_path
            if return_kp_weights or not return_channels:
                raise NotImplementedError("training-mode rendering returns channels and no key-point weights")
            from .train_path import render_train
            return render_train(self, kp_pos, kp_feat, extr, intr, resolution, sample, rng=rng, knn_mode=knn_mode)
        B, T = extr.shape[:2]
        field, agg = self.field, self.field.aggregator
        grid = agg.voxel_grid
        promoted = False
        M = agg.max_shading_pts
        R = resolution * resolution
        Nr = B * T * R
        # Round 6, opt-in (hr.FUSED_RAYS, NPCD_RENDER_FUSED_RAYS=1; measured no faster: docs/experiments.md R6.6): in the published
        # configuration the rays can be generated INSIDE the neighbour-query launch and the limits of rays that miss the cube fixed inside
        # the march (hr.query_compact_rays / ray_march_compact(fused=...)): two launches per view fewer, bit-identical results.  Fixed ray
        # limits, view directions (their bias table needs the directions before the query) and the dense option paths always generate
        # the rays with the separate kernel.
        rays_in_query = (knn_mode == 0 and not self.disparity_space_sampling and not return_kp_weights and self.ray_limits is None
                         and not field.use_dir and hr.FUSED_RAYS and hr.COMPACT_ORDERED)
        fused_pack = None
        if not rays_in_query:
            o, d, t0, t1 = hr.ray_gen(extr.flatten(0, 1), intr.flatten(0, 1), resolution, self.cube_scale)
            t0, t1 = self.limits(t0, t1)
            rays = (o.view(B, T * R, 3), d.view(B, T * R, 3), t0.view(B, T * R), t1.view(B, T * R))
            dir_bias = field.dir_bias(d.view(-1, 3))           # None unless use_view_dir
        else:
            rays, dir_bias = None, None
        kp_weights = None
        if knn_mode == 0 and not self.disparity_space_sampling and not return_kp_weights:
            # fused path: compact shading-point lists are produced on the device; the shading kernels read the
            # point count from device memory, so nothing round-trips through the host until the result is used
            if dir_bias is not None and not hr.COMPACT_ORDERED:
                raise RuntimeError("use_view_dir needs the ray-ordered compact lists (NPCD_COMPACT_ORDERED=0 is set)")
            worst = B * T * R * M                              # every slot of every ray valid
            # ("promote" decides on the host what to launch next: it takes the path that reads the counters)
            sync_free = worst <= self.sync_free_points and not (self.range_guard == "promote" and self.shade_dtype is None)
            capacity = worst if sync_free else max(4096, int(worst * self.capacity_fraction))
            while True:
                got = None
                if rays_in_query:
                    got = grid.query_compact_rays(agg.k, agg.r, M, extr, intr, resolution, self.cube_scale, self.depth_resolution, capacity,
                                                  points=kp_pos)
                    if got is None:            # (a cube smaller than the grid's range: the separate ray kernel after all)
                        rays_in_query = False
                        o, d, t0, t1 = hr.ray_gen(extr.flatten(0, 1), intr.flatten(0, 1), resolution, self.cube_scale)
                        rays = (o.view(B, T * R, 3), d.view(B, T * R, 3), t0.view(B, T * R), t1.view(B, T * R))
                if got is not None:
                    (o, d, t0, t1), lim, counter, ray_base, _, ray_bits, nb, pts = got
                    fused_pack = (t0.view(-1), lim)
                else:
                    counter, ray_base, _, ray_bits, nb, pts = grid.query_compact(agg.k, agg.r, M, rays, self.depth_resolution, capacity,
                                                                               points=kp_pos)
                point_ray = None
                if dir_bias is not None:       # lists are in ray order: row p belongs to the last ray whose base is <= p
                    point_ray = (torch.searchsorted(ray_base, torch.arange(capacity, dtype=torch.int32, device=ray_base.device),
                                                    right=True) - 1).clamp_(min=0).to(torch.int32)
                if self.shade_dtype == torch.float32:
                    Pn, overflow = counter[:2].tolist()
                    if overflow and capacity < worst:
                        capacity = worst
                        continue
                    s32, c32 = field.shade_fp32(nb[:Pn], pts[:Pn], kp_pos, kp_feat,
                                                None if point_ray is None else d.view(-1, 3)[point_ray[:Pn].long()])
                    sigma = torch.zeros(capacity, dtype=torch.float32, device=nb.device)
                    rgb = torch.zeros((capacity, 3), dtype=torch.float32, device=nb.device)
                    sigma[:Pn], rgb[:Pn] = s32, c32
                else:
                    sigma, rgb = hr.shade_points(field.packed_weights(kp_pos.device), agg.in_dim, nb, pts, kp_pos.reshape(-1, 3),
                                                 kp_feat.reshape(-1, kp_feat.shape[-1]), n_points=counter[:1], n_freqs=agg.n_freqs,
                                                 hidden=field.hid_dim, dir_bias=dir_bias, point_ray=point_ray, status=counter[2:3])
                mask, depth, chan = hr.ray_march_compact(sigma, rgb, ray_bits, pts, ray_base, o.view(-1, 3), d.view(-1, 3), t1.reshape(-1),
                                                         M, self.white_back, fused=fused_pack)
                if sync_free:
                    # buffers sized for the worst case cannot overflow: nothing to check, the call returns without a host
                    # round trip (the point count stays a device scalar) and the next call's launches overlap this one
                    P, shading_status = counter[0], counter[2]
                    break
                P, overflow, shading_status, _ = counter.tolist()      # (the range-guard word rides on the read of the point count)
                if overflow:
                    capacity = worst
                    continue
                if shading_status and self.range_guard == "promote" and self.shade_dtype is None:
                    self._require_promotable()
                    s32, c32 = field.shade_fp32(nb[:P], pts[:P], kp_pos, kp_feat,
                                                None if point_ray is None else d.view(-1, 3)[point_ray[:P].long()])
                    sigma, rgb = torch.zeros_like(sigma), torch.zeros_like(rgb)
                    sigma[:P], rgb[:P] = s32, c32
                    mask, depth, chan = hr.ray_march_compact(sigma, rgb, ray_bits, pts, ray_base, o.view(-1, 3), d.view(-1, 3), t1.reshape(-1),
                                                             M, self.white_back, fused=fused_pack)
                    promoted = True
                break
            n_pairs = int((nb[:int(P)] >= 0).sum()) if self.count_pairs else -1
        else:
            # dense path: the reference's voxel_grid=None branch (knn_mode 1), explicit sample positions (disparity-space sampling)
            # and the key-point weights; per-ray slot tables on the device, compaction by torch indexing (host round trips)
            if self.disparity_space_sampling:
                jit = None if not rng or "jitter_disp" not in rng else rng["jitter_disp"].reshape(B, T * R, self.depth_resolution)
                dep = self.disparity_depths(rays[2], rays[3], jit)                                   # [B, T R, S]
                x = rays[0][:, :, None, :] + dep[..., None] * rays[1][:, :, None, :]
                idx, loc, _, _ = grid.query_dense(agg.k, agg.scaled_r if knn_mode else agg.r, M, x=x.contiguous(), mode=knn_mode,
                                                  points=kp_pos)
            else:
                idx, loc, _, _ = grid.query_dense(agg.k, agg.scaled_r if knn_mode else agg.r, M, rays=rays, S=self.depth_resolution,
                                                  mode=knn_mode, points=kp_pos)
            valid = (idx[..., 0] >= 0).view(Nr, M)
            per_ray = valid.sum(dim=1, dtype=torch.int32)
            base = torch.cumsum(per_ray, dim=0, dtype=torch.int32) - per_ray
            nb = idx.view(Nr, M, agg.k)[valid]                            # compact, row-major over [ray, slot]
            pts = loc.view(Nr, M, 3)[valid]
            point_ray = torch.nonzero(valid)[:, 0].to(torch.int32)        # the ray of every compact point
            status = torch.zeros(1, dtype=torch.int32, device=nb.device)
            if self.shade_dtype == torch.float32:
                sigma, rgb = field.shade_fp32(nb, pts, kp_pos, kp_feat, None if dir_bias is None else d.view(-1, 3)[point_ray.long()])
            else:
                sigma, rgb = field.shade(nb, pts, kp_pos, kp_feat, dir_bias, None if dir_bias is None else point_ray, status=status)
            shading_status = int(status)
            if shading_status and self.range_guard == "promote" and self.shade_dtype is None:
                self._require_promotable()
                sigma, rgb = field.shade_fp32(nb, pts, kp_pos, kp_feat, None if dir_bias is None else d.view(-1, 3)[point_ray.long()])
                promoted = True
            march = (valid, loc.view(Nr, M, 3), base, o.view(-1, 3), d.view(-1, 3), t1.reshape(-1), self.white_back)
            mask, depth, chan = hr.ray_march(sigma, rgb, *march)
            P, n_pairs = int(nb.shape[0]), int((nb >= 0).sum())
            if return_kp_weights:
                kp_weights = self._kp_weights(sigma, rgb, march, nb, pts, point_ray, kp_pos).view(B, T, R, kp_pos.shape[1])
        out = AttrDict(mask=mask.view(B, T, R, 1), depth=depth.view(B, T, R, 1))
        if return_channels:
            out["channels"] = chan.view(B, T, R, 3)
        if kp_weights is not None:
            out["kp_weights"] = kp_weights
        out["num_shading_points"] = P
        out["num_pairs"] = n_pairs
        out["shading_status"] = shading_status
        out["shading_promoted"] = promoted
        out["shading_numerics"] = ("fp32-class (EMULATED fp32: two bf16 halves per operand, 16 mantissa bits, fp32's range)"
                                   if (self.shade_dtype == torch.float32 or promoted) else "fp16 operands, fp32 accumulation")
        if not torch.is_tensor(shading_status) and not promoted:
            self.check_shading_status(shading_status)
        # which neighbour search produced this render: the reading of the (absent) torch_knnquery source, or
        # the reference's in-repo brute-force branch -- so that evaluation logs and saved renders say what they were made with
        out["grid_level"] = "brute_force" if knn_mode else getattr(grid, "grid_level", None)
        return out

    def _require_promotable(self):
        if not self.field.fp32_class_ok():
            raise FloatingPointError("fused fp16 shading left the fp16 range and range_guard='promote' cannot re-shade this field: the "
                                     "fp32-class kernels cover the published architecture only (Field.fp32_class_ok())")

    def check_shading_status(self, status) -> int:
        """The range-guard word of a render (out["shading_status"]: int, or a device scalar after a sync-free call -- reading it
        here waits for that call).  Non-zero: an fp16 activation of the fused shading MLPs overflowed, the pixels are not the
        reference's (fp32) pixels.  Acts as `range_guard` says and returns the word."""
        status = int(status)
        if self.range_guard not in ("warn", "raise", "off", "promote"):
            raise ValueError(f"range_guard {self.range_guard!r}: 'warn', 'raise', 'off' or 'promote'")
        if status and self.range_guard != "off":
            where = [n for bit, n in ((hr.SHADE_NONFINITE_PAIRS, "per-pair aggregator layers"), (hr.SHADE_NONFINITE_HEADS, "density / colour heads"))
                     if status & bit]
            msg = ("fused fp16 shading left the fp16 range (|activation| >= 65,520 -> inf / NaN) in the " + " and the ".join(where) +
                   ": these pixels differ from the reference's fp32 shading; render with mlp_dtype=torch.float32 "
                   "or set renderer.range_guard = 'promote'")
            if self.range_guard in ("raise", "promote"):      # ("promote" reaches here only for a word read AFTER a call: too late to re-shade)
                raise FloatingPointError(msg)
            import warnings
            warnings.warn(msg, RuntimeWarning, stacklevel=3)
        return status

    @staticmethod
    def _kp_weights(sigma, rgb, march, nb, pts, point_ray, kp_pos):
        """renderer.py:177-184 + aggregators/mlp.py:84,93-98: per ray and key point, the sum over the ray's shading points of
        (ray-march weight of the point) x (normalised inverse-distance weight of the pair) -> [Nr, N].  The march weight of a
        point is d channels[ray, 0] / d rgb[point, 0] -- read off the ray-march backward kernel instead of a second march."""
        Nr, N = march[0].shape[0], kp_pos.shape[1]
        with torch.enable_grad():
            rgb_g = rgb.detach().requires_grad_(True)
            _, _, chan = hr.ray_march_train(sigma.detach(), rgb_g, *march)
            w_point = torch.autograd.grad(chan[:, 0].sum(), rgb_g)[0][:, 0]                          # [P]
        pair = nb >= 0
        owner = torch.nonzero(pair)[:, 0]
        flat = nb[pair].long()
        rel = pts[owner] - kp_pos.reshape(-1, 3)[flat]
        w = 1.0 / (rel.norm(dim=-1) + 1e-5)
        norm = torch.zeros(nb.shape[0], device=w.device).index_add_(0, owner, w)
        w = w / norm[owner]
        out = torch.zeros(Nr * N, device=w.device)
        out.index_add_(0, point_ray.long()[owner] * N + flat % N, w_point[owner] * w)
        return out.view(Nr, N)
