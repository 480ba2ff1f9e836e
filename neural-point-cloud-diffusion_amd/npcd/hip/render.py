"""Render-path operators on the HIP kernels (csrc/geometry.hip, csrc/shade.hip)."""
import ctypes
import os
from typing import Optional, Sequence

import torch

from . import GridParams, check, lib, ptr, require_gpu, stream_ptr

_i32, _f32 = torch.int32, torch.float32


GRID_LEVELS = {"fine": 0, "scaled": 1}
# Which reading of the (unavailable) torch_knnquery source the grid follows when the caller does not say: DESIGN.md section 3.
DEFAULT_GRID_LEVEL = os.environ.get("NPCD_GRID_LEVEL", "scaled")


def make_grid_params(voxel_size, voxel_scale, kernel_size, max_points_per_voxel, max_occ_voxels_per_example, ranges,
                     grid_level=None) -> GridParams:
    g = GridParams()
    level = DEFAULT_GRID_LEVEL if grid_level is None else grid_level
    if level not in GRID_LEVELS:
        raise ValueError(f"grid_level must be one of {sorted(GRID_LEVELS)}; got {level!r}")
    g.grid_level = GRID_LEVELS[level]
    for a in range(3):
        g.voxel_size[a] = float(voxel_size[a])
        g.voxel_scale[a] = int(voxel_scale[a])
        g.kernel_size[a] = int(kernel_size[a])
        g.range_min[a] = float(ranges[a])
        g.range_max[a] = float(ranges[3 + a])
        g.dims[a] = int(round((float(ranges[3 + a]) - float(ranges[a])) / float(voxel_size[a])))   # host, float64
        g.cdims[a] = (g.dims[a] + g.voxel_scale[a] - 1) // g.voxel_scale[a]
    g.max_points_per_voxel = int(max_points_per_voxel)
    g.max_occ_voxels_per_example = int(max_occ_voxels_per_example)
    return g


def ray_gen(extr: torch.Tensor, intr: torch.Tensor, res: int, box: float = 1.0, pixel_ids=None):
    """extr [V,4,4] world2cam, intr [V,3,3] -> rays_o, rays_d [V,R,3], t0, t1 [V,R] (R = res*res, or len(pixel_ids) when a
    subset of row-major pixel numbers is given).  ray_sampler.py:10-49 + renderer.py:36-47 (rays that miss the cube get the
    global limits of the generated set)."""
    require_gpu(extr, intr)
    extr = extr.to(_f32).contiguous()
    intr = intr.to(_f32).contiguous()
    V = extr.shape[0]
    dev = extr.device
    if pixel_ids is not None:
        ids = pixel_ids.to(device=dev, dtype=_i32).contiguous()
        R = ids.numel()
        o = torch.empty((V, R, 3), dtype=_f32, device=dev)
        d = torch.empty((V, R, 3), dtype=_f32, device=dev)
        t0 = torch.empty((V, R), dtype=_f32, device=dev)
        t1 = torch.empty((V, R), dtype=_f32, device=dev)
        ws = torch.empty(lib().npcd_ray_gen_ws_floats(V, res, R), dtype=_f32, device=dev)
        check(lib().npcd_ray_gen_subset(ptr(extr), ptr(intr), V, res, float(box), ptr(ids), R, ptr(o), ptr(d), ptr(t0), ptr(t1), ptr(ws),
                                        stream_ptr()), "npcd_ray_gen_subset")
        return o, d, t0, t1
    R = res * res
    o = torch.empty((V, R, 3), dtype=_f32, device=dev)
    d = torch.empty((V, R, 3), dtype=_f32, device=dev)
    t0 = torch.empty((V, R), dtype=_f32, device=dev)
    t1 = torch.empty((V, R), dtype=_f32, device=dev)
    ws = torch.empty(lib().npcd_ray_gen_ws_floats(V, res, 0), dtype=_f32, device=dev)
    check(lib().npcd_ray_gen(ptr(extr), ptr(intr), V, res, float(box), ptr(o), ptr(d), ptr(t0), ptr(t1), ptr(ws), stream_ptr()),
          "npcd_ray_gen")
    return o, d, t0, t1


# When set to a list, every query_compact() call (counter fill + the neighbour-query kernel, one C call) is bracketed by HIP events
# recorded on the launch stream (bench.py: roofline entry of the renderer's second-largest kernel).
QUERY_EVENTS = None

# The compact lists of shading points are laid out in RAY ORDER by a prefix sum over the per-ray counts (two launches, no atomics):
# a render is bit-identical from run to run.  NPCD_COMPACT_ORDERED=0 selects the one-launch form whose list order is the order in
# which the rays' waves reach an atomic counter (same per-ray content, ~2 % faster, last bits of the image vary between runs
# through the order of the float atomics in the weight gradients only -- the forward is order-independent either way).
COMPACT_ORDERED = os.environ.get("NPCD_COMPACT_ORDERED", "1") != "0"
# Round 6 (OPT-IN, NPCD_RENDER_FUSED_RAYS=1): the fused render can generate its rays inside the query launch and fix the limits of missing
# rays inside the march (npcd_render_rays_query / npcd_ray_march_compact_fused: two launches per view fewer, the same bits --
# tests/test_gpu_render.py::test_rays_generated_inside_the_query_launch_are_the_same_bits).  Measured: 0.353 against 0.351-0.352 ms per view
# at 128 depth samples, 0.210 against 0.207 at 64 -- what the two 5-us launches cost, the per-wave ray arithmetic and the limit pairs cost
# again inside the 4,096-workgroup kernels (docs/experiments.md R6.6).  Not the default.
FUSED_RAYS = os.environ.get("NPCD_RENDER_FUSED_RAYS", "0") == "1"


class HipVoxelGrid:
    """Device-side state of a torch_knnquery.VoxelGrid: parameters + the workspace written by
    set_pointset (per point fine-voxel coordinates / kept flag, per example coarse occupancy)."""

    def __init__(self, voxel_size, voxel_scale, kernel_size, max_points_per_voxel, max_occ_voxels_per_example, ranges, grid_level=None):
        """The six arguments of torch_knnquery.VoxelGrid (pointnerf.py:147-153).  `grid_level` ("fine" / "scaled"; default
        DEFAULT_GRID_LEVEL, env NPCD_GRID_LEVEL) selects which reading of the unavailable upstream source is followed: see
        include/npcd_hip.h (npcd_grid_params.grid_level) and DESIGN.md section 3."""
        self.params = make_grid_params(voxel_size, voxel_scale, kernel_size, max_points_per_voxel,
                                       max_occ_voxels_per_example, ranges, grid_level)
        self.grid_level = "scaled" if self.params.grid_level == 1 else "fine"
        self.vsize_tup = tuple(float(v) for v in voxel_size)
        self.points: Optional[torch.Tensor] = None
        self.workspace: Optional[torch.Tensor] = None
        self._built_from = None          # (tensor sharing the storage the grid was built from, its version, geometry)

    def set_grid_level(self, grid_level: str):
        """Switch between the two readings ("fine" / "scaled") on an existing grid; the next set_pointset rebuilds."""
        if grid_level not in GRID_LEVELS:
            raise ValueError(f"grid_level must be one of {sorted(GRID_LEVELS)}; got {grid_level!r}")
        self.params.grid_level = GRID_LEVELS[grid_level]
        self.grid_level = grid_level
        self._built_from, self.points = None, None

    def set_pointset(self, points: torch.Tensor, counts: Optional[torch.Tensor] = None):
        """Build the grid's device-side state for a batch of clouds.  Rendering many views of the same cloud (the evaluation
        protocol: 251 views per object) calls this once per view with the same, unmodified tensor: the build (16 us, 4 % of a 128^2
        view) is skipped when the storage, its version counter and the view geometry are the ones of the last build and every
        cloud is complete (counts None).  The cache holds a reference to that storage, so its address cannot be reused."""
        require_gpu(points)
        key = (points._version, points.storage_offset(), tuple(points.shape), tuple(points.stride()), points.dtype, points.device)
        if (counts is None and self._built_from is not None and self._built_from[1] == key
                and self._built_from[0].untyped_storage().data_ptr() == points.untyped_storage().data_ptr()):
            return
        self._built_from = None
        pts = points.detach().to(_f32).contiguous()
        B, N, _ = pts.shape
        nbytes = lib().npcd_grid_workspace_bytes(ctypes.byref(self.params), B, N)
        if nbytes < 0:
            raise RuntimeError(f"voxel grid configuration / point count (N={N}) not supported by the HIP kernels")
        if self.workspace is None or self.workspace.numel() < nbytes or self.workspace.device != pts.device:
            self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=pts.device)
        cnt = None if counts is None else counts.to(device=pts.device, dtype=_i32).contiguous()
        check(lib().npcd_grid_build(ctypes.byref(self.params), ptr(pts), ptr(cnt), B, N, ptr(self.workspace), stream_ptr()),
              "npcd_grid_build")
        self.points = pts
        if counts is None:
            self._built_from = (points.detach(), key)

    def query_dense(self, k: int, r: float, M: int, *, x: Optional[torch.Tensor] = None, rays=None, S: Optional[int] = None,
                    mode: int = 0, points: Optional[torch.Tensor] = None):
        """Dense per-ray result: idx [B,R,M,k] int32, loc [B,R,M,3], slot_sample [B,R,M] int32, nsel [B,R] int32.
        Either x [B,R,S,3] or rays=(o [B,R,3], d [B,R,3], t0 [B,R], t1 [B,R]) with S depth samples."""
        pts = self.points if points is None else points.detach().to(_f32).contiguous()
        if pts is None:
            raise RuntimeError("VoxelGrid.query before set_pointset")
        B, N, _ = pts.shape
        if x is not None:
            require_gpu(x)
            x = x.to(_f32).contiguous()
            assert x.shape[0] == B
            R, S = x.shape[1], x.shape[2]
            o = d = t0 = t1 = None
        else:
            o, d, t0, t1 = (t.to(_f32).contiguous() for t in rays)
            R = o.shape[1]
        dev = pts.device
        idx = torch.empty((B, R, M, k), dtype=_i32, device=dev)
        loc = torch.empty((B, R, M, 3), dtype=_f32, device=dev)
        ss = torch.empty((B, R, M), dtype=_i32, device=dev)
        nsel = torch.empty((B, R), dtype=_i32, device=dev)
        check(lib().npcd_grid_query(ctypes.byref(self.params), ptr(self.workspace if mode == 0 else None), ptr(pts),
                                    B, N, R, int(S), int(M), int(k), float(r), int(mode), ptr(x), ptr(o), ptr(d), ptr(t0), ptr(t1),
                                    ptr(idx), ptr(loc), ptr(ss), ptr(nsel), stream_ptr()), "npcd_grid_query")
        return idx, loc, ss, nsel

    def query_compact(self, k: int, r: float, M: int, rays, S: int, capacity: int, points: Optional[torch.Tensor] = None):
        """Fused-render form (npcd_grid_query_compact): returns counter [4] int32 (P, overflow flag, shading status word = 0, 0),
        ray_base [B*R] int32, ray_nsel [B*R] int32, ray_bits [B*R] int64 (valid-slot masks),
        nb [capacity,k] int32, pts [capacity,3] fp32 -- all on the device, no host round trip."""
        pts_t = self.points if points is None else points.detach().to(_f32).contiguous()
        B, N, _ = pts_t.shape
        o, d, t0, t1 = (t.to(_f32).contiguous() for t in rays)
        R = o.shape[1]
        dev = pts_t.device
        counter = torch.empty(4, dtype=_i32, device=dev)
        ray_base = torch.empty(B * R, dtype=_i32, device=dev)
        ray_nsel = torch.empty(B * R, dtype=_i32, device=dev)
        ray_bits = torch.empty(B * R, dtype=torch.int64, device=dev)
        nb = torch.empty((capacity, k), dtype=_i32, device=dev)
        cpts = torch.empty((capacity, 3), dtype=_f32, device=dev)
        order_ws = None
        if COMPACT_ORDERED:
            order_ws = torch.empty(lib().npcd_grid_query_order_ws_bytes(B, R, int(M), int(k)), dtype=torch.uint8, device=dev)
        ev = None
        if QUERY_EVENTS is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if COMPACT_ORDERED:
            check(lib().npcd_grid_query_compact_ordered(
                ctypes.byref(self.params), ptr(self.workspace), ptr(pts_t), B, N, R, int(S), int(M), int(k), float(r), ptr(o), ptr(d),
                ptr(t0), ptr(t1), ptr(counter), int(capacity), ptr(ray_base), ptr(ray_nsel), ptr(ray_bits), ptr(nb), ptr(cpts),
                ptr(order_ws), stream_ptr()), "npcd_grid_query_compact_ordered")
        else:
            check(lib().npcd_grid_query_compact(
                ctypes.byref(self.params), ptr(self.workspace), ptr(pts_t), B, N, R, int(S), int(M), int(k), float(r), ptr(o), ptr(d),
                ptr(t0), ptr(t1), ptr(counter), int(capacity), ptr(ray_base), ptr(ray_nsel), ptr(ray_bits), ptr(nb), ptr(cpts),
                stream_ptr()), "npcd_grid_query_compact")
        if ev is not None:
            ev[1].record()
            QUERY_EVENTS.append(ev)
        return counter, ray_base, ray_nsel, ray_bits, nb, cpts

    def query_compact_rays(self, k: int, r: float, M: int, extr: torch.Tensor, intr: torch.Tensor, res: int, box: float, S: int,
                           capacity: int, points: Optional[torch.Tensor] = None):
        """Fused render form (npcd_render_rays_query, round 6): ray generation inside the query launch.  extr [B, T, 4, 4], intr [B, T, 3, 3]
        -> (rays (o [B, T R, 3], d, t0 [B, T R], t1 -- t0 / t1 RAW for rays that miss the cube), lim_part, counter, ray_base, ray_nsel,
        ray_bits, nb, pts) or None when the configuration is not covered (cube smaller than the grid's range, unordered lists): the
        caller then generates the rays with ray_gen and calls query_compact."""
        if not COMPACT_ORDERED or not FUSED_RAYS:
            return None
        g = self.params
        if any(not (box >= g.range_max[a] and -box <= g.range_min[a]) for a in range(3)):
            return None
        pts_t = self.points if points is None else points.detach().to(_f32).contiguous()
        B, N, _ = pts_t.shape
        T = extr.shape[1]
        R = T * res * res
        dev = pts_t.device
        extr = extr.to(_f32).contiguous()
        intr = intr.to(_f32).contiguous()
        o = torch.empty((B, R, 3), dtype=_f32, device=dev)
        d = torch.empty((B, R, 3), dtype=_f32, device=dev)
        t0 = torch.empty((B, R), dtype=_f32, device=dev)
        t1 = torch.empty((B, R), dtype=_f32, device=dev)
        counter = torch.empty(4, dtype=_i32, device=dev)
        ray_base = torch.empty(B * R, dtype=_i32, device=dev)
        ray_nsel = torch.empty(B * R, dtype=_i32, device=dev)
        ray_bits = torch.empty(B * R, dtype=torch.int64, device=dev)
        nb = torch.empty((capacity, k), dtype=_i32, device=dev)
        cpts = torch.empty((capacity, 3), dtype=_f32, device=dev)
        L = lib()
        order_ws = torch.empty(L.npcd_grid_query_order_ws_bytes(B, R, int(M), int(k)), dtype=torch.uint8, device=dev)
        lim = torch.empty(L.npcd_render_lim_words(B, R), dtype=_i32, device=dev)
        ev = None
        if QUERY_EVENTS is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        check(L.npcd_render_rays_query(ctypes.byref(self.params), ptr(self.workspace), ptr(pts_t), B, N, ptr(extr), ptr(intr), T, int(res),
                                       float(box), int(S), int(M), int(k), float(r), ptr(o), ptr(d), ptr(t0), ptr(t1), ptr(counter),
                                       int(capacity), ptr(ray_base), ptr(ray_nsel), ptr(ray_bits), ptr(nb), ptr(cpts), ptr(order_ws), ptr(lim),
                                       stream_ptr()), "npcd_render_rays_query")
        if ev is not None:
            ev[1].record()
            QUERY_EVENTS.append(ev)
        return (o, d, t0, t1), lim, counter, ray_base, ray_nsel, ray_bits, nb, cpts

    def query(self, x: torch.Tensor, k: int, r: float, max_shading_pts: int):
        """torch_knnquery.VoxelGrid.query contract (aggregator.py:63-73):
        sample_idx [R_valid, M, k], sample_loc [R_valid, M, 3], ray_mask [B, R]."""
        idx, loc, _, nsel = self.query_dense(k, r, max_shading_pts, x=x)
        ray_mask = nsel > 0
        return idx[ray_mask], loc[ray_mask], ray_mask


# ---- fused shading -------------------------------------------------------------------------------
FIELD_ORDER = tuple([f"aggregator.local_field.{i}" for i in (0, 2, 4, 6, 8)] + ["shape_net.0", "shape_net.2"]
                    + [f"channel_net.{i}" for i in (0, 2, 4, 6, 8)])


def pack_field_weights(state: dict, feat_dim: int, device, n_freqs: int = 10, hidden: int = 256) -> torch.Tensor:
    """Pack the 12 Linear layers of a Field (state_dict keys relative to the Field module) into the
    fp16 MFMA-fragment order the shading kernels stream from L2.  Returns a uint8 device tensor."""
    L = lib()
    nbytes = L.npcd_shade_wpack_bytes(feat_dim, n_freqs, hidden)
    if nbytes < 0:
        raise RuntimeError(f"shading kernels support feat_dim in (32, 128), n_freqs=10, hidden=256; got "
                           f"({feat_dim}, {n_freqs}, {hidden})")
    ws = [state[n + ".weight"].detach().to("cpu", _f32).contiguous() for n in FIELD_ORDER]
    bs = [state[n + ".bias"].detach().to("cpu", _f32).contiguous() for n in FIELD_ORDER]
    wp = (ctypes.c_void_p * 12)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * 12)(*[b.data_ptr() for b in bs])
    host = torch.empty(nbytes, dtype=torch.uint8)
    check(L.npcd_shade_pack_weights(wp, bp, feat_dim, n_freqs, hidden, ctypes.c_void_p(host.data_ptr())), "npcd_shade_pack_weights")
    return host.to(device)


def pairs_x2_pack(state: dict, feat_dim: int, device) -> torch.Tensor:
    """The four non-linear per-pair layers of a Field (aggregator.local_field.{0,2,4,6}) as hi / lo bf16 fragments for npcd_pairs_x2."""
    L = lib()
    nbytes = L.npcd_pairs_x2_wpack_bytes(feat_dim)
    if nbytes < 0:
        raise RuntimeError(f"npcd_pairs_x2 supports feat_dim in (32, 128); got {feat_dim}")
    ws = [state[n + ".weight"].detach().to("cpu", _f32).contiguous() for n in FIELD_ORDER]
    bs = [state[n + ".bias"].detach().to("cpu", _f32).contiguous() for n in FIELD_ORDER]
    wp = (ctypes.c_void_p * 12)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * 12)(*[b.data_ptr() for b in bs])
    host = torch.empty(nbytes, dtype=torch.uint8)
    check(L.npcd_pairs_x2_pack(wp, bp, feat_dim, ctypes.c_void_p(host.data_ptr())), "npcd_pairs_x2_pack")
    return host.to(device)


def pairs_x2(wpack: torch.Tensor, feat_dim: int, nb_idx: torch.Tensor, pts: torch.Tensor, kp_pos: torch.Tensor, kp_feat: torch.Tensor):
    """nb_idx [P, k] (global indices, -1 pad anywhere), pts [P, 3], kp_pos [B N, 3], kp_feat [B N, F] -> G [P, 256] fp32: the four
    non-linear per-pair layers and the inverse-distance mean in the reference's fp32 numerics class (csrc/points_x2.hip, forward only)."""
    require_gpu(wpack, nb_idx, pts, kp_pos, kp_feat)
    P, k = nb_idx.shape
    G = torch.empty((P, 256), dtype=_f32, device=pts.device)
    if P == 0:
        return G
    check(lib().npcd_pairs_x2(ptr(wpack), feat_dim, ptr(nb_idx.to(_i32).contiguous()), ptr(pts.to(_f32).contiguous()),
                              ptr(kp_pos.to(_f32).contiguous()), ptr(kp_feat.to(_f32).contiguous()), None, P, k, ptr(G), stream_ptr()),
          "npcd_pairs_x2")
    return G


def points_x2_pack(state: dict, device) -> torch.Tensor:
    """The point-level layers of a Field (local_field.8, shape_net, channel_net; state_dict keys relative to the Field module) as hi / lo
    bf16 fragments for npcd_points_x2 (csrc/points_x2.hip).  Returns a uint8 device tensor."""
    L = lib()
    ws = [state[n + ".weight"].detach().to("cpu", _f32).contiguous() for n in FIELD_ORDER]
    bs = [state[n + ".bias"].detach().to("cpu", _f32).contiguous() for n in FIELD_ORDER]
    wp = (ctypes.c_void_p * 12)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * 12)(*[b.data_ptr() for b in bs])
    host = torch.empty(L.npcd_points_x2_wpack_bytes(), dtype=torch.uint8)
    check(L.npcd_points_x2_pack(wp, bp, int(ws[7].shape[1]), ctypes.c_void_p(host.data_ptr())), "npcd_points_x2_pack")
    return host.to(device)


def points_x2_pack_device(mods, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The same pack from the eight Linear modules on the device (local_field.8, shape_net.{0,2}, channel_net.{0,2,4,6,8}), one
    launch; `out`: a pack to overwrite (training re-packs after every optimizer step)."""
    L = lib()
    ws = [m.weight.detach().to(_f32).contiguous() for m in mods]
    bs = [m.bias.detach().to(_f32).contiguous() for m in mods]
    require_gpu(*ws)
    if out is None:
        out = torch.zeros(L.npcd_points_x2_wpack_bytes(), dtype=torch.uint8, device=ws[0].device)      # (the pad words stay zero)
    wp = (ctypes.c_void_p * 8)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * 8)(*[b.data_ptr() for b in bs])
    check(L.npcd_points_x2_pack_dev(wp, bp, int(ws[3].shape[1]), ptr(out), stream_ptr()), "npcd_points_x2_pack_dev")
    return out


def points_x2(wpack: torch.Tensor, feat: torch.Tensor, dir_bias: Optional[torch.Tensor] = None, point_ray: Optional[torch.Tensor] = None,
              save: bool = False):
    """feat [P, 256] fp32 (aggregated per-point features) -> sigma [P], rgb [P, 3]: the last aggregator layer and both heads in the
    reference's fp32 numerics class on the matrix cores (split bf16 operands).  dir_bias [rows, 256] fp32 + point_ray [P] int32: the
    direction part of the first colour layer's pre-activation (use_view_dir)."""
    require_gpu(wpack, feat)
    if (dir_bias is None) != (point_ray is None):
        raise ValueError("dir_bias and point_ray go together")
    P = feat.shape[0]
    feat = feat.to(_f32).contiguous()
    if save:
        # training forward: (pre [P, 4] = the heads' pre-activations r, g, b, sigma; saved [6, P, 256] = feat, s0, c0, c1, c2, c3)
        if dir_bias is not None:
            raise ValueError("the training forward covers the published configuration (no view directions)")
        pre = torch.empty((P, 4), dtype=_f32, device=feat.device)
        saved = torch.empty((6, P, 256), dtype=_f32, device=feat.device)
        if P:
            check(lib().npcd_points_x2_train(ptr(wpack), ptr(feat), P, ptr(saved), ptr(pre), stream_ptr()), "npcd_points_x2_train")
        return pre, saved
    sigma = torch.empty(P, dtype=_f32, device=feat.device)
    rgb = torch.empty((P, 3), dtype=_f32, device=feat.device)
    if P == 0:
        return sigma, rgb
    if dir_bias is not None:
        dir_bias, point_ray = dir_bias.to(_f32).contiguous(), point_ray.to(_i32).contiguous()
    check(lib().npcd_points_x2(ptr(wpack), ptr(feat), None, P, ptr(sigma), ptr(rgb), ptr(dir_bias), ptr(point_ray), stream_ptr()),
          "npcd_points_x2")
    return sigma, rgb


# When set to a list, the two shading kernels of every shade_points() call (shade_pairs_kernel + shade_points_kernel, one C call)
# are bracketed by HIP events recorded on the launch stream (bench.py: per-kernel roofline of the renderer's dominant kernels).
SHADE_EVENTS = None
SHADE_NONFINITE_PAIRS, SHADE_NONFINITE_HEADS = 1, 2          # bits of the range-guard word (include/npcd_hip.h)


def shade_points(wpack: torch.Tensor, feat_dim: int, nb_idx: torch.Tensor, pts: torch.Tensor, kp_pos: torch.Tensor,
                 kp_feat: torch.Tensor, n_points: Optional[torch.Tensor] = None, n_freqs: int = 10, hidden: int = 256,
                 dir_bias: Optional[torch.Tensor] = None, point_ray: Optional[torch.Tensor] = None,
                 status: Optional[torch.Tensor] = None):
    """nb_idx [P,k] int32 (global indices, -1 pad), pts [P,3], kp_pos [B*N,3], kp_feat [B*N,F] -> sigma [P], rgb [P,3].
    use_view_dir (fields/mlp.py:67-70): dir_bias [n_rays, hidden] fp32 = the direction part of the first colour layer's
    pre-activation per ray, point_ray [P] int32 = the ray of every compact point (npcd_shade_points_dir).
    status: optional device int32 word (zeroed by the caller) that the kernels OR with SHADE_NONFINITE_PAIRS / _HEADS when an
    fp16 activation left its range (include/npcd_hip.h, range guard)."""
    require_gpu(wpack, nb_idx, pts, kp_pos, kp_feat)
    if status is not None:
        require_gpu(status)
        assert status.dtype == _i32 and status.numel() >= 1
    if (dir_bias is None) != (point_ray is None):
        raise ValueError("dir_bias and point_ray go together")
    P, k = nb_idx.shape
    dev = pts.device
    nb_idx = nb_idx.to(_i32).contiguous()
    pts, kp_pos, kp_feat = pts.to(_f32).contiguous(), kp_pos.to(_f32).contiguous(), kp_feat.to(_f32).contiguous()
    sigma = torch.empty(P, dtype=_f32, device=dev)
    rgb = torch.empty((P, 3), dtype=_f32, device=dev)
    if P == 0:
        return sigma, rgb
    if n_points is None:
        n_points = torch.full((1,), P, dtype=_i32, device=dev)
    L = lib()
    wsb = L.npcd_shade_workspace_bytes(P, hidden)
    work = torch.empty(wsb, dtype=torch.uint8, device=dev)
    ev = None
    if SHADE_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if dir_bias is None:
        check(L.npcd_shade_points(ptr(wpack), feat_dim, n_freqs, hidden, ptr(nb_idx), ptr(pts), ptr(kp_pos), ptr(kp_feat),
                                  ptr(n_points), P, k, ptr(sigma), ptr(rgb), ptr(work), ptr(status), stream_ptr()), "npcd_shade_points")
    else:
        require_gpu(dir_bias, point_ray)
        dir_bias = dir_bias.to(_f32).contiguous()
        point_ray = point_ray.to(_i32).contiguous()
        assert dir_bias.shape[1] == hidden and point_ray.shape[0] == P
        check(L.npcd_shade_points_dir(ptr(wpack), feat_dim, n_freqs, hidden, ptr(nb_idx), ptr(pts), ptr(kp_pos), ptr(kp_feat),
                                      ptr(n_points), P, k, ptr(sigma), ptr(rgb), ptr(work), ptr(dir_bias), ptr(point_ray),
                                      ptr(status), stream_ptr()), "npcd_shade_points_dir")
    if ev is not None:
        ev[1].record()
        SHADE_EVENTS.append(ev)
    return sigma, rgb


def ray_march_compact(sigma, rgb, ray_bits, pts, ray_base, rays_o, rays_d, t1, M, white_back=True, fused=None):
    """Ray march on the compact layout of HipVoxelGrid.query_compact.  `fused` = (t0, lim_part) of query_compact_rays: the march then
    ends rays that miss the cube at the global end itself (npcd_ray_march_compact_fused)."""
    Nr = ray_base.shape[0]
    dev = ray_base.device
    mask = torch.empty(Nr, dtype=_f32, device=dev)
    depth = torch.empty(Nr, dtype=_f32, device=dev)
    chan = torch.empty((Nr, 3), dtype=_f32, device=dev)
    ws = torch.empty(lib().npcd_ray_march_ws_floats(Nr), dtype=_f32, device=dev)
    if fused is not None:
        t0, lim = fused
        check(lib().npcd_ray_march_compact_fused(ptr(sigma), ptr(rgb), ptr(ray_bits), ptr(pts), ptr(ray_base), ptr(rays_o.contiguous()),
                                                 ptr(rays_d.contiguous()), ptr(t0.contiguous()), ptr(t1.contiguous()), ptr(lim), lim.numel() // 2,
                                                 Nr, int(M), int(sigma.shape[0]), int(bool(white_back)), ptr(mask), ptr(depth),
                                                 ptr(chan), ptr(ws), stream_ptr()), "npcd_ray_march_compact_fused")
        return mask, depth, chan
    check(lib().npcd_ray_march_compact(ptr(sigma), ptr(rgb), ptr(ray_bits), ptr(pts), ptr(ray_base), ptr(rays_o.contiguous()),
                                       ptr(rays_d.contiguous()), ptr(t1.contiguous()), Nr, int(M), int(sigma.shape[0]), int(bool(white_back)), ptr(mask),
                                       ptr(depth), ptr(chan), ptr(ws), stream_ptr()), "npcd_ray_march_compact")
    return mask, depth, chan


def ray_march(sigma, rgb, slot_valid, slot_loc, point_base, rays_o, rays_d, t1, white_back=True):
    """sigma [P], rgb [P,3] compact (row-major over valid [ray, slot]); slot_valid [Nr,M] bool/uint8,
    slot_loc [Nr,M,3], point_base [Nr] int32 -> mask [Nr], depth [Nr], channels [Nr,3]."""
    require_gpu(sigma, slot_valid, slot_loc)
    Nr, M = slot_valid.shape
    dev = slot_loc.device
    sv = slot_valid.to(torch.uint8).contiguous()
    mask = torch.empty(Nr, dtype=_f32, device=dev)
    depth = torch.empty(Nr, dtype=_f32, device=dev)
    chan = torch.empty((Nr, 3), dtype=_f32, device=dev)
    ws = torch.empty(lib().npcd_ray_march_ws_floats(Nr), dtype=_f32, device=dev)
    if sigma.numel() == 0:                      # keep the pointers valid
        sigma = torch.zeros(1, dtype=_f32, device=dev)
        rgb = torch.zeros((1, 3), dtype=_f32, device=dev)
    check(lib().npcd_ray_march(ptr(sigma.contiguous()), ptr(rgb.contiguous()), ptr(sv), ptr(slot_loc.contiguous()),
                               ptr(point_base.to(_i32).contiguous()), ptr(rays_o.contiguous()), ptr(rays_d.contiguous()),
                               ptr(t1.contiguous()), Nr, M, int(bool(white_back)), ptr(mask), ptr(depth), ptr(chan), ptr(ws),
                               stream_ptr()), "npcd_ray_march")
    return mask, depth, chan


# ---- stage-1 training path: pair inputs and aggregation with hand-written forward AND backward (csrc/pairs.hip) -------------
class _PairInput(torch.autograd.Function):
    """feat table [Nt, F] (differentiable), flat / owner [Q] int64, pts [P, 3], pos table [Nt, 3] -> x0 [Q, F + 3 + 6 nf], w [Q]"""

    @staticmethod
    def forward(ctx, feat, flat, owner, pts, pos, n_freqs):
        require_gpu(feat, flat, owner, pts, pos)
        feat, pts, pos = feat.contiguous(), pts.contiguous(), pos.contiguous()
        Q, F_ = flat.numel(), feat.shape[1]
        ncol = F_ + 3 + 6 * n_freqs
        x0 = torch.empty((Q, ncol), dtype=_f32, device=feat.device)
        w = torch.empty(Q, dtype=_f32, device=feat.device)
        check(lib().npcd_pair_input_fwd(ptr(flat), ptr(owner), ptr(pts), ptr(pos), ptr(feat), F_, n_freqs, Q, ptr(x0), ptr(w), stream_ptr()),
              "npcd_pair_input_fwd")
        ctx.save_for_backward(flat)
        ctx.dims = (feat.shape[0], F_, ncol)
        ctx.mark_non_differentiable(w)
        return x0, w

    @staticmethod
    def backward(ctx, dx0, _dw):
        (flat,) = ctx.saved_tensors
        Nt, F_, ncol = ctx.dims
        dx0 = dx0.contiguous()
        dfeat = torch.zeros((Nt, F_), dtype=_f32, device=dx0.device)
        check(lib().npcd_pair_input_bwd(ptr(flat), ptr(dx0), F_, ncol, flat.numel(), ptr(dfeat), stream_ptr()), "npcd_pair_input_bwd")
        return dfeat, None, None, None, None, None


class _PairAggregate(torch.autograd.Function):
    """local [Q, C] (differentiable), w [Q], off / cnt [P] int64 -> agg [P, C] = inverse-distance weighted mean over a point's pairs"""

    @staticmethod
    def forward(ctx, local, w, off, cnt):
        require_gpu(local, w, off, cnt)
        local = local.contiguous()
        P, C = off.numel(), local.shape[1]
        agg = torch.empty((P, C), dtype=_f32, device=local.device)
        check(lib().npcd_pair_aggregate(0, ptr(local), ptr(w), ptr(off), ptr(cnt), C, P, ptr(agg), stream_ptr()), "npcd_pair_aggregate")
        ctx.save_for_backward(w, off, cnt)
        ctx.shape = tuple(local.shape)
        return agg

    @staticmethod
    def backward(ctx, dagg):
        w, off, cnt = ctx.saved_tensors
        dagg = dagg.contiguous()
        dlocal = torch.empty(ctx.shape, dtype=_f32, device=dagg.device)
        check(lib().npcd_pair_aggregate(1, ptr(dagg), ptr(w), ptr(off), ptr(cnt), ctx.shape[1], off.numel(), ptr(dlocal), stream_ptr()),
              "npcd_pair_aggregate(bwd)")
        return dlocal, None, None, None


class _RayMarch(torch.autograd.Function):
    """ray_march with a hand-written backward w.r.t. the compact densities / colours (everything else is constant in stage 1)"""

    @staticmethod
    def forward(ctx, sigma, rgb, slot_valid, slot_loc, point_base, rays_o, rays_d, t1, white_back):
        require_gpu(sigma, slot_valid, slot_loc)
        Nr, M = slot_valid.shape
        dev = slot_loc.device
        sv = slot_valid.to(torch.uint8).contiguous()
        sigma, rgb, slot_loc = sigma.contiguous(), rgb.contiguous(), slot_loc.contiguous()
        base, o, d, t1 = point_base.to(_i32).contiguous(), rays_o.contiguous(), rays_d.contiguous(), t1.contiguous()
        mask = torch.empty(Nr, dtype=_f32, device=dev)
        depth = torch.empty(Nr, dtype=_f32, device=dev)
        chan = torch.empty((Nr, 3), dtype=_f32, device=dev)
        ws = torch.empty(lib().npcd_ray_march_ws_floats(Nr), dtype=_f32, device=dev)
        check(lib().npcd_ray_march(ptr(sigma), ptr(rgb), ptr(sv), ptr(slot_loc), ptr(base), ptr(o), ptr(d), ptr(t1), Nr, M,
                                   int(bool(white_back)), ptr(mask), ptr(depth), ptr(chan), ptr(ws), stream_ptr()), "npcd_ray_march")
        ctx.save_for_backward(sigma, rgb, sv, slot_loc, base, o, d, t1, ws)
        ctx.white_back = int(bool(white_back))
        return mask, depth, chan

    @staticmethod
    def backward(ctx, g_mask, g_depth, g_chan):
        sigma, rgb, sv, slot_loc, base, o, d, t1, ws = ctx.saved_tensors
        Nr, M = sv.shape
        dsigma = torch.empty_like(sigma)
        drgb = torch.empty_like(rgb)
        check(lib().npcd_ray_march_bwd(ptr(sigma), ptr(rgb), ptr(sv), ptr(slot_loc), ptr(base), ptr(o), ptr(d), ptr(t1), Nr, M, ctx.white_back,
                                       ptr(ws), ptr(g_mask.contiguous()), ptr(g_depth.contiguous()), ptr(g_chan.contiguous()), ptr(dsigma),
                                       ptr(drgb), stream_ptr()), "npcd_ray_march_bwd")
        return dsigma, drgb, None, None, None, None, None, None, None


def ray_march_train(sigma, rgb, slot_valid, slot_loc, point_base, rays_o, rays_d, t1, white_back):
    """differentiable (sigma, rgb) form of ray_march: HIP forward and backward"""
    return _RayMarch.apply(sigma, rgb, slot_valid, slot_loc, point_base, rays_o, rays_d, t1, white_back)


def leaky_bwd_colsum(dz, z, slope):
    """dy = dz * (z > 0 ? 1 : slope), dbias [N] (fp32) = column sum of dy: one pass + a fixed-order finalize (csrc/pairs.hip).
    dz, z [R, N] contiguous, fp32 or bf16.  Returns None when the shape is not covered (the caller falls back to torch)."""
    from . import dtype_code
    R, N = z.shape
    esz = z.element_size()
    ct = N * esz // 16
    if z.dtype not in (_f32, torch.bfloat16) or (N * esz) % 16 or ct < 1 or ct > 256 or 256 % ct:
        return None
    L = lib()
    dz, z = dz.contiguous(), z.contiguous()
    nblk = L.npcd_leaky_bwd_blocks(R)
    dy = torch.empty_like(z)
    part = torch.empty((nblk + L.npcd_colsum_scratch_rows(), N), dtype=_f32, device=z.device)
    db = torch.empty(N, dtype=_f32, device=z.device)
    check(L.npcd_leaky_bwd_colsum(ptr(dz), ptr(z), ptr(dy), ptr(part), R, N, float(slope), dtype_code(z), stream_ptr()), "npcd_leaky_bwd_colsum")
    check(L.npcd_colsum_finalize(ptr(part), nblk, N, ptr(db), 0, stream_ptr()), "npcd_colsum_finalize")
    return dy, db


PAIR_MLP_BF16, PAIR_MLP_X2 = 0, 1          # `precision` of the npcd_pair_mlp_* entry points (include/npcd_hip.h)


class _PairMLP(torch.autograd.Function):
    """The four non-linear layers of the per-pair aggregator network + the inverse-distance mean over each point's pairs, forward
    and backward on the matrix cores (csrc/pairs_mlp.hip).  precision PAIR_MLP_BF16: bf16 operands, fp32 accumulation (narrower than
    the reference); PAIR_MLP_X2: fp32-class -- every operand as two bf16 halves, three matrix instructions per product, ~1e-5
    relative; fp32 weight / bias gradients either way.
    feat [Nt, F] (differentiable), w0, b0 .. w3, b3 (differentiable), nb [P, k] int64 (-1 pad), pts [P, 3], pos [Nt, 3],
    off / owner / flat: the compact pair lists of csrc/pairs.hip  ->  G [P, 256] fp32."""

    @staticmethod
    def forward(ctx, feat, w0, b0, w1, b1, w2, b2, w3, b3, nb, pts, pos, off, owner, flat, precision):
        G, wpack, x0, acts, wn = pair_mlp_forward_raw(feat, (w0, w1, w2, w3), (b0, b1, b2, b3), nb, pts, pos, off, flat.numel(), precision)
        ctx.save_for_backward(wpack, x0, acts, wn, owner.contiguous(), flat.contiguous())
        ctx.dims = (feat.shape[0], feat.shape[1], tuple(tuple(w.shape) for w in (w0, w1, w2, w3)), precision)
        return G

    @staticmethod
    def backward(ctx, dG):
        wpack, x0, acts, wn, owner, flat = ctx.saved_tensors
        Nt, F_, wshapes, precision = ctx.dims
        dfeat_tab, dW, db = pair_mlp_backward_raw(dG, wpack, x0, acts, wn, owner, flat, Nt, F_, wshapes, precision)
        return (dfeat_tab, dW[0], db[0], dW[1], db[1], dW[2], db[2], dW[3], db[3], None, None, None, None, None, None, None)


def pair_mlp_pack(weights, biases, F_, precision, dev):
    """npcd_pair_mlp_pack: the four (weight, bias) pairs -> the packed fragment buffer of the given precision."""
    L = lib()
    ws = [t.detach().to(_f32).contiguous() for t in weights]
    bs = [t.detach().to(_f32).contiguous() for t in biases]
    nbytes = L.npcd_pair_mlp_wpack_bytes(F_, precision)
    if nbytes < 0:
        raise RuntimeError(f"the fused pair MLP supports feat_dim in (32, 128) and precision 0 / 1; got {F_} / {precision}")
    wpack = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    wp = (ctypes.c_void_p * 4)(*[w.data_ptr() for w in ws])
    bp = (ctypes.c_void_p * 4)(*[b.data_ptr() for b in bs])
    check(L.npcd_pair_mlp_pack(wp, bp, F_, precision, ptr(wpack), stream_ptr()), "npcd_pair_mlp_pack")
    return wpack


def pair_mlp_forward_raw(feat, weights, biases, nb, pts, pos, off, Q, precision=PAIR_MLP_BF16, save=True, wpack=None):
    """npcd_pair_mlp_pack + npcd_pair_mlp_fwd -> (G [P,256] fp32, wpack, x0 [(2,) Q,F+64] bf16, acts [4,(2,) Q,256] bf16, wn [Q] fp32);
    the 16-bit arrays hold two planes (hi, lo) at PAIR_MLP_X2.  save=False (rendering): nothing is kept for a backward."""
    require_gpu(feat, nb, pts, pos, off)
    L = lib()
    F_ = feat.shape[1]
    P, k = nb.shape
    dev = feat.device
    if wpack is None:
        wpack = pair_mlp_pack(weights, biases, F_, precision, dev)
    feat_c, pts_c, pos_c = feat.detach().to(_f32).contiguous(), pts.to(_f32).contiguous(), pos.to(_f32).contiguous()
    nb_c, off_c = nb.contiguous(), off.contiguous()
    planes = (2,) if precision == PAIR_MLP_X2 else ()
    x0 = acts = wn = None
    if save:
        x0 = torch.empty(planes + (max(Q, 1), F_ + 64), dtype=torch.bfloat16, device=dev)
        acts = torch.empty((4,) + planes + (max(Q, 1), 256), dtype=torch.bfloat16, device=dev)
        wn = torch.empty(max(Q, 1), dtype=_f32, device=dev)
    G = torch.zeros((P, 256), dtype=_f32, device=dev)
    check(L.npcd_pair_mlp_fwd(ptr(wpack), F_, precision, ptr(nb_c), ptr(pts_c), ptr(pos_c), ptr(feat_c), ptr(off_c), P, k, Q, ptr(x0), ptr(acts),
                              ptr(wn), ptr(G), stream_ptr()), "npcd_pair_mlp_fwd")
    return G, wpack, x0, acts, wn


def pair_mlp_backward_raw(dG, wpack, x0, acts, wn, owner, flat, Nt, F_, wshapes, precision=PAIR_MLP_BF16):
    """npcd_pair_mlp_bwd + the feature scatter -> (dfeat [Nt,F], [dW_l], [db_l]) fp32."""
    dev = dG.device
    L = lib()
    Q = flat.numel()
    dW = [torch.empty(s, dtype=_f32, device=dev) for s in wshapes]
    db = [torch.empty(256, dtype=_f32, device=dev) for _ in range(4)]
    dfeat_tab = torch.zeros((Nt, F_), dtype=_f32, device=dev)
    if Q == 0:
        for t in dW + db:
            t.zero_()
        return dfeat_tab, dW, db
    dact = torch.empty((2, 2 if precision == PAIR_MLP_X2 else 1, Q, 256), dtype=torch.bfloat16, device=dev)
    dfeat = torch.empty((Q, F_), dtype=_f32, device=dev)
    part = torch.empty(L.npcd_pair_mlp_bwd_workspace_floats(F_, Q, precision), dtype=_f32, device=dev)
    dWp = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in dW])
    dbp = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in db])
    dG = dG.contiguous().to(_f32)
    check(L.npcd_pair_mlp_bwd(ptr(wpack), F_, precision, ptr(dG), ptr(owner), ptr(wn), ptr(x0), ptr(acts), Q, ptr(dact),
                              ptr(dfeat), ptr(part), dWp, dbp, stream_ptr()), "npcd_pair_mlp_bwd")
    check(L.npcd_pair_input_bwd(ptr(flat), ptr(dfeat), F_, F_, Q, ptr(dfeat_tab), stream_ptr()), "npcd_pair_input_bwd")
    return dfeat_tab, dW, db


def pair_mlp(feat, layers, nb, pts, pos, off, owner, flat, precision=PAIR_MLP_BF16):
    """layers: the four (weight, bias) pairs of aggregator.local_field.{0,2,4,6}."""
    (w0, b0), (w1, b1), (w2, b2), (w3, b3) = layers
    return _PairMLP.apply(feat, w0, b0, w1, b1, w2, b2, w3, b3, nb, pts, pos, off, owner, flat, precision)


def pair_input(feat, flat, owner, pts, pos, n_freqs):
    return _PairInput.apply(feat, flat, owner, pts, pos, n_freqs)


def pair_aggregate(local, w, off, cnt):
    return _PairAggregate.apply(local, w, off, cnt)

