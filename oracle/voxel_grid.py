"""Oracle (numpy, CPU) for the `torch_knnquery.VoxelGrid` boundary.  TEST INFRASTRUCTURE.

PARITY UNPINNED: torch_knnquery is a third-party CUDA extension (README.md:26, unpinned git
HEAD) whose source is not available; the reference holds no test or golden vector for it.
This file implements the deterministic specification of DESIGN.md ("VoxelGrid spec"), which
is derived from the reference's call sites:

  * constructor kwargs            npcd/models/pointnerf/pointnerf.py:147-153
  * set_pointset(points, counts)  pointnerf.py:67-75, 116-124
  * query(x, k, r, max_shading)   npcd/models/pointnerf/fields/aggregators/aggregator.py:59-73
  * vsize_tup                     aggregator.py:20  (radius = r * max(vsize_tup))

All floating-point steps are single fp32 operations in a fixed order (no FMA contraction), so
that the HIP kernels can reproduce the integer outputs bit-exactly.
"""
from typing import Tuple

import numpy as np

f32 = np.float32


class VoxelGridOracle:
    def __init__(self, voxel_size=(0.04, 0.04, 0.04), voxel_scale=(2, 2, 2), kernel_size=(3, 3, 3),
                 max_points_per_voxel=4, max_occ_voxels_per_example=5000,
                 ranges=(-1.0, -1.0, -1.0, 1.0, 1.0, 1.0), grid_level="scaled"):
        """grid_level "fine": point lists / cap / candidate window on the voxel_size grid, occupancy on the grid coarsened by
        voxel_scale.  grid_level "scaled": ONE grid of edge fp32(voxel_size) * voxel_scale (one fp32 product) with
        ceil(dims / voxel_scale) cells per axis carries lists, caps, occupancy and window -- the fine-grid algorithm below run
        with that edge and voxel_scale 1.  The query radius is r * max(voxel_size), the UNSCALED edge (aggregator.py:20), in both."""
        assert grid_level in ("fine", "scaled")
        self.grid_level = grid_level
        self.vsize_tup = tuple(float(v) for v in voxel_size)          # what aggregator.py:20 reads: never scaled
        self.vsize = np.asarray(voxel_size, dtype=f32)
        self.scale = np.asarray(voxel_scale, dtype=np.int64)
        self.kernel = np.asarray(kernel_size, dtype=np.int64)
        self.half = (self.kernel - 1) // 2
        self.max_ppv = int(max_points_per_voxel)
        self.max_occ = int(max_occ_voxels_per_example)
        self.rmin = np.asarray(ranges[:3], dtype=f32)
        self.rmax = np.asarray(ranges[3:], dtype=f32)
        # grid dimensions are computed on the host in float64
        self.dims = np.array([int(round((float(ranges[3 + a]) - float(ranges[a])) / float(voxel_size[a])))
                              for a in range(3)], dtype=np.int64)
        self.cdims = (self.dims + self.scale - 1) // self.scale
        if grid_level == "scaled":
            self.vsize = (self.vsize * self.scale.astype(f32)).astype(f32)
            self.dims = self.cdims.copy()
            self.scale = np.ones(3, dtype=np.int64)

    # ---- fine voxel coordinates ------------------------------------------------------------
    def fine_coords(self, p: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """p [...,3] fp32 -> (int coords [...,3], in-range flag [...])."""
        p = np.asarray(p, dtype=f32)
        with np.errstate(invalid="ignore", over="ignore"):
            q = np.floor((p - self.rmin) / self.vsize)       # two fp32 roundings: sub, div
        ok = np.all((q >= 0) & (q < self.dims.astype(f32)), axis=-1)    # NaN -> False
        c = np.where(ok[..., None], q, 0).astype(np.int64)
        return c, ok

    # ---- set_pointset ----------------------------------------------------------------------
    def set_pointset(self, points: np.ndarray, counts: np.ndarray):
        points = np.asarray(points, dtype=f32)
        B, N, _ = points.shape
        self.points, self.B, self.N = points, B, N
        self.pcoord = np.zeros((B, N, 3), dtype=np.int64)
        self.kept = np.zeros((B, N), dtype=bool)
        self.occ = np.zeros((B,) + tuple(self.cdims), dtype=bool)
        for b in range(B):
            n = int(counts[b])
            c, ok = self.fine_coords(points[b, :n])
            lin = (c[:, 0] * self.dims[1] + c[:, 1]) * self.dims[2] + c[:, 2]
            # (1) at most max_ppv points per fine voxel, in ascending point index
            seen = {}
            keep = np.zeros(n, dtype=bool)
            for i in range(n):
                if not ok[i]:
                    continue
                r = seen.get(int(lin[i]), 0)
                seen[int(lin[i])] = r + 1
                keep[i] = r < self.max_ppv
            # (2) at most max_occ occupied fine voxels per example, in ascending linear voxel id
            occupied = np.array(sorted(seen.keys()), dtype=np.int64)
            if occupied.size > self.max_occ:
                allowed = set(occupied[: self.max_occ].tolist())
                keep &= np.array([int(v) in allowed for v in lin], dtype=bool)
            self.pcoord[b, :n] = c
            self.kept[b, :n] = keep
            # (3) coarse occupancy, dilated by the kernel
            cc = c[keep] // self.scale
            for off in np.ndindex(*self.kernel):
                o = cc + (np.asarray(off) - self.half)
                inb = np.all((o >= 0) & (o < self.cdims), axis=1)
                o = o[inb]
                self.occ[b, o[:, 0], o[:, 1], o[:, 2]] = True

    # ---- query -----------------------------------------------------------------------------
    def query_dense(self, x: np.ndarray, k: int, r: float, max_shading_pts: int):
        """x [B,R,S,3] -> dense per-ray results (before the ray compaction of `query`):
        idx [B,R,M,k] int32 (global index b*N+i, -1 pad), loc [B,R,M,3] f32,
        nsel [B,R] (number of selected slots), sel_sample [B,R,M] (depth-sample index or -1)."""
        x = np.asarray(x, dtype=f32)
        B, R, S, _ = x.shape
        M = int(max_shading_pts)
        radius = f32(float(r) * max(self.vsize_tup))
        r2 = f32(radius * radius)
        idx = np.full((B, R, M, k), -1, dtype=np.int32)
        loc = np.zeros((B, R, M, 3), dtype=f32)
        nsel = np.zeros((B, R), dtype=np.int32)
        sel_sample = np.full((B, R, M), -1, dtype=np.int32)
        for b in range(B):
            c, ok = self.fine_coords(x[b])                        # [R,S,3], [R,S]
            cc = c // self.scale
            occ = ok & self.occ[b][cc[..., 0], cc[..., 1], cc[..., 2]]
            rank = np.cumsum(occ, axis=1) - 1                     # slot of each selected sample
            take = occ & (rank < M)
            rr, ss = np.nonzero(take)
            slot = rank[rr, ss]
            nsel[b] = np.minimum(occ.sum(axis=1), M)
            sel_sample[b, rr, slot] = ss
            loc[b, rr, slot] = x[b, rr, ss]
            if rr.size == 0:
                continue
            kept = np.nonzero(self.kept[b])[0]
            pk = self.points[b, kept]                             # [Nk,3]
            ck = self.pcoord[b, kept]                             # [Nk,3]
            CH = 8192
            for s0 in range(0, rr.size, CH):
                sl = slice(s0, s0 + CH)
                xs = x[b, rr[sl], ss[sl]]                         # [n,3]
                cs = c[rr[sl], ss[sl]]                            # [n,3]
                adj = np.all(np.abs(cs[:, None, :] - ck[None, :, :]) <= self.half, axis=-1)
                d = xs[:, None, :] - pk[None, :, :]               # fp32
                d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
                cand = adj & (d2 < r2)
                key = np.where(cand, d2, f32(np.inf))
                order = np.argsort(key, axis=1, kind="stable")[:, :k]   # ties -> lower point index
                kd = np.take_along_axis(key, order, axis=1)
                gi = (kept[order] + b * self.N).astype(np.int32)
                gi[~np.isfinite(kd)] = -1
                if gi.shape[1] < k:
                    gi = np.concatenate((gi, np.full((gi.shape[0], k - gi.shape[1]), -1, np.int32)), axis=1)
                idx[b, rr[sl], slot[sl]] = gi
        return idx, loc, nsel, sel_sample

    def query(self, x: np.ndarray, k: int, r: float, max_shading_pts: int):
        """The `VoxelGrid.query` contract: (sample_idx [Rv,M,k], sample_loc [Rv,M,3], ray_mask [B,R])."""
        idx, loc, nsel, _ = self.query_dense(x, k, r, max_shading_pts)
        ray_mask = nsel > 0
        return idx[ray_mask], loc[ray_mask], ray_mask


def brute_force_query(x: np.ndarray, pts: np.ndarray, k: int, radius: float, max_shading_pts: int):
    """The in-repo fallback (aggregator.py:42-58) with exact fp32 distances:
    a sample is valid iff its nearest point is closer than `radius`; the first M valid samples of
    each ray fill slots 0..; neighbours = the k nearest points with dist < radius.
    Returns idx [B,R,M,k] int32 sorted by (dist^2, index), loc [B,R,M,3], nvalid [B,R]."""
    x = np.asarray(x, dtype=f32)
    pts = np.asarray(pts, dtype=f32)
    B, R, S, _ = x.shape
    N = pts.shape[1]
    M = int(max_shading_pts)
    r2 = f32(f32(radius) * f32(radius))
    idx = np.full((B, R, M, k), -1, dtype=np.int32)
    loc = np.zeros((B, R, M, 3), dtype=f32)
    nvalid = np.zeros((B, R), dtype=np.int32)
    CH = max(1, 4096 // S)
    for b in range(B):
        for r0 in range(0, R, CH):
            xs = x[b, r0:r0 + CH].reshape(-1, 3)
            d = xs[:, None, :] - pts[b][None, :, :]
            d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
            key = np.where(d2 < r2, d2, f32(np.inf))
            order = np.argsort(key, axis=1, kind="stable")[:, :k]
            kd = np.take_along_axis(key, order, axis=1)
            gi = (order + b * N).astype(np.int32)
            gi[~np.isfinite(kd)] = -1
            if gi.shape[1] < k:
                gi = np.concatenate((gi, np.full((gi.shape[0], k - gi.shape[1]), -1, np.int32)), axis=1)
            nr = xs.shape[0] // S
            gi = gi.reshape(nr, S, k)
            valid = gi[..., 0] >= 0
            rank = np.cumsum(valid, axis=1) - 1
            take = valid & (rank < M)
            rr, ss = np.nonzero(take)
            slot = rank[rr, ss]
            idx[b, r0 + rr, slot] = gi[rr, ss]
            loc[b, r0 + rr, slot] = x[b, r0 + rr, ss]
            nvalid[b, r0:r0 + nr] = np.minimum(valid.sum(axis=1), M)
    return idx, loc, nvalid
