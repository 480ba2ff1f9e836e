"""GPU parity of the render path (through the C ABI) against the CPU oracle and the golden vectors.

Bars:
  * neighbour indices, slot sample ids, counts, slot positions: BIT-EXACT vs oracle/voxel_grid.py
  * rays / limits: <= 2e-6 abs (fp32, different summation order in the 3x3 transform)
  * shading (fp16 MFMA inputs, fp32 accumulation): sigma <= 2e-3 * max(1, sigma), rgb <= 2e-3 abs
  * rendered pixels: max-abs <= 5e-3 and PSNR(hip, oracle) >= 50 dB  (north-star: PSNR within 0.1 dB)
"""
import math

import numpy as np
import pytest
import torch

from oracle import renderer as orr
from oracle import voxel_grid as ovg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _scene(res, n_views=2, N=512, F_=32, seed=0, B=1):
    coords, feats = orr.synthetic_cloud(N, F_, B, seed=seed)
    poses = [orr.look_at_pose(30 + 97 * i, 20 - 13 * i) for i in range(n_views)]
    extr = torch.stack(poses)[None].expand(B, -1, -1, -1).contiguous()
    K = orr.srn_intrinsics().clone()
    K[0, 0] = K[1, 1] = 131.25 * res / 128
    K[0, 2] = K[1, 2] = res / 2
    intr = K[None, None].expand(B, n_views, 3, 3).contiguous()
    return coords, feats, extr, intr


def _model(F_, N, params):
    from npcd.models.pointnerf import PointNeRF
    m = PointNeRF(1, F_, N, False)
    m.field.load_state_dict(params)
    return m.cuda().eval()


def test_ray_gen_matches_golden_and_oracle(golden):
    from npcd.hip import render as hr
    g = golden("rays")
    for intr_key, okey, dkey in (("intr", "o8", "d8"), ("intr_skew", "o8_skew", "d8_skew")):
        o, d, t0, t1 = hr.ray_gen(T(g["extr"]).cuda(), T(g[intr_key]).cuda(), 8)
        np.testing.assert_allclose(o.cpu().numpy(), g[okey], atol=2e-6)
        np.testing.assert_allclose(d.cpu().numpy(), g[dkey], atol=2e-6)
    o, d, t0, t1 = hr.ray_gen(T(g["extr"]).cuda(), T(g["intr"]).cuda(), 8)
    np.testing.assert_allclose(t0.cpu().numpy(), g["lim_start"][0, ..., 0], atol=2e-6)
    np.testing.assert_allclose(t1.cpu().numpy(), g["lim_end"][0, ..., 0], atol=2e-6)
    # partial miss: rays that miss the cube receive the global min start / max end (renderer.py:40-43)
    o, d, t0, t1 = hr.ray_gen(T(g["extr"]).cuda(), T(g["intr_wide"]).cuda(), 8)
    np.testing.assert_allclose(t0.cpu().numpy(), g["limw_start"][0, ..., 0], atol=2e-6)
    np.testing.assert_allclose(t1.cpu().numpy(), g["limw_end"][0, ..., 0], atol=2e-6)
    st = int(g["stride128"])
    o, d, _, _ = hr.ray_gen(T(g["extr"]).cuda(), T(g["intr"]).cuda(), 128)
    np.testing.assert_allclose(d.cpu().numpy()[:, ::st], g["d128"], atol=2e-6)


def test_ray_gen_all_miss_keeps_sentinels():
    from npcd.hip import render as hr
    extr = orr.look_at_pose(10, 5)[None].clone()
    extr[0, :3, 3] += torch.tensor([0.0, 60.0, 0.0])            # camera far off-axis: every ray misses
    K = orr.srn_intrinsics()[None]
    o, d, t0, t1 = hr.ray_gen(extr.cuda(), K.cuda(), 8)
    ro, rd = orr.camera_rays(extr, K, 8)
    s, e = orr.ray_box_limits(ro, rd)
    assert float(s.max()) == -1.0 and float(e.max()) == -2.0
    assert (t0.cpu() == -1).all() and (t1.cpu() == -2).all()


@pytest.mark.parametrize("level", ["fine", "scaled"])
@pytest.mark.parametrize("N,seed", [(512, 0), (64, 3), (2048, 5)])
def test_grid_query_bit_exact(N, seed, level):
    from npcd.hip import render as hr
    res, S, M, k = 24, 128, 50, 8
    coords, _, extr, intr = _scene(res, 2, N, 32, seed)
    o, d = orr.camera_rays(extr[0], intr[0], res)
    s, e = orr.ray_box_limits(o, d)
    V, R = o.shape[:2]
    ro, rd, rs, re = o.reshape(1, V * R, 3), d.reshape(1, V * R, 3), s.reshape(1, V * R), e.reshape(1, V * R)
    x = (ro[:, :, None] + orr.depth_samples(rs[..., None], re[..., None], S)[..., None] * rd[:, :, None]).numpy()
    cfg = dict(orr.DEFAULT_GRID, grid_level=level)
    g = ovg.VoxelGridOracle(**cfg)
    g.set_pointset(coords.numpy(), np.array([N], dtype=np.int32))
    ridx, rloc, rnsel, rss = g.query_dense(x, k, 2.0, M)
    hg = hr.HipVoxelGrid(**cfg)
    hg.set_pointset(coords.cuda(), torch.full((1,), N, dtype=torch.int32, device="cuda"))
    for kw in (dict(rays=(ro.cuda(), rd.cuda(), rs.cuda(), re.cuda()), S=S), dict(x=T(x).cuda())):
        idx, loc, ss, nsel = hg.query_dense(k, 2.0, M, **kw)
        np.testing.assert_array_equal(nsel.cpu().numpy(), rnsel)
        np.testing.assert_array_equal(ss.cpu().numpy(), rss)
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
        np.testing.assert_array_equal(loc.cpu().numpy(), rloc)
    assert (ridx[..., 0] >= 0).sum() > 1000
    # the torch_knnquery.VoxelGrid.query contract (compaction over rays)
    from torch_knnquery import VoxelGrid
    vg = VoxelGrid(**cfg)
    assert vg.vsize_tup == (0.04, 0.04, 0.04) and vg.grid_level == level
    vg.set_pointset(coords.cuda(), torch.full((1,), N, dtype=torch.int32, device="cuda"))
    sidx, sloc, ray_mask = vg.query(T(x).cuda(), k, 2.0, M)
    oidx, oloc, omask = g.query(x, k, 2.0, M)
    np.testing.assert_array_equal(ray_mask.cpu().numpy(), omask)
    np.testing.assert_array_equal(sidx.cpu().numpy(), oidx)
    np.testing.assert_array_equal(sloc.cpu().numpy(), oloc)
    # brute-force branch
    bidx, bloc, bn = ovg.brute_force_query(x, coords.numpy(), k, 0.08, M)
    idx, loc, ss, nsel = hg.query_dense(k, 0.08, M, x=T(x).cuda(), mode=1)
    np.testing.assert_array_equal(idx.cpu().numpy(), bidx)
    np.testing.assert_array_equal(nsel.cpu().numpy(), bn)
    np.testing.assert_array_equal(loc.cpu().numpy(), bloc)


@pytest.mark.parametrize("level", ["fine", "scaled"])
def test_grid_query_dense_cluster_takes_the_many_candidate_path(level):
    """Samples with MORE than 32 in-radius candidates in their window (the packed rank selection of the query kernel holds one
    candidate per lane of a 32-lane half; past that it falls back to the broadcast loop).  Two ingredients in one cloud: 3,000
    points in a 0.3 cube (every 0.04 cell filled to its cap: the fine reading sees ~80 candidates per sample) and a constructed
    cluster of 4 points in each of the 27 cells of the 0.08 grid around one cell centre, all within the radius of that centre (the
    scaled reading sees all 108).  Dense tables (explicit positions and rays), compact lists and the oracle agree bit for bit."""
    from npcd.hip import render as hr
    gen = torch.Generator().manual_seed(11)
    M, k, S = 24, 8, 48
    cube = (torch.rand(3000, 3, generator=gen) - 0.5) * 0.3
    c0 = torch.tensor([0.56, 0.56, 0.56])                          # centre of cell 19 of the 0.08 grid on every axis
    offs = torch.tensor([[dx, dy, dz] for dx in (-1, 0, 1) for dy in (-1, 0, 1) for dz in (-1, 0, 1)], dtype=torch.float32)
    cluster = (c0 + 0.043 * offs)[:, None, :] + (torch.rand(27, 4, 3, generator=gen) - 0.5) * 0.004
    coords = torch.cat((cube, cluster.reshape(-1, 3)))[None]
    N = coords.shape[1]
    assert int(((cluster.reshape(-1, 3) - c0).norm(dim=-1) < 0.0799).sum()) == 108
    R = 96
    ro = torch.cat(((torch.rand(1, R, 2, generator=gen) - 0.5) * 0.3, torch.full((1, R, 1), -0.6)), dim=-1)
    rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0]) + 0.05 * torch.randn(1, R, 3, generator=gen), dim=-1)
    rs, re = torch.full((1, R), 0.3), torch.full((1, R), 0.9)
    x = ro[:, :, None] + orr.depth_samples(rs[..., None], re[..., None], S)[..., None] * rd[:, :, None]
    x[0, 0, :3] = c0 + torch.tensor([[0.0, 0.0, 0.0], [0.001, -0.002, 0.0015], [-0.003, 0.001, 0.002]])      # samples at the cluster
    x = x.numpy()
    cfg = dict(orr.DEFAULT_GRID, grid_level=level)
    g = ovg.VoxelGridOracle(**cfg)
    g.set_pointset(coords.numpy(), np.array([N], dtype=np.int32))
    ridx, rloc, rnsel, rss = g.query_dense(x, k, 2.0, M)
    kept = coords[0][torch.from_numpy(g.kept[0][:N].astype(bool))]
    slots = torch.from_numpy(rloc[0][ridx[0][..., 0] >= 0])                    # positions of all slots that have a neighbour
    probe = c0 if level == "scaled" else slots[slots.norm(dim=-1).argmin()]      # fine: the slot deepest inside the cube
    assert int(((kept - probe).norm(dim=-1) < 0.08).sum()) > 32     # the premise
    hg = hr.HipVoxelGrid(**cfg)
    hg.set_pointset(coords.cuda(), torch.full((1,), N, dtype=torch.int32, device="cuda"))
    idx, loc, ss, nsel = hg.query_dense(k, 2.0, M, x=T(x).cuda())
    np.testing.assert_array_equal(nsel.cpu().numpy(), rnsel)
    np.testing.assert_array_equal(ss.cpu().numpy(), rss)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(loc.cpu().numpy(), rloc)
    assert (ridx[0, 0, 0] >= 0).all()                               # the sample at the cluster centre has k neighbours
    # the ray forms (dense table and compact lists) on the same cloud
    rays = (ro.cuda(), rd.cuda(), rs.cuda(), re.cuda())
    xr = (ro[:, :, None] + orr.depth_samples(rs[..., None], re[..., None], S)[..., None] * rd[:, :, None]).numpy()
    qidx, qloc, qnsel, _ = g.query_dense(xr, k, 2.0, M)
    idx, loc, ss, nsel = hg.query_dense(k, 2.0, M, rays=rays, S=S)
    np.testing.assert_array_equal(idx.cpu().numpy(), qidx)
    np.testing.assert_array_equal(loc.cpu().numpy(), qloc)
    counter, base, _, bits, nb, pts = hg.query_compact(k, 2.0, M, rays, S, R * M)
    valid = (idx[..., 0] >= 0).flatten(0, 1)
    P = int(valid.sum())
    assert int(counter[0]) == P and P > 500
    assert torch.equal(nb[:P], idx.flatten(0, 1)[valid]) and torch.equal(pts[:P], loc.flatten(0, 1)[valid])


def test_grid_capacity_limits_and_batches():
    """> max_points_per_voxel points in one voxel, occupied-voxel cap, ragged counts, out-of-range points."""
    from npcd.hip import render as hr
    rng = np.random.default_rng(4)
    B, N = 3, 96
    pts = rng.normal(0, 0.25, size=(B, N, 3)).astype(np.float32)
    pts[0, :12] = np.array([0.01, 0.01, 0.01], np.float32) + rng.uniform(0, 0.02, size=(12, 3)).astype(np.float32)
    pts[1, 5] = [1.5, 0, 0]
    pts[1, 6] = [np.nan, 0, 0]
    pts[2, 7] = [1.0, 1.0, 1.0]                     # exactly on the upper bound -> outside
    counts = np.array([N, N - 10, N], dtype=np.int32)
    x = (pts[:, rng.integers(0, N, size=40)][:, :, None, :] + rng.normal(0, 0.03, size=(B, 40, 24, 3))).astype(np.float32)
    for cap, level in ((5000, "fine"), (20, "fine"), (5000, "scaled"), (6, "scaled")):
        cfg = dict(orr.DEFAULT_GRID, max_occ_voxels_per_example=cap, grid_level=level)
        g = ovg.VoxelGridOracle(**cfg)
        g.set_pointset(pts, counts)
        ridx, rloc, rnsel, rss = g.query_dense(x, 8, 2.0, 10)
        hg = hr.HipVoxelGrid(**cfg)
        hg.set_pointset(T(pts).cuda(), T(counts).cuda())
        idx, loc, ss, nsel = hg.query_dense(8, 2.0, 10, x=T(x).cuda())
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
        np.testing.assert_array_equal(nsel.cpu().numpy(), rnsel)
        np.testing.assert_array_equal(ss.cpu().numpy(), rss)
    # the one-sample-per-trip forms of the query kernel: no cell table (more than 4 points kept per cell; a 5^3 window), and a
    # table with a window of more than 32 cells (3 x 3 x 5)
    for extra in (dict(max_points_per_voxel=6), dict(kernel_size=(5, 5, 5)), dict(kernel_size=(3, 3, 5))):
        for level in ("fine", "scaled"):
            cfg = dict(orr.DEFAULT_GRID, grid_level=level, **extra)
            g = ovg.VoxelGridOracle(**cfg)
            g.set_pointset(pts, counts)
            ridx, rloc, rnsel, rss = g.query_dense(x, 8, 2.0, 10)
            hg = hr.HipVoxelGrid(**cfg)
            hg.set_pointset(T(pts).cuda(), T(counts).cuda())
            idx, loc, ss, nsel = hg.query_dense(8, 2.0, 10, x=T(x).cuda())
            np.testing.assert_array_equal(idx.cpu().numpy(), ridx, err_msg=str((extra, level)))
            np.testing.assert_array_equal(nsel.cpu().numpy(), rnsel)
            np.testing.assert_array_equal(loc.cpu().numpy(), rloc)
    # k < 8 and M < 8 paths
    for level in ("fine", "scaled"):
        cfg = dict(orr.DEFAULT_GRID, grid_level=level)
        g = ovg.VoxelGridOracle(**cfg); g.set_pointset(pts, counts)
        hg = hr.HipVoxelGrid(**cfg); hg.set_pointset(T(pts).cuda(), T(counts).cuda())
        ridx, _, rnsel, _ = g.query_dense(x, 3, 2.0, 5)
        idx, _, _, nsel = hg.query_dense(3, 2.0, 5, x=T(x).cuda())
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
        np.testing.assert_array_equal(nsel.cpu().numpy(), rnsel)
        # switching the level on an existing grid rebuilds it
        other = "scaled" if level == "fine" else "fine"
        hg.set_grid_level(other); hg.set_pointset(T(pts).cuda(), T(counts).cuda())
        g2 = ovg.VoxelGridOracle(**dict(cfg, grid_level=other)); g2.set_pointset(pts, counts)
        np.testing.assert_array_equal(hg.query_dense(3, 2.0, 5, x=T(x).cuda())[0].cpu().numpy(), g2.query_dense(x, 3, 2.0, 5)[0])
        hg.set_grid_level(level); hg.set_pointset(T(pts).cuda(), T(counts).cuda())
    with pytest.raises(RuntimeError, match="unsupported"):
        hg.query_dense(9, 2.0, 5, x=T(x).cuda())


@pytest.fixture(params=["tiles", "rows", "tiles32"])
def shade_form(request, monkeypatch):
    """The forms of the shading kernels: LDS tiles of 16 points with the layers on v_mfma_f32_16x16x32_f16 (default since round 5),
    NPCD_SHADE_ROWS=1 (csrc/shade_rows.hip: activations in registers, aggregation as a matrix product), and the tiles with the
    32x32x16 layers of rounds 1-4 in both kernels (NPCD_SHADE_PAIRS16=0 NPCD_SHADE_POINTS16=0); the library reads the switches at every call."""
    for v in ("NPCD_SHADE_ROWS", "NPCD_SHADE_PAIRS16", "NPCD_SHADE_POINTS16"):
        monkeypatch.delenv(v, raising=False)
    if request.param == "rows":
        monkeypatch.setenv("NPCD_SHADE_ROWS", "1")
    elif request.param == "tiles32":
        monkeypatch.setenv("NPCD_SHADE_PAIRS16", "0")
        monkeypatch.setenv("NPCD_SHADE_POINTS16", "0")
    return request.param


def test_shading_matches_reference_golden(golden, shade_form):
    """The reference's own neighbour lists (brute-force branch) and its own sigma / rgb."""
    g = golden("render_brute")
    p = orr.init_field_params(32, seed=int(g["field_seed"]))
    m = _model(32, 64, p)
    sig, rgb = m.field.shade(T(g["nb_idx"]).int().cuda(), T(g["shading_pts"]).cuda(), T(g["coords"]).cuda(), T(g["feats"]).cuda())
    np.testing.assert_allclose(sig.cpu().numpy(), g["sigma"][:, 0], atol=2e-3)
    np.testing.assert_allclose(rgb.cpu().numpy(), g["rgb"], atol=2e-3)


@pytest.mark.parametrize("F_", [32, 128])
def test_shading_large_weights(F_, shade_form):
    """Scaled-up weights (activations ~ O(10)) so that fp16 rounding is actually exercised."""
    p = orr.init_field_params(F_, seed=5)
    for kname in p:
        if kname.endswith("weight"):
            p[kname] = p[kname] * 1.7
    coords, feats = orr.synthetic_cloud(256, F_, 1, seed=2)
    gen = torch.Generator().manual_seed(0)
    P = 1000
    base = coords[0][torch.randint(0, 256, (P,), generator=gen)]
    pts = base + torch.randn(P, 3, generator=gen) * 0.02
    d = torch.cdist(pts, coords[0])
    dist, nb = torch.topk(d, 8, largest=False)
    nb[dist >= 0.08] = -1
    nb[::7, 3:] = -1                                 # ragged neighbour counts
    keep = (nb >= 0).any(dim=1)
    nb, pts = nb[keep], pts[keep]
    sig_ref, rgb_ref, _ = orr.shade_points(p, nb, pts, coords, feats)
    m = _model(F_, 256, p)
    sig, rgb = m.field.shade(nb.int().cuda(), pts.cuda(), coords.cuda(), feats.cuda())
    es = ((sig.cpu() - sig_ref[:, 0]).abs() / sig_ref[:, 0].clamp_min(1.0)).max()
    er = (rgb.cpu() - rgb_ref).abs().max()
    assert float(es) < 5e-3 and float(er) < 5e-3, (float(es), float(er))


def _overflow_case(which, F_=32):
    """weights that push an inter-layer activation past the fp16 range: `pairs` scales the second aggregator layer, `heads` the first
    colour layer (whose output is rounded to fp16 for the next one); `last_heads` scales the LAST colour hidden layer only -- its
    output stays in the fp32 accumulators, which must not raise the guard"""
    p = orr.init_field_params(F_, seed=3)
    key = {"pairs": "aggregator.local_field.2", "heads": "channel_net.0", "last_heads": "channel_net.6"}[which]
    p[key + ".weight"] = p[key + ".weight"] * 3.0e5
    if which != "pairs":       # (default-initialised layers shrink the activations ~0.4 x each: give the heads inputs of O(1) to overflow with)
        for n in ("weight", "bias"):
            p["aggregator.local_field.8." + n] = p["aggregator.local_field.8." + n] * 50.0
    coords, feats = orr.synthetic_cloud(256, F_, 1, seed=4)
    gen = torch.Generator().manual_seed(1)
    P = 700
    base = coords[0][torch.randint(0, 256, (P,), generator=gen)]
    pts = base + torch.randn(P, 3, generator=gen) * 0.02
    dist, nb = torch.topk(torch.cdist(pts, coords[0]), 8, largest=False)
    nb[dist >= 0.08] = -1
    nb[::5, 2:] = -1                                  # points with fewer than 8 neighbours (the aggregation re-reads their last row)
    keep = (nb >= 0).any(dim=1)
    return p, coords, feats, nb[keep].int(), pts[keep]


@pytest.mark.parametrize("which,bit", [("pairs", 1), ("heads", 2), ("last_heads", 0), (None, 0)])
def test_shading_range_guard_raises_its_bit_when_fp16_overflows(which, bit, shade_form):
    """the fused shading kernels round activations to fp16 between layers and used to turn an overflow into NaN
    pixels without a word.  npcd_shade_points(status=) must raise NPCD_SHADE_NONFINITE_PAIRS / _HEADS exactly when a constructed
    overflow happens in that kernel, and nothing for ordinary weights."""
    from npcd.hip import render as hr
    if which is None:
        p = orr.init_field_params(32, seed=3)
        _, coords, feats, nb, pts = _overflow_case("pairs")
    else:
        p, coords, feats, nb, pts = _overflow_case(which)
    m = _model(32, 256, p)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    sig, rgb = m.field.shade(nb.cuda(), pts.cuda(), coords.cuda(), feats.cuda(), status=status)
    torch.cuda.synchronize()
    got = int(status)
    if which == "pairs":
        assert got & hr.SHADE_NONFINITE_PAIRS, got          # (the point kernel then sees non-finite inputs and may add its own bit)
        assert not bool(torch.isfinite(rgb).all() and torch.isfinite(sig).all()) or got & hr.SHADE_NONFINITE_HEADS
    else:
        assert got == bit, (which, got)
    if bit == 0:
        assert bool(torch.isfinite(sig).all()) and bool(torch.isfinite(rgb).all())
    # the word is sticky (never cleared by the library) and optional
    sig2, rgb2 = m.field.shade(nb.cuda(), pts.cuda(), coords.cuda(), feats.cuda(), status=status)
    assert int(status) == got
    sig3, _ = m.field.shade(nb.cuda(), pts.cuda(), coords.cuda(), feats.cuda())
    assert torch.equal(torch.nan_to_num(sig3), torch.nan_to_num(sig2))


def test_render_surfaces_the_range_guard(monkeypatch):
    """PointNeRF.render: out["shading_status"] is 0 for ordinary weights; with overflowing weights the renderer warns (default), raises
    (range_guard="raise"), and on the sync-free path hands the word back as a device scalar that check_shading_status() reads."""
    res = 32
    coords, feats, extr, intr = _scene(res, 1, 512, 32, seed=1)
    args = (coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res)
    m = _model(32, 512, orr.init_field_params(32, seed=0))
    with torch.no_grad():
        out = m.render(*args)
        assert torch.is_tensor(out["shading_status"]) and int(out["shading_status"]) == 0            # sync-free path: device scalar
        m.renderer.sync_free_points = 0                                                               # the path that reads the point count
        out = m.render(*args)
        assert out["shading_status"] == 0
        p = orr.init_field_params(32, seed=0)
        p["aggregator.local_field.2.weight"] = p["aggregator.local_field.2.weight"] * 3.0e5
        bad = _model(32, 512, p)
        bad.renderer.sync_free_points = 0
        with pytest.warns(RuntimeWarning, match="fp16 range"):
            out = bad.render(*args)
        assert out["shading_status"] & 1
        bad.renderer.range_guard = "raise"
        with pytest.raises(FloatingPointError):
            bad.render(*args)
        bad.renderer.sync_free_points = 1 << 23
        out = bad.render(*args)                                   # sync-free: no host read inside the call, the caller checks
        assert torch.is_tensor(out["shading_status"])
        with pytest.raises(FloatingPointError):
            bad.renderer.check_shading_status(out["shading_status"])
        for mode in (1,):                                         # the dense path (brute-force neighbour search) carries the word too
            bad.renderer.range_guard = "off"
            out = bad.renderer(coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res, False, knn_mode=mode)
            assert out["shading_status"] & 1


def test_range_guard_promotes_to_the_fp32_class_on_the_gpu():
    """VolumeRenderer.range_guard = "promote" (VERDICT r5 next 2a): with weights that overflow fp16 the call re-shades the same compact
    lists in the fp32-class kernels (npcd_pairs_x2 / npcd_points_x2) on the GPU and returns THAT -- finite pixels equal to the explicit
    mlp_dtype=torch.float32 render bit for bit and within the fp32-class bar of the fp32 oracle; with ordinary weights nothing is promoted
    and the fp16 kernels' pixels come back unchanged.  Sync-free and counter-reading paths, the dense (brute-force) path, and the
    module's own record of what happened (out["shading_promoted"], out["shading_numerics"])."""
    import warnings
    res = 32
    coords, feats, extr, intr = _scene(res, 1, 512, 32, seed=1)
    args = (coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res)
    p = orr.init_field_params(32, seed=0)
    p["aggregator.local_field.2.weight"] = p["aggregator.local_field.2.weight"] * 3.0e5
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error")                            # a promoted render must not warn: its pixels ARE reference-class
        good = _model(32, 512, orr.init_field_params(32, seed=0))
        plain = good.render(*args)["channels"]
        good.renderer.range_guard = "promote"
        out = good.render(*args)
        assert out["shading_promoted"] is False and out["shading_status"] == 0 and "fp16" in out["shading_numerics"]
        assert torch.equal(out["channels"], plain)
        bad = _model(32, 512, p)
        ref32 = bad.render(*args, mlp_dtype=torch.float32)
        bad.renderer.shade_dtype = None
        bad.renderer.range_guard = "promote"
        for sync_free_points in (1 << 23, 0):
            bad.renderer.sync_free_points = sync_free_points
            out = bad.render(*args)
            assert out["shading_promoted"] is True and out["shading_status"] & 1 and "fp32-class" in out["shading_numerics"]
            assert torch.isfinite(out["channels"]).all() and torch.isfinite(out["mask"]).all()
            assert torch.equal(out["channels"], ref32["channels"]) and torch.equal(out["mask"], ref32["mask"])
        ref = orr.render(p, coords, feats, extr, intr, res=res)
        assert float((out["channels"].cpu() - ref["channels"]).abs().max()) < 1e-3          # (activations of O(1e5): the bar of the fp32-class test)
        dense = bad.renderer(coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res, False, knn_mode=1)
        assert dense["shading_promoted"] is True and torch.isfinite(dense["channels"]).all()
        bad.renderer.range_guard = "nonsense"
        with pytest.raises(ValueError):
            bad.renderer.check_shading_status(1)


@pytest.mark.parametrize("use_dir", [False, True])
def test_fused_point_level_layers_in_the_fp32_class(use_dir, monkeypatch):
    """csrc/points_x2.hip (round 5): the last aggregator layer and both heads on split bf16 operands (three matrix instructions per
    product, fp32 accumulation) against the same layers as float64 torch modules on the same inputs: sigma to 1e-4 of max(1, |sigma|),
    rgb to 2e-5 (16 mantissa bits per product on features of O(10) and weights 1.7 x the default scale: pre-activations of O(100) with
    cancellation; the fp16-operand kernels on such inputs: ~1e-2), a point count that is not a multiple of the 64-point tile, with and
    without view-direction rows, bitwise repeatable.  Field.shade_fp32 reaches the kernel by default (the fp32-class render test
    holds its 2e-5 pixel bar through it); NPCD_FP32_HEADS_LIBRARY=1 selects the fp32 library GEMMs it replaces."""
    from npcd.hip import render as hr
    torch.manual_seed(5)
    from npcd.models.pointnerf import PointNeRF
    p = orr.init_field_params(32, seed=1, dir_dim=51 if use_dir else 0)
    for kname in p:
        if kname.endswith("weight"):
            p[kname] = p[kname] * 1.7
    m = PointNeRF(1, 32, 64, use_dir)
    m.field.load_state_dict(p)
    field = m.cuda().eval().field
    P = 64 * 37 + 13
    G = (torch.randn(P, 256, device="cuda") * 4.0).contiguous()
    pd = torch.nn.functional.normalize(torch.randn(P, 3, device="cuda"), dim=-1) if use_dir else None
    wp = hr.points_x2_pack(field.state_dict(), "cuda")
    db = ray = None
    if use_dir:
        from npcd.models.pointnerf.field import encode_dir
        db = encode_dir(pd, field.dir_freqs) @ field.channel_net[0].weight[:, 256:].float().t()
        ray = torch.arange(P, dtype=torch.int32, device="cuda")
    sig, rgb = hr.points_x2(wp, G, db, ray)
    sig2, rgb2 = hr.points_x2(wp, G, db, ray)
    torch.cuda.synchronize()
    assert torch.equal(sig, sig2) and torch.equal(rgb, rgb2)
    import copy
    f64 = copy.deepcopy(field).double()
    with torch.no_grad():
        feat = f64.aggregator.local_field[8](G.double())
        cin = feat if not use_dir else torch.cat((feat, encode_dir(pd.double(), field.dir_freqs)), dim=-1)
        want_s = torch.nn.functional.softplus(f64.shape_net(feat) - 1.0)[:, 0]
        want_c = torch.sigmoid(f64.channel_net(cin))
    es = float(((sig.double() - want_s).abs() / want_s.abs().clamp_min(1.0)).max())
    ec = float((rgb.double() - want_c).abs().max())
    print("points_x2 vs float64: sigma", es, "rgb", ec)
    assert es < 1e-4 and ec < 2e-5, (es, ec)


@pytest.mark.parametrize("F_", [32, 128])
def test_forward_only_pair_layers_in_the_fp32_class(F_):
    """csrc/points_x2.hip, pairs_x2_kernel (round 5): the four non-linear per-pair layers + the inverse-distance mean on split bf16
    operands, forward only, on lists with holes anywhere in a row, neighbour-less points, 1..8 neighbours (every number of occupied
    16-row blocks of an eight-point tile), a point count that is not a multiple of eight -- against the training kernel of the same
    numerics class (npcd_pair_mlp_fwd, precision 1: rel-L2 1e-5) and against a float64 restatement (3e-5); bitwise repeatable."""
    from npcd.hip import render as hr
    torch.manual_seed(7)
    Np, k, Ntab = 2003, 8, 512
    p = orr.init_field_params(F_, seed=2)
    from npcd.models.pointnerf import PointNeRF
    m = PointNeRF(1, F_, 64, False)
    m.field.load_state_dict(p)
    field = m.cuda().eval().field
    nb = torch.randint(0, Ntab, (Np, k), dtype=torch.int32, device="cuda")
    nb[torch.rand(Np, k, device="cuda") < 0.35] = -1
    nb[5:40] = -1
    for i, keep in enumerate((1, 2, 3, 4, 5, 6, 7)):
        nb[160 + 8 * i:168 + 8 * i, keep:] = -1
    pts = torch.rand(Np, 3, device="cuda") - 0.5
    kp = torch.rand(Ntab, 3, device="cuda") - 0.5
    kf = torch.randn(Ntab, F_, device="cuda")
    wp = hr.pairs_x2_pack(field.state_dict(), F_, "cuda")
    G = hr.pairs_x2(wp, F_, nb, pts, kp, kf)
    G2 = hr.pairs_x2(wp, F_, nb, pts, kp, kf)
    torch.cuda.synchronize()
    assert torch.equal(G, G2) and bool(torch.isfinite(G).all())
    # float64 restatement (aggregators/mlp.py:62-125)
    lf = [field.aggregator.local_field[i] for i in (0, 2, 4, 6)]
    valid = nb >= 0
    idx = nb.clamp_min(0).long()
    rel = (pts[:, None, :] - kp[idx]).double()
    w = 1.0 / (rel.norm(dim=-1) + 1e-5) * valid
    freqs = (2.0 ** torch.arange(10, device="cuda", dtype=torch.float64)) * math.pi
    ang = rel[..., None] * freqs                                            # [P, k, 3, 10]
    enc = torch.cat((torch.sin(ang), torch.cos(ang)), dim=-1).flatten(-2)   # per coordinate: sin f0..9, cos f0..9
    x = torch.cat((kf[idx].double(), rel, enc), dim=-1)
    with torch.no_grad():
        for l in lf:
            x = torch.nn.functional.leaky_relu(torch.nn.functional.linear(x, l.weight.double(), l.bias.double()), 0.01)
    wn = w / w.sum(dim=1, keepdim=True).clamp_min(1e-300)
    want = (x * wn[..., None]).sum(dim=1) * (valid.any(dim=1, keepdim=True))
    e64 = float((G.double() - want).norm() / want.norm())
    # the training kernel's forward on the same lists (valid entries first, as it expects)
    order = torch.argsort((~valid).int(), dim=1, stable=True)
    nbs = torch.gather(nb, 1, order)
    cnt = (nbs >= 0).sum(dim=1)
    off = torch.cumsum(cnt, 0) - cnt
    pack = hr.pair_mlp_pack([l.weight for l in lf], [l.bias for l in lf], F_, hr.PAIR_MLP_X2, "cuda")
    Gt = hr.pair_mlp_forward_raw(kf, None, None, nbs.long(), pts, kp, off, 0, hr.PAIR_MLP_X2, save=False, wpack=pack)[0]
    et = float((G - Gt).norm() / Gt.norm())
    print("pairs_x2: rel-L2 vs float64", e64, "vs the training kernel", et)
    assert e64 < 3e-5 and et < 1e-5, (e64, et)
    empty = ~valid.any(dim=1)
    assert int(empty.sum()) >= 35 and float(G[empty].abs().max()) == 0.0


def test_render_in_the_reference_numerics_class(golden):
    """PointNeRF.render(mlp_dtype=torch.float32) (VERDICT r4 missing 3): the field MLPs in the reference's fp32 class -- per-pair layers
    on the fp32-class matrix-core kernel, heads in fp32 -- against the fp32 CPU oracle: pixels to 2e-5 (the fp16-operand kernels'
    bar is 5e-3), with and without the host read of the point count; with weights that overflow fp16 (the range guard's case) it stays
    finite and on the oracle where the fp16 kernels return NaN; with view directions on the reference's own brute-force branch
    (fixture render_options.npz: 1e-4 instead of the 2e-3 of the fp16 kernels); on the grid path with view directions it agrees with
    the fp16 kernels to their rounding."""
    res = 24
    coords, feats, extr, intr = _scene(res, 2, 512, 32, seed=3)
    p = orr.init_field_params(32, seed=2)
    for name in p:
        if "shape_net.2" in name:
            p[name] = p[name] * 8 + 1.0                       # rays terminate on the object: the shading decides the pixels
    m = _model(32, 512, p)
    args = (coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res)
    ref = orr.render(p, coords, feats, extr, intr, res=res, return_aux=True)
    with torch.no_grad():
        out32 = m.render(*args, mlp_dtype=torch.float32)
        out16 = m.render(*args)
        m.renderer.sync_free_points = 0
        out32b = m.render(*args, mlp_dtype=torch.float32)
        m.renderer.sync_free_points = 1 << 23
    assert out32["shading_numerics"].startswith("fp32-class") and m.renderer.shade_dtype is None
    e32 = float((out32["channels"].cpu() - ref["channels"]).abs().max())
    e16 = float((out16["channels"].cpu() - ref["channels"]).abs().max())
    print("fp32-class render: max |pixel - oracle|", e32, "fp16-operand kernels", e16)
    assert e32 < 2e-5 and e16 < 5e-3 and e32 < e16, (e32, e16)
    assert torch.equal(out32["channels"], out32b["channels"])
    assert float((out32["depth"].cpu() - ref["depth"]).abs().max()) < 1e-4
    # no shading point at all (every ray misses the cloud: principal point far outside the image): white image, no error
    intr_miss = intr.clone()
    intr_miss[..., 0, 2] = 1.0e5
    with torch.no_grad():
        om = m.render(coords.cuda(), feats.cuda(), extr.cuda(), intr_miss.cuda(), res, mlp_dtype=torch.float32)
        om16 = m.render(coords.cuda(), feats.cuda(), extr.cuda(), intr_miss.cuda(), res)
    assert int(om["num_shading_points"]) == 0 and torch.equal(om["channels"], om16["channels"]) and float(om["mask"].abs().max()) == 0.0
    # weights that leave the fp16 range: the fp16 kernels flag it (and return NaN), the fp32-class path follows the oracle
    pb = dict(p)
    pb["aggregator.local_field.2.weight"] = p["aggregator.local_field.2.weight"] * 3.0e5
    bad = _model(32, 512, pb)
    bad.renderer.range_guard = "off"
    refb = orr.render(pb, coords, feats, extr, intr, res=res, return_aux=True)
    with torch.no_grad():
        o16 = bad.render(*args)
        o32 = bad.render(*args, mlp_dtype=torch.float32)
    assert int(o16["shading_status"]) & 1 and not bool(torch.isfinite(o16["channels"]).all())
    assert bool(torch.isfinite(o32["channels"]).all())
    assert float((o32["channels"].cpu() - refb["channels"]).abs().max()) < 1e-3
    # view directions: the reference's own numbers (voxel_grid=None branch) ...
    g = golden("render_options")
    gargs = (T(g["coords"]).cuda(), T(g["feats"]).cuda(), T(g["extr"]).cuda(), T(g["intr"]).cuda(), int(g["res"]), False)
    with torch.no_grad():
        out = _options_model(g, use_dir=True, shade_dtype=torch.float32).renderer(*gargs, knn_mode=1)
    for key in ("mask", "depth", "channels"):
        np.testing.assert_allclose(out[key].cpu().numpy(), g["dir_" + key], atol=1e-4)
    # ... and on the fused grid path against the fp16 kernels
    from npcd.models.pointnerf import PointNeRF
    pd = orr.init_field_params(32, seed=0, dir_dim=51)
    for name in pd:
        if "shape_net.2" in name:
            pd[name] = pd[name] * 8 + 1.0
    md = PointNeRF(1, 32, 512, True)
    md.field.load_state_dict(pd)
    md = md.cuda().eval()
    with torch.no_grad():
        d16, d32 = md.render(*args), md.render(*args, mlp_dtype=torch.float32)
    assert float((d16["channels"] - d32["channels"]).abs().max()) < 5e-3 and torch.equal(d16["mask"] > 0, d32["mask"] > 0)


def test_ordered_compaction_above_32768_rays_uses_block_sums_and_gives_the_same_lists():
    """More than 32,768 rays per call: a first pass adds the per-ray counts up per 1,024 rays and the compaction workgroups read block
    sums instead of every earlier count (the one-level form is quadratic in the rays).  The bases must still be the
    exclusive prefix sum of the counts and the rows those of the dense query, in ray order."""
    from npcd.hip import render as hr
    res, M, k, S, V = 112, 24, 8, 48, 3                           # 37,632 rays in one example
    coords, feats, extr, intr = _scene(res, V, 512, 32, seed=5)
    m = _model(32, 512, orr.init_field_params(32, seed=0))
    agg = m.field.aggregator
    grid = agg.voxel_grid
    with torch.no_grad():
        grid.set_pointset(coords.cuda(), None)
        o, d, t0, t1 = hr.ray_gen(extr.flatten(0, 1).cuda(), intr.flatten(0, 1).cuda(), res, 1.0)
        rays = tuple(t.reshape(1, V * res * res, *t.shape[2:]) for t in (o, d, t0, t1))
        assert rays[0].shape[1] > 32768 and hr.COMPACT_ORDERED
        idx, loc, _, nsel = grid.query_dense(k, agg.r, M, rays=rays, S=S)
        cap = V * res * res * M
        c1, base1, nsel1, bits1, nb1, pts1 = grid.query_compact(k, agg.r, M, rays, S, cap)
    torch.cuda.synchronize()
    valid = (idx[..., 0] >= 0).flatten(0, 1)
    cnt = valid.sum(1)
    P = int(cnt.sum())
    assert c1.tolist() == [P, 0, 0, 0] and P > 20000
    assert torch.equal(base1.long(), torch.cumsum(cnt, 0) - cnt)
    assert torch.equal(nb1[:P], idx.flatten(0, 1)[valid]) and torch.equal(pts1[:P], loc.flatten(0, 1)[valid])


def test_rows_form_of_the_shading_kernel(monkeypatch):
    """csrc/shade_rows.hip against the tile form on neighbour lists with everything the C interface allows: points without any
    neighbour (their sigma / rgb come from the biases alone), -1 entries before valid ones, 1..8 neighbours, a point count that is
    not a multiple of anything, a device-side count above the allocation.  The two forms round differently (LeakyReLU on the
    packed fp16 pair, fp16 aggregation weights): 3e-3 of max(1, |value|).  Same input order -> same windows -> bitwise repeatable."""
    from npcd.hip import render as hr
    torch.manual_seed(1)
    Np, k, F_, Ntab = 3001, 8, 32, 512
    p = orr.init_field_params(F_, seed=0)
    for kname in p:
        if kname.endswith("weight"):
            p[kname] = p[kname] * 1.5
    wp = hr.pack_field_weights(p, F_, "cuda")
    nb = torch.randint(0, Ntab, (Np, k), dtype=torch.int32, device="cuda")
    nb[torch.rand(Np, k, device="cuda") < 0.35] = -1                              # holes anywhere in a row
    nb[5:40] = -1                                                                 # a run of points without neighbours
    nb[torch.rand(Np, device="cuda") < 0.05] = -1
    pts = torch.rand(Np, 3, device="cuda") - 0.5
    kp = torch.rand(Ntab, 3, device="cuda") - 0.5
    kf = torch.randn(Ntab, F_, device="cuda")
    monkeypatch.delenv("NPCD_SHADE_ROWS", raising=False)
    s_t, c_t = hr.shade_points(wp, F_, nb, pts, kp, kf)
    monkeypatch.setenv("NPCD_SHADE_ROWS", "1")
    s_r, c_r = hr.shade_points(wp, F_, nb, pts, kp, kf)
    s_r2, c_r2 = hr.shade_points(wp, F_, nb, pts, kp, kf, n_points=torch.full((1,), 7 * Np, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    assert torch.isfinite(s_r).all() and torch.isfinite(c_r).all()
    assert float(((s_r - s_t).abs() / s_t.abs().clamp_min(1.0)).max()) < 3e-3
    assert float((c_r - c_t).abs().max()) < 3e-3
    assert torch.equal(s_r, s_r2) and torch.equal(c_r, c_r2)
    empty = (nb < 0).all(dim=1)
    assert int(empty.sum()) > 35 and torch.equal(s_r[empty], s_t[empty]) and torch.equal(c_r[empty], c_t[empty])   # zero features in both


@pytest.mark.parametrize("F_", [32, 128])
def test_eight_wave_form_of_the_pair_kernel_is_the_same_bits(monkeypatch, F_):
    """`NPCD_SHADE_PAIRS8=1` (csrc/shade.hip, shade_pairs8_kernel: the tile on 512 threads, a wave owns 32 output channels, four waves
    per SIMD; opt-in, experiments R5.12): the same products in the same order per output element -- sigma / rgb bit-identical to
    the four-wave form on neighbour lists with holes, neighbour-less points, 1..8 neighbours, a clamped device-side count; the
    range guard raises the same bit."""
    from npcd.hip import render as hr
    torch.manual_seed(2)
    Np, k, Ntab = 2999, 8, 512
    p = orr.init_field_params(F_, seed=0)
    wp = hr.pack_field_weights(p, F_, "cuda")
    nb = torch.randint(0, Ntab, (Np, k), dtype=torch.int32, device="cuda")
    nb[torch.rand(Np, k, device="cuda") < 0.35] = -1
    nb[5:40] = -1
    nb[100:164, 1:] = -1                                                          # tiles with few packed rows: 1-3 row blocks
    pts = torch.rand(Np, 3, device="cuda") - 0.5
    kp = torch.rand(Ntab, 3, device="cuda") - 0.5
    kf = torch.randn(Ntab, F_, device="cuda")
    over = torch.full((1,), 7 * Np, dtype=torch.int32, device="cuda")
    monkeypatch.delenv("NPCD_SHADE_PAIRS8", raising=False)
    monkeypatch.setenv("NPCD_SHADE_PAIRS16", "0")                                  # (both are forms of the 32x32x16 layers)
    s4, c4 = hr.shade_points(wp, F_, nb, pts, kp, kf, n_points=over)
    monkeypatch.setenv("NPCD_SHADE_PAIRS8", "1")
    s8, c8 = hr.shade_points(wp, F_, nb, pts, kp, kf, n_points=over)
    torch.cuda.synchronize()
    assert torch.isfinite(s8).all() and torch.equal(s4, s8) and torch.equal(c4, c8)
    # the range guard: weights that overflow fp16 in the pair layers
    big = {kn: (v * 300 if kn.startswith("aggregator.local_field") and kn.endswith("weight") else v) for kn, v in p.items()}
    wpb = hr.pack_field_weights(big, F_, "cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    hr.shade_points(wpb, F_, nb, pts, kp, kf, status=status)
    torch.cuda.synchronize()
    assert int(status) & hr.SHADE_NONFINITE_PAIRS


@pytest.mark.parametrize("F_", [32, 128])
def test_pair_layers_on_both_matrix_instruction_shapes(monkeypatch, F_):
    """The per-pair layers run on v_mfma_f32_16x16x32_f16 by default (round 5, experiments R5.13: the kernel is power-bound and the chip
    holds a higher clock under that instruction; row blocks of 16 instead of 32); `NPCD_SHADE_PAIRS16=0` selects the 32x32x16 layers
    the kernel had before.  Same inputs, same fp16 activations between layers: the two agree to fp16 rounding of a few values (3e-3 of
    max(1, |value|); in practice bit for bit) on lists with holes, neighbour-less points, every number of occupied row blocks, a clamped
    device-side count; both are bitwise repeatable and raise the range-guard bit on overflowing weights."""
    from npcd.hip import render as hr
    torch.manual_seed(3)
    Np, k, Ntab = 3001, 8, 512
    p = orr.init_field_params(F_, seed=0)
    for kname in p:
        if kname.endswith("weight"):
            p[kname] = p[kname] * 1.5
    wp = hr.pack_field_weights(p, F_, "cuda")
    nb = torch.randint(0, Ntab, (Np, k), dtype=torch.int32, device="cuda")
    nb[torch.rand(Np, k, device="cuda") < 0.35] = -1
    nb[5:40] = -1
    for i, keep in enumerate((1, 2, 3, 4, 5, 6, 7)):                               # tiles of 16 points with 16 * keep packed rows
        nb[160 + 16 * i:176 + 16 * i, keep:] = -1
        nb[160 + 16 * i:176 + 16 * i, :keep] = torch.randint(0, Ntab, (16, keep), dtype=torch.int32, device="cuda")
    pts = torch.rand(Np, 3, device="cuda") - 0.5
    kp = torch.rand(Ntab, 3, device="cuda") - 0.5
    kf = torch.randn(Ntab, F_, device="cuda")
    over = torch.full((1,), 7 * Np, dtype=torch.int32, device="cuda")
    # with and without view directions (per-ray bias rows of the first colour layer)
    rays = torch.randint(0, 97, (Np,), dtype=torch.int32, device="cuda")
    dirb = torch.randn(97, 256, device="cuda") * 0.3
    for kw in ({}, {"dir_bias": dirb, "point_ray": rays}):
        monkeypatch.setenv("NPCD_SHADE_PAIRS16", "0")
        monkeypatch.setenv("NPCD_SHADE_POINTS16", "0")
        s32, c32 = hr.shade_points(wp, F_, nb, pts, kp, kf, n_points=over, **kw)
        monkeypatch.delenv("NPCD_SHADE_POINTS16")
        s32p, c32p = hr.shade_points(wp, F_, nb, pts, kp, kf, n_points=over, **kw)       # only the point kernel on the new shape
        monkeypatch.delenv("NPCD_SHADE_PAIRS16")
        s16, c16 = hr.shade_points(wp, F_, nb, pts, kp, kf, n_points=over, **kw)
        s16b, c16b = hr.shade_points(wp, F_, nb, pts, kp, kf, **kw)
        torch.cuda.synchronize()
        assert torch.isfinite(s16).all() and torch.isfinite(c16).all()
        for sx, cx in ((s16, c16), (s32p, c32p)):
            assert float(((sx - s32).abs() / s32.abs().clamp_min(1.0)).max()) < 3e-3 and float((cx - c32).abs().max()) < 3e-3
        assert torch.equal(s16, s16b) and torch.equal(c16, c16b)
    assert float((c16 - hr.shade_points(wp, F_, nb, pts, kp, kf)[1]).abs().max()) > 1e-3          # (the direction rows do enter)
    big = {kn: (v * 300 if kn.startswith("aggregator.local_field") and kn.endswith("weight") else v) for kn, v in p.items()}
    wpb = hr.pack_field_weights(big, F_, "cuda")
    for mode in ("0", "1"):
        monkeypatch.setenv("NPCD_SHADE_PAIRS16", mode)
        monkeypatch.setenv("NPCD_SHADE_POINTS16", mode)
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        hr.shade_points(wpb, F_, nb, pts, kp, kf, status=status)
        torch.cuda.synchronize()
        assert int(status) & hr.SHADE_NONFINITE_PAIRS, mode


def test_ray_march_golden(golden):
    from npcd.hip import render as hr
    g = golden("raymarch")
    Nr, M = g["mask"].shape[2:4]
    mask = T(g["mask"]).reshape(Nr, M)
    per_ray = mask.sum(1).int()
    base = (torch.cumsum(per_ray, 0) - per_ray).int()
    sig_c = T(g["sigma"]).reshape(Nr, M)[mask]
    tot, dep, ch = hr.ray_march(sig_c.cuda(), T(g["rgb_compact"]).cuda(), mask.cuda(), T(g["pts"]).reshape(Nr, M, 3).cuda(),
                                base.cuda(), T(g["o"]).reshape(Nr, 3).cuda(), T(g["d"]).reshape(Nr, 3).cuda(),
                                T(g["ray_end"]).reshape(Nr).cuda(), True)
    np.testing.assert_allclose(tot.cpu().numpy(), g["out_mask"].reshape(Nr), atol=2e-6)
    np.testing.assert_allclose(dep.cpu().numpy(), g["out_depth"].reshape(Nr), atol=2e-5)
    np.testing.assert_allclose(ch.cpu().numpy(), g["out_channels"].reshape(Nr, 3), atol=2e-6)


def test_ray_march_wave_form_against_the_slot_loop(monkeypatch):
    """csrc/geometry.hip ray_march_wave_kernel (one wave per ray: prefix maximum / prefix product / sums by DPP, depth limits as
    per-workgroup pairs combined by the clamp kernel) against the thread-per-ray loop over the slots (NPCD_MARCH_PER_THREAD=1): same
    mask / depth / colour to a few fp32 ulps (scan order instead of slot order), on rays with 0 .. M valid slots, M < 64, a ray
    count that is not a multiple of the workgroup's four rays; the global depth limits (first two scratch words, which the backward
    reads) are identical, and rays without any weight get the clamped depth in both forms."""
    from npcd.hip import render as hr
    g = torch.Generator().manual_seed(3)
    Nr, M = 1003, 48
    valid = torch.rand(Nr, M, generator=g) < 0.3
    valid[:17] = False                                    # empty rays
    valid[17:25] = True                                   # full rays
    o = torch.randn(Nr, 3, generator=g) * 0.1 + torch.tensor([0.0, 0.0, -2.0])
    d = torch.nn.functional.normalize(torch.randn(Nr, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, 1.0]), dim=1)
    tt = torch.sort(torch.rand(Nr, M, generator=g) * 2.0 + 1.0, dim=1).values
    loc = o[:, None, :] + tt[..., None] * d[:, None, :]
    per_ray = valid.sum(1).int()
    base = (torch.cumsum(per_ray, 0) - per_ray).int()
    P = int(per_ray.sum())
    sigma = torch.rand(P, generator=g) * 5.0
    rgb = torch.rand(P, 3, generator=g)
    t1 = torch.full((Nr,), 3.5)
    args = [x.cuda() for x in (sigma, rgb, valid, loc, base, o, d, t1)]
    monkeypatch.setenv("NPCD_MARCH_PER_THREAD", "1")
    m0, d0, c0 = hr.ray_march(*args, True)
    monkeypatch.delenv("NPCD_MARCH_PER_THREAD")
    m1, d1, c1 = hr.ray_march(*args, True)
    torch.cuda.synchronize()
    assert float(m0.max()) > 0.9 and float(m0[:17].abs().max()) == 0.0
    assert float((m1 - m0).abs().max()) < 2e-6 and float((c1 - c0).abs().max()) < 2e-6
    assert float((d1 - d0).abs().max()) < 2e-5
    assert torch.equal(d1[:17], d0[:17])                  # no weight: the depth is the clamp limit, identical in both forms


def test_render_matches_reference_golden_brute(golden):
    """End to end against the reference's own rendering (its voxel_grid=None branch)."""
    g = golden("render_brute")
    p = orr.init_field_params(32, seed=int(g["field_seed"]))
    m = _model(32, 64, p)
    agg = m.field.aggregator
    agg.max_shading_pts, agg.k = int(g["M"]), int(g["k"])
    m.renderer.depth_resolution = int(g["S"])
    with torch.no_grad():
        out = m.renderer(T(g["coords"]).cuda(), T(g["feats"]).cuda(), T(g["extr"]).cuda(), T(g["intr"]).cuda(),
                         int(g["res"]), False, knn_mode=1)
    np.testing.assert_allclose(out["mask"].cpu().numpy(), g["out_mask"], atol=2e-3)
    np.testing.assert_allclose(out["channels"].cpu().numpy(), g["out_channels"], atol=2e-3)
    np.testing.assert_allclose(out["depth"].cpu().numpy(), g["out_depth"], atol=2e-3)


def _options_model(g, use_dir=False, **renderer_kw):
    from npcd.models.pointnerf import PointNeRF
    m = PointNeRF(1, 32, 64, use_dir)
    m.field.load_state_dict(orr.init_field_params(32, seed=int(g["field_seed"]), dir_dim=51 if use_dir else 0))
    agg = m.field.aggregator
    agg.max_shading_pts, agg.k = int(g["M"]), int(g["k"])
    m.renderer.depth_resolution = int(g["S"])
    for key, val in renderer_kw.items():
        setattr(m.renderer, key, val)
    return m.cuda().eval()


def test_renderer_options_match_reference_golden(golden):
    """The renderer options outside the published configuration, each against the reference's own voxel_grid=None branch
    (fixture render_options.npz, tests/golden/make_golden.py gen_render_options):
      use_view_dir (models/npcd.py:8 -> fields/mlp.py:30-36,67-70), ray_limits (renderers/renderer.py:44-46),
      return_kp_weights (renderer.py:177-184,254-259 + aggregators/mlp.py:84,93-98).
    disparity_space_sampling cannot be pinned: the reference's branch raises (renderer.py:62-65 expands a [1,1,S,1] tensor to
    [N,M,1,1]; the fixture records disparity_runs = 0) -- its intent is checked by properties in the next test."""
    g = golden("render_options")
    args = (T(g["coords"]).cuda(), T(g["feats"]).cuda(), T(g["extr"]).cuda(), T(g["intr"]).cuda(), int(g["res"]), False)
    assert int(g["disparity_runs"]) == 0
    with torch.no_grad():
        out = _options_model(g, use_dir=True).renderer(*args, knn_mode=1)
        for key in ("mask", "depth", "channels"):
            np.testing.assert_allclose(out[key].cpu().numpy(), g["dir_" + key], atol=2e-3)
        m = _options_model(g, ray_limits=(float(g["lim_near"]), float(g["lim_far"])))
        out = m.renderer(*args, knn_mode=1)
        for key in ("mask", "depth", "channels"):
            np.testing.assert_allclose(out[key].cpu().numpy(), g["lim_" + key], atol=2e-3)
        m = _options_model(g)
        out = m.renderer(*args, knn_mode=1, return_kp_weights=True)
    assert out["kp_weights"].shape == g["kpw"].shape == (1, 2, int(g["res"]) ** 2, 64)
    np.testing.assert_allclose(out["kp_weights"].cpu().numpy(), g["kpw"], atol=2e-3)
    np.testing.assert_allclose(out["channels"].cpu().numpy(), g["kpw_channels"], atol=2e-3)
    # a ray's key-point weights add up to its opacity (normalised pair weights x march weights)
    np.testing.assert_allclose(out["kp_weights"].sum(-1, keepdim=True).cpu().numpy(), out["mask"].cpu().numpy(), atol=1e-4)


def test_view_directions_on_the_fused_grid_path_and_in_training():
    """use_view_dir on the voxel-grid paths: (1) the fused path (compact lists, ray of a row by binary search over the ray bases)
    equals the dense-table path bit for bit (same kernels, same row order); (2) the colours depend on the direction columns;
    (3) the training-mode forward (torch heads on [feat | enc(direction)]) agrees with the fused kernels to fp16 rounding and the
    direction columns of channel_net.0 receive a gradient."""
    from npcd.models.pointnerf import PointNeRF
    res = 32
    coords, feats, extr, intr = _scene(res, 2, 512, 32, seed=1, B=2)
    coords[1] = coords[1].flip(-1) * 0.8
    p = orr.init_field_params(32, seed=0, dir_dim=51)
    for kname in p:
        if "shape_net.2" in kname:
            p[kname] = p[kname] * 8 + 1.0
    m = PointNeRF(1, 32, 512, True)
    m.field.load_state_dict(p)
    m = m.cuda().eval()
    args = (coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res)
    with torch.no_grad():
        fused = m.render(*args)
        dense = m.renderer(*args, False, return_kp_weights=True)
        for key in ("mask", "depth", "channels"):
            assert torch.equal(fused[key], dense[key]), key
        m.field.channel_net[0].weight[:, 256:] *= -1.0
        flipped = m.render(*args)
        m.field.channel_net[0].weight[:, 256:] *= -1.0
    assert float(fused["mask"].max()) > 0.5
    assert float((flipped["channels"] - fused["channels"]).abs().max()) > 1e-3
    assert torch.equal(flipped["mask"], fused["mask"])
    # training-mode forward with the jitter replayed as zero = the evaluation samples
    m.renderer.randomize_depth_samples = True
    S = m.renderer.depth_resolution
    out = m.renderer(*args, False, rng={"jitter": torch.zeros(2, 2, res * res, S)})
    for key in ("mask", "channels"):
        assert float((out[key].detach() - fused[key]).abs().max()) < 1e-2, key
    out["channels"].square().sum().backward()
    gw = m.field.channel_net[0].weight.grad
    assert float(gw[:, 256:].abs().max()) > 0 and float(gw[:, :256].abs().max()) > 0


def test_disparity_space_sampling_and_fixed_limits_on_the_grid_path():
    """disparity_space_sampling (renderer.py:60-68; unpinnable, see above): the depth samples are evenly spaced in 1 / depth
    between the limits, jittered forward by less than one spacing, monotonic; with the jitter replayed the render is
    reproducible and close to the depth-space render of the same scene (same object, denser samples near the camera).
    ray_limits on the fused grid path: limits inside the box's span give the same image as the box limits wherever the object
    lies between them."""
    res = 32
    coords, feats, extr, intr = _scene(res, 1, 512, 32, seed=1, B=1)
    p = orr.init_field_params(32, seed=0)
    for kname in p:
        if "shape_net.2" in kname:
            p[kname] = p[kname] * 8 + 1.0
    m = _model(32, 512, p)
    args = (coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res)
    ren = m.renderer
    t0, t1 = torch.tensor([0.5, 0.8], device="cuda"), torch.tensor([2.0, 1.9], device="cuda")
    S = ren.depth_resolution
    jit = torch.rand(2, S)
    dep = ren.disparity_depths(t0, t1, jit)
    inv = 1.0 / dep
    u = (1.0 / t0[:, None] - inv) / (1.0 / t0 - 1.0 / t1)[:, None]                # position in disparity space, 0..1 (+ jitter)
    grid_u = torch.arange(S, device="cuda") / (S - 1)
    assert float((u - grid_u - jit.cuda() / (S - 1)).abs().max()) < 1e-4
    assert bool((dep[:, 1:] > dep[:, :-1]).all()) and bool((dep[:, 0] >= t0 - 1e-6).all())
    with torch.no_grad():
        base = m.render(*args)
        ren.disparity_space_sampling = True
        jit = torch.rand(1, 1, res * res, S)
        a = ren(*args, False, rng={"jitter_disp": jit})
        b = ren(*args, False, rng={"jitter_disp": jit})
        ren.disparity_space_sampling = False
        for key in ("mask", "depth", "channels"):
            assert torch.equal(a[key], b[key])
        assert orr.psnr(orr.unflatten_image(a["channels"].cpu()), orr.unflatten_image(base["channels"].cpu())) > 20.0
        ren.ray_limits = (0.3, 2.3)            # camera at 1.3 from the origin, object within 0.5: the whole object lies inside
        wide = m.render(*args)
        ren.ray_limits = None
    assert orr.psnr(orr.unflatten_image(wide["channels"].cpu()), orr.unflatten_image(base["channels"].cpu())) > 20.0


@pytest.mark.parametrize("res,B,views,S", [(32, 2, 2, 128), (128, 1, 1, 128), (128, 1, 1, 64)])
def test_render_vs_oracle_grid(res, B, views, S, shade_form):
    """Full pipeline vs the oracle with voxel-grid semantics: 128 depth samples per ray is the reference's code
    (pointnerf.py:184), (128, 1, 1, 64) is BASELINE.json configs[2] as written (128x128, k=8, 64 samples/ray)."""
    coords, feats, extr, intr = _scene(res, views, 512, 32, seed=1, B=B)
    if B > 1:
        coords[1] = coords[1].flip(-1) * 1.3              # a different cloud per batch element
    p = orr.init_field_params(32, seed=0)
    for kname in p:
        if "shape_net.2" in kname:
            p[kname] = p[kname] * 8 + 1.0                 # raise densities so the object is opaque-ish
    m = _model(32, 512, p)
    m.renderer.count_pairs = True
    m.renderer.depth_resolution = S
    with torch.no_grad():
        out = m.render(coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res)
    assert out["grid_level"] == m.voxel_grid.grid_level
    ref = orr.render(p, coords, feats, extr, intr, res=res, S=S, mode="grid", return_aux=True)
    assert out["num_shading_points"] == ref["aux"]["P"] and out["num_pairs"] == ref["aux"]["Q"]
    assert float(ref["mask"].max()) > 0.5
    for key in ("mask", "depth", "channels"):
        err = float((out[key].cpu() - ref[key]).abs().max())
        assert err < 5e-3, (key, err)
    img_h = orr.unflatten_image(out["channels"].cpu())
    img_r = orr.unflatten_image(ref["channels"])
    assert orr.psnr(img_h, img_r) > 50.0


def test_compact_lists_are_in_ray_order_and_reproducible(monkeypatch):
    """npcd_grid_query_compact_ordered: ray_base is the exclusive prefix sum of the per-ray counts (ray order), the rows of a ray are
    the valid slots of the dense query in slot order, two calls give identical buffers bit for bit, and the one-launch atomic form
    (NPCD_COMPACT_ORDERED=0) holds the same rows per ray at whatever base the race gave it.  Ragged: R is not a multiple of 256
    and the first / last rays miss the cloud."""
    from npcd.hip import render as hr
    res, M, k, S = 50, 50, 8, 64                                   # 2500 rays per view, 2 views, 2 clouds
    coords, feats, extr, intr = _scene(res, 2, 512, 32, seed=2, B=2)
    coords[1] = coords[1].flip(-1) * 0.7
    m = _model(32, 512, orr.init_field_params(32, seed=0))
    agg = m.field.aggregator
    grid = agg.voxel_grid
    B, V = 2, 2
    with torch.no_grad():
        grid.set_pointset(coords.cuda(), None)
        o, d, t0, t1 = hr.ray_gen(extr.flatten(0, 1).cuda(), intr.flatten(0, 1).cuda(), res, 1.0)
        rays = tuple(t.reshape(B, V * res * res, *t.shape[2:]) for t in (o, d, t0, t1))
        idx, loc, _, nsel = grid.query_dense(k, agg.r, M, rays=rays, S=S)
        cap = B * V * res * res * M
        assert hr.COMPACT_ORDERED
        c1, base1, nsel1, bits1, nb1, pts1 = grid.query_compact(k, agg.r, M, rays, S, cap)
        c2, base2, _, bits2, nb2, pts2 = grid.query_compact(k, agg.r, M, rays, S, cap)
        monkeypatch.setattr(hr, "COMPACT_ORDERED", False)
        c3, base3, _, bits3, nb3, pts3 = grid.query_compact(k, agg.r, M, rays, S, cap)
    torch.cuda.synchronize()
    valid = (idx[..., 0] >= 0).flatten(0, 1)                         # [B*R, M]
    cnt = valid.sum(1)
    P = int(cnt.sum())
    assert int(c1[0]) == P and int(c1[1]) == 0 and int(c3[0]) == P and P > 1000
    assert int((cnt == 0).sum()) > 100                               # rays that miss
    assert torch.equal(base1.long(), torch.cumsum(cnt, 0) - cnt)
    assert torch.equal(nsel1, nsel.flatten())
    rows_nb, rows_pts = idx.flatten(0, 1)[valid], loc.flatten(0, 1)[valid]      # ray-major, slot order: the ordered layout
    assert torch.equal(nb1[:P], rows_nb) and torch.equal(pts1[:P], rows_pts)
    for a, b in ((c1, c2), (base1, base2), (bits1, bits2), (nb1[:P], nb2[:P]), (pts1[:P], pts2[:P])):
        assert torch.equal(a, b)
    # atomic form: same masks, same rows per ray at its own base
    assert torch.equal(bits3, bits1)
    ray_of_row = torch.repeat_interleave(torch.arange(cnt.numel(), device="cuda"), cnt)
    within = torch.arange(P, device="cuda") - (torch.cumsum(cnt, 0) - cnt)[ray_of_row]
    src = base3.long()[ray_of_row] + within
    assert torch.equal(nb3[src], rows_nb) and torch.equal(pts3[src], rows_pts)


@pytest.mark.parametrize("ordered", [True, False])
def test_compact_list_overflow_retries_without_out_of_bounds_access(ordered, monkeypatch):
    """The compact shading-point lists are sized for a FRACTION of the worst case once a call is too large for worst-case
    buffers; when they overflow the query only raises a flag and keeps counting, the shading / ray-march kernels run on the
    clamped lists, and the host retries with worst-case buffers.  Force that path (sync_free_points = 0, a fraction far
    below the ~10 % slot fill of this scene) and require the result to equal the worst-case-buffer render bit for bit;
    canaries allocated right behind the first attempt's buffers must stay untouched."""
    from npcd.hip import render as hr
    monkeypatch.setattr(hr, "COMPACT_ORDERED", ordered)
    res = 64
    coords, feats, extr, intr = _scene(res, 2, 512, 32, seed=1, B=1)
    p = orr.init_field_params(32, seed=0)
    for kname in p:
        if "shape_net.2" in kname:
            p[kname] = p[kname] * 8 + 1.0
    m = _model(32, 512, p)
    args = (coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res)
    with torch.no_grad():
        ref = m.render(*args)
        P = int(ref["num_shading_points"])
        m.renderer.sync_free_points = 0
        m.renderer.capacity_fraction = 0.25 * P / (2 * res * res * 50)          # a quarter of what is needed
        calls = []
        grid = m.field.aggregator.voxel_grid
        orig, orig_rays = grid.query_compact, grid.query_compact_rays

        def spy(*a, **k):
            out = orig(*a, **k)
            calls.append(int(out[4].shape[0]))
            return out

        def spy_rays(*a, **k):           # (the ordered form can generate its rays inside the query launch: NPCD_RENDER_FUSED_RAYS=1)
            out = orig_rays(*a, **k)
            if out is not None:
                calls.append(int(out[6].shape[0]))
            return out
        grid.query_compact, grid.query_compact_rays = spy, spy_rays
        try:
            out = m.render(*args)
        finally:
            grid.query_compact, grid.query_compact_rays = orig, orig_rays
    assert len(calls) == 2 and calls[0] < P <= calls[1], (calls, P)              # overflowed once, then the worst case
    assert int(out["num_shading_points"]) == P
    for key in ("mask", "depth", "channels"):
        assert torch.equal(out[key], ref[key]), key


def test_shade_kernels_clamp_device_count_to_allocated_rows():
    """npcd_shade_points reads the point count from device memory; a count above `max_points` (what an overflowed compact
    query leaves behind) must not make the kernels touch rows past the allocation: guard rows behind sigma / rgb keep
    their fill value and the first max_points rows equal a plain call."""
    from npcd.hip import render as hr
    torch.manual_seed(0)
    Np, k, F_, Ntab = 1000, 8, 32, 512
    p = orr.init_field_params(F_, seed=0)
    wp = hr.pack_field_weights(p, F_, "cuda")
    nb = torch.randint(-1, Ntab, (Np, k), dtype=torch.int32, device="cuda")
    nb, _ = torch.sort(nb, dim=1, descending=True)                              # -1 pads last
    pts = torch.rand(Np, 3, device="cuda") - 0.5
    kp = torch.rand(Ntab, 3, device="cuda") - 0.5
    kf = torch.randn(Ntab, F_, device="cuda")
    s_ref, c_ref = hr.shade_points(wp, F_, nb, pts, kp, kf)
    big = torch.full((1,), 50 * Np, dtype=torch.int32, device="cuda")           # the counter of an overflowed query
    s, c = hr.shade_points(wp, F_, nb, pts, kp, kf, n_points=big)
    torch.cuda.synchronize()
    assert torch.equal(s, s_ref) and torch.equal(c, c_ref)


def test_grid_render_is_tied_to_the_pinned_brute_force_branch():
    """The voxel-grid semantics of torch_knnquery are a spec of this build (third-party source absent: DESIGN section 2 / 3);
    the reference's in-repo branch (`voxel_grid is None`, aggregator.py:42-58: exact radius ball) IS pinned end to end by
    render_brute.npz.  This test ties BOTH readings of the grid (grid_level "fine" / "scaled", include/npcd_hip.h) to that branch
    ON THE GPU, on the benchmark scene (bench.py render leg: 512-point ellipsoid, 128 x 128, k = 8, M = 50, S = 128 and 64):
      * every grid neighbour lies inside the radius ball of its shading point (grid result is a subset of the in-radius set),
      * a shading slot of the grid path that the brute-force path also has carries the same position, and where the
        in-radius set has fewer than k members the grid list is a subset of it,
      * identical-neighbour-set rate, keypoints dropped by the per-voxel cap, keypoints the TV loss's self-query
        (neural_point_cloud_tv_loss.py:41-43) loses, and PSNR(grid render, brute render) are printed per reading for DESIGN.md;
        bars: fine > 0.25 identical, scaled > 0.35 identical, PSNR > 27 dB both (random MLP weights)."""
    from npcd.hip import render as hr
    res, M, k = 128, 50, 8
    coords, feats, extr, intr = _scene(res, 1, 512, 32, seed=0, B=1)
    p = orr.init_field_params(32, seed=0)
    for kname in p:
        if "shape_net.2" in kname:
            p[kname] = p[kname] * 8 + 1.0
    m = _model(32, 512, p)
    agg = m.field.aggregator
    grid = agg.voxel_grid
    bars = {"fine": (0.25, 27.0), "scaled": (0.35, 27.0)}
    for level in ("fine", "scaled"):
        grid.set_grid_level(level)
        with torch.no_grad():
            grid.set_pointset(coords.cuda(), None)
            kept = (grid.workspace[:512 * 4].view(torch.int32) >> 30) & 1
            ti, _, _, _ = grid.query_dense(k, agg.r, 1, x=coords.cuda().view(1, 512, 1, 3), mode=0)       # the TV loss's self-query
        n_dropped, n_lost = int((kept == 0).sum()), int((ti[0, :, 0, 0] < 0).sum())
        print(f"\n[grid_level={level}] keypoints dropped by the {grid.params.max_points_per_voxel}-per-voxel cap: {n_dropped} / 512; "
              f"keypoints without any neighbour on their own position: {n_lost}")
        for S in (128, 64):
            m.renderer.depth_resolution = S
            with torch.no_grad():
                grid.set_pointset(coords.cuda(), None)
                o, d, t0, t1 = hr.ray_gen(extr.flatten(0, 1).cuda(), intr.flatten(0, 1).cuda(), res, 1.0)
                rays = (o, d, t0, t1)
                gi, gl, gs, gn = grid.query_dense(k, agg.r, M, rays=rays, S=S, mode=0)
                bi, bl, bs, bn = grid.query_dense(k, agg.scaled_r, M, rays=rays, S=S, mode=1)
                out_g = m.renderer(coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res, False, knn_mode=0)
                out_b = m.renderer(coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), res, False, knn_mode=1)
            kp = coords.cuda().reshape(-1, 3)
            r2 = torch.tensor(agg.scaled_r, dtype=torch.float32) ** 2
            # (1) subset: every grid neighbour is within the radius of its shading point
            valid = gi >= 0
            dist2 = ((gl[..., None, :] - kp[gi.clamp_min(0).long()]) ** 2).sum(-1)
            assert bool((dist2[valid] < float(r2) * (1 + 1e-6)).all())
            # (2) slot-by-slot: match the two paths through the depth-sample number of each slot
            R = gi.shape[1]
            gmap = torch.full((R, S), -1, dtype=torch.long, device="cuda")
            rr = torch.arange(R, device="cuda")[:, None].expand(R, M)
            gv = gs[0] >= 0
            gmap[rr[gv], gs[0][gv].long()] = torch.arange(M, device="cuda")[None].expand(R, M)[gv]
            bv = bs[0] >= 0
            slot_in_grid = gmap[rr[bv], bs[0][bv].long()]                       # grid slot of every brute-force slot (or -1)
            # a brute-force slot (has an in-radius neighbour) may be absent from the grid path only when the grid ran out of slots
            # earlier on that ray (its M slots also hold neighbour-less samples) or when no neighbour lies in the window
            present = slot_in_grid >= 0
            b_idx = bi[0][bv]
            g_idx = gi[0][rr[bv][present], slot_in_grid[present]]
            b_sorted = torch.sort(torch.where(b_idx[present] < 0, torch.full_like(b_idx[present], 1 << 30), b_idx[present]), dim=1).values
            g_sorted = torch.sort(torch.where(g_idx < 0, torch.full_like(g_idx, 1 << 30), g_idx), dim=1).values
            same = (b_sorted == g_sorted).all(dim=1)
            assert torch.equal(gl[0][rr[bv][present], slot_in_grid[present]], bl[0][bv][present])    # same sample position, bit for bit
            n_b, n_present, n_same = int(bv.sum()), int(present.sum()), int(same.sum())
            img_g = orr.unflatten_image(out_g["channels"].cpu())
            img_b = orr.unflatten_image(out_b["channels"].cpu())
            psnr = orr.psnr(img_g, img_b)
            print(f"[grid-vs-brute grid_level={level} S={S}] brute slots {n_b}, present in grid path {n_present} ({n_present / n_b:.4f}), "
                  f"identical neighbour sets {n_same} ({n_same / max(n_present, 1):.4f}); grid slots with a neighbour {int((gi[..., 0] >= 0).sum())}; "
                  f"PSNR(grid render, brute render) = {psnr:.2f} dB; max |rgb diff| = "
                  f"{float((out_g['channels'] - out_b['channels']).abs().max()):.4f}")
            # where the radius ball holds fewer than k points the brute-force list IS the ball: the grid list must be a subset
            few = (b_idx[present] >= 0).sum(dim=1) < k
            sub = ((g_idx[few][:, :, None] == b_idx[present][few][:, None, :]).any(-1) | (g_idx[few] < 0)).all(dim=1)
            assert bool(sub.all())
            assert n_present / n_b > 0.98 and n_same / n_present > bars[level][0]
            assert psnr > bars[level][1]
    grid.set_grid_level(hr.DEFAULT_GRID_LEVEL)


def test_pointnerf_forward_surface():
    """PointNeRF.forward(obj_idx, intrinsics, extrinsics, sample_rays) -> (pred, aux) like pointnerf.py:56-105."""
    from npcd.models import NPCD
    coords, feats, extr, intr = _scene(16, 1, 512, 32, seed=3, B=2)
    net = NPCD(n_obj=5, coords_dim=3, feats_dim=32, num_points=512, use_view_dir=False, width=64, layers=1, heads=1,
               pointnerf_only=True).cuda().eval()
    net.pointnerf.opt.sizes.default_resolution = 16               # 128 in the reference (pointnerf.py:192)
    net.pointnerf.set_all_coords(torch.cat([coords, coords, coords[:1]]).cuda())
    with torch.no_grad():
        table = net.pointnerf.feats.get_emb().weight.view(5, 512, 64)
        table[:2, :, :32] = feats.cuda()
        pred, aux = net.pointnerf(torch.tensor([0, 1]).cuda(), intr.cuda(), extr.cuda(), sample_rays=False)
    assert pred.mask.shape == (2, 1, 256, 1) and pred.channels.shape == (2, 1, 256, 3) and pred.depth.shape == (2, 1, 256, 1)
    assert set(aux) == {"coords", "feats", "feats_mean", "feats_log_var", "feats_std"}
    fp = {k: v.cpu() for k, v in net.pointnerf.field.state_dict().items()}
    ref = orr.render(fp, coords, feats, extr, intr, res=16)
    assert float((pred.channels.cpu() - ref["channels"]).abs().max()) < 5e-3
    with torch.no_grad():
        r2 = net.pointnerf.render(coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda(), resolution=16, max_shading_points=20)
    ref2 = orr.render(fp, coords, feats, extr, intr, res=16, M=20)
    assert float((r2.channels.cpu() - ref2["channels"]).abs().max()) < 5e-3
    assert net.pointnerf.field.aggregator.max_shading_pts == 50
    assert net.pointnerf.get_all_coords().shape == (5, 512, 3) and net.pointnerf.get_all_feats().shape == (5, 512, 32)
    # sample_rays=True (training mode) is covered by tests/test_gpu_train_render.py


def test_evaluation_protocol_counterpart():
    """npcd.eval.evaluate_pointnerf (reference pointnerf_evaluation.py:152-257): per-view renders with the burn-in rule, PSNR against
    images rendered by the CPU oracle from the same weights (>= 50 dB: the HIP path IS the model, up to f16 shading), timing records
    only after the burn-in objects."""
    from npcd.eval import evaluate_pointnerf
    from npcd.models import NPCD
    res = 16
    coords, feats, extr, intr = _scene(res, 2, 512, 32, seed=3, B=1)
    net = NPCD(n_obj=5, coords_dim=3, feats_dim=32, num_points=512, use_view_dir=False, width=64, layers=1, heads=1, pointnerf_only=True).cuda().eval()
    net.pointnerf.opt.sizes.default_resolution = res
    net.pointnerf.set_all_coords(coords.expand(5, -1, -1).contiguous().cuda())
    with torch.no_grad():
        net.pointnerf.feats.get_emb().weight.view(5, 512, 64)[:, :, :32] = feats.cuda()
        for name, p_ in net.pointnerf.field.named_parameters():
            if "shape_net.2" in name:
                p_.mul_(8).add_(1.0)
    fp = {k: v.cpu() for k, v in net.pointnerf.field.state_dict().items()}
    ref = orr.render(fp, coords, feats, extr, intr, res=res)
    images = orr.unflatten_image(ref["channels"])                                  # [1, V, 3, res, res]
    samples = [{"obj_idx": torch.tensor([i]), "intrinsics": intr, "extrinsics": extr, "images": images} for i in range(5)]
    out = evaluate_pointnerf(net.pointnerf, samples, eval_batch_size=1, burn_in_samples=3)
    assert len(out["views"]) == 10 and out["psnr"] > 50.0
    timed = [r for r in out["views"] if r["runtime_model_in_msec"] == r["runtime_model_in_msec"]]
    assert len(timed) == 4 and all(r["sample"] >= 3 for r in timed) and out["runtime_model_in_msec"] > 0


def test_grid_build_is_skipped_for_an_unchanged_cloud_and_redone_after_a_write():
    """HipVoxelGrid.set_pointset recognises the cloud of the previous call (same storage, same version counter, same geometry) and
    does not rebuild; an in-place write bumps the version counter and the grid follows.  Results are those of a fresh grid."""
    from npcd.hip import render as hrender
    coords, feats, extr, intr = _scene(32, 1, 512, 32, seed=3)
    p = orr.init_field_params(32, seed=0)
    for kname in p:
        if "shape_net.2" in kname:
            p[kname] = p[kname] * 8 + 1.0
    m = _model(32, 512, p)
    c, f, e, k = coords.cuda(), feats.cuda(), extr.cuda(), intr.cuda()
    builds = []

    class CountingLib:                                         # the library with npcd_grid_build counted
        def __init__(self, L):
            self._L = L

        def __getattr__(self, name):
            fn = getattr(self._L, name)
            if name != "npcd_grid_build":
                return fn
            return lambda *args: (builds.append(1), fn(*args))[1]

    orig, proxy = hrender.lib, CountingLib(hrender.lib())
    hrender.lib = lambda: proxy
    try:
        with torch.no_grad():
            a = m.render(c, f, e, k, 32)
            b = m.render(c, f, e, k, 32)                      # same cloud: no second build
            assert len(builds) == 1 and torch.equal(a["channels"], b["channels"])
            c.mul_(0.9)                                        # in-place write -> version bump -> rebuild
            d = m.render(c, f, e, k, 32)
            assert len(builds) == 2
    finally:
        hrender.lib = orig
    m2 = _model(32, 512, p)                                    # a fresh grid on the modified cloud
    with torch.no_grad():
        d2 = m2.render(c.clone(), f, e, k, 32)
    assert torch.equal(d["channels"], d2["channels"]) and not torch.equal(a["channels"], d["channels"])


def test_generate_then_render_loop_of_the_diffusion_evaluation():
    """npcd.eval.sample_and_render (reference diffusion_evaluation.py:146-183) on a small NPCD model with a short diffusion chain: the
    bundled 251 SRN-cars test poses load, every generated cloud is rendered from every pose, images are 8-bit quantised values in
    [0, 1] and identical to a direct PointNeRF.render of the same cloud."""
    from npcd.eval import load_test_poses, sample_and_render, unflatten_pred
    from npcd.models import NPCD
    from npcd.models.diffusion.gaussian_diffusion import GaussianDiffusion
    poses, intr = load_test_poses("srncars")
    assert poses.shape == (251, 4, 4) and intr.shape == (251, 3, 3)
    torch.manual_seed(0)
    net = NPCD(n_obj=1, coords_dim=3, feats_dim=32, num_points=512, use_view_dir=False, width=64, layers=1, heads=1).cuda().eval()
    net.diffusion.diffusion_process = GaussianDiffusion(num_timesteps=8).cuda()          # a short chain: the loop is what is tested
    with torch.no_grad():                                                              # clip ranges / scales of a "trained" normaliser
        net.diffusion.coords_normalization.min.fill_(-2.5); net.diffusion.coords_normalization.max.fill_(2.5)
        net.diffusion.coords_normalization.scale.fill_(0.25)
        net.diffusion.feats_normalization.min.fill_(-1.0); net.diffusion.feats_normalization.max.fill_(1.0)
    got = []
    sub = slice(0, 251, 50)                                                            # 6 of the poses keep the test short
    res = sample_and_render(net, poses[sub], intr[sub], num_samples=3, generate_batch_size=2, render_batch_size=4, resolution=32,
                            feed=lambda im: got.append(im.cpu()))
    assert res["clouds"] == 3 and res["poses_per_cloud"] == 6 and len(got) == 3 and got[0].shape == (6, 3, 32, 32)
    for im in got:
        assert float(im.min()) >= 0.0 and float(im.max()) <= 1.0
        assert torch.equal(torch.round(im * 255), im * 255) or float((torch.round(im * 255) - im * 255).abs().max()) < 1e-4
    assert res["views_per_s"] > 0 and res["generate_seconds"] > 0


@pytest.mark.parametrize("S", [128, 64])
def test_rays_generated_inside_the_query_launch_are_the_same_bits(S, monkeypatch):
    """Round 6: the fused render generates its rays inside the neighbour-query launch and ends rays that miss the cube
    at the global end inside the march (npcd_render_rays_query / npcd_ray_march_compact_fused) -- two
    launches per view fewer.  Against the separate launches (NPCD_RENDER_FUSED_RAYS=0: ray_gen + limits fix-up, query, march): image,
    mask, depth, point count BIT for bit -- every ray hitting the cube, some rays missing it (their depth comes from the
    global limits), all rays missing, two examples x three views per call, the counter-reading path and the sync-free one, a second
    render through the same objects."""
    from npcd.hip import render as hr
    res = 40
    coords, feats, extr, intr = _scene(res, 3, 512, 32, seed=2, B=2)
    p = orr.init_field_params(32, seed=3)
    for name in p:
        if "shape_net.2" in name:
            p[name] = p[name] * 8 + 1.0
    m = _model(32, 512, p)
    m.renderer.depth_resolution = S
    intr_some = intr.clone()
    intr_some[..., 0, 0] = intr_some[..., 1, 1] = 131.25 * res / 128 * 0.45          # zoomed out: the corner rays miss the cube
    intr_none = intr.clone()
    intr_none[..., 0, 2] = 1.0e5                                                       # every ray misses
    for tag, K in (("hit", intr), ("some miss", intr_some), ("all miss", intr_none)):
        for sync_free_points in (1 << 23, 0):
            m.renderer.sync_free_points = sync_free_points
            outs = {}
            for fused in (True, False, True):
                monkeypatch.setattr(hr, "FUSED_RAYS", fused)
                with torch.no_grad():
                    outs.setdefault(fused, []).append(m.render(coords.cuda(), feats.cuda(), extr.cuda(), K.cuda(), res))
            a, b, a2 = outs[True][0], outs[False][0], outs[True][1]
            assert int(a["num_shading_points"]) == int(b["num_shading_points"]), tag
            if tag == "some miss":
                o_, d_, t0_, t1_ = hr.ray_gen(extr.flatten(0, 1).cuda(), K.flatten(0, 1).cuda(), res, 1.0)
                assert 0 < int((t1_ > t0_).sum()) and torch.isfinite(t0_).all()          # (the fix-up gave the missing rays finite limits)
                assert int(a["num_shading_points"]) > 0
            for key in ("channels", "mask", "depth"):
                assert torch.equal(a[key], b[key]), (tag, sync_free_points, key, float((a[key] - b[key]).abs().max()))
                assert torch.equal(a[key], a2[key]), (tag, key)
    # the oracle agrees with the opt-in path too
    ref = orr.render(p, coords, feats, extr, intr_some, res=res, S=S)
    monkeypatch.setattr(hr, "FUSED_RAYS", True)
    with torch.no_grad():
        out = m.render(coords.cuda(), feats.cuda(), extr.cuda(), intr_some.cuda(), res)
    assert float((out["channels"].cpu() - ref["channels"]).abs().max()) < 5e-3
    assert float((out["depth"].cpu() - ref["depth"]).abs().max()) < 1e-3
