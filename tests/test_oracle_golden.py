"""Pin the CPU oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import denoiser as od
from oracle import diffusion as odf
from oracle import renderer as orr
from oracle import voxel_grid as ovg

T = torch.from_numpy


def close(a, b, atol, rtol=0.0):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


# ------------------------------------------------------------------ denoiser ----------------
@pytest.mark.parametrize("tag", ["n513_h1_d64", "n130_h4_d64", "n17_h4_d32"])
def test_attention_matches_reference(golden, tag):
    g = golden("attention_" + tag)
    qkv = T(g["qkv"]).requires_grad_(True)
    out = od.attention_qkvpacked(qkv, int(g["heads"]))
    (out * T(g["gout"])).sum().backward()
    close(out, g["out"], 2e-6)
    close(qkv.grad, g["dqkv"], 5e-6)


def test_timestep_embedding(golden):
    g = golden("timestep_embedding")
    t = T(g["t"])
    for dim in (128, 1024, 7):
        close(od.timestep_embedding(t, dim), g[f"dim{dim}"], 1e-6)


@pytest.mark.parametrize("tag", ["f32_w64", "f128_w64", "f32_w128_h2"])
def test_denoiser_forward_backward(golden, tag):
    g = golden("denoiser_" + tag)
    p = {k[2:]: T(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith("w:")}
    ec, ef = od.denoiser_forward(p, T(g["coords"]), T(g["feats"]), T(g["t"]), int(g["heads"]))
    close(ec, g["eps_coords"], 2e-5)
    close(ef, g["eps_feats"], 2e-5)
    ((ec * T(g["gc"])).sum() + (ef * T(g["gf"])).sum()).backward()
    for k, v in g.items():
        if k.startswith("g:"):
            close(p[k[2:]].grad, v, 2e-4 * max(1.0, float(np.abs(v).max())))


def test_init_params_keys_match_reference_state_dict(golden):
    g = golden("denoiser_f32_w64")
    ref_keys = {k[2:]: v.shape for k, v in g.items() if k.startswith("w:")}
    mine = od.init_params(3, 32, 64, 2, 1)
    assert {k: tuple(v.shape) for k, v in mine.items()} == {k: tuple(s) for k, s in ref_keys.items()}


# ------------------------------------------------------------------ diffusion ---------------
def test_schedule_tables_bitexact(golden):
    g = golden("diffusion")
    tab = odf.schedule_tables()
    for k, v in g.items():
        if k.startswith("tab:"):
            np.testing.assert_array_equal(tab[k[4:]].numpy(), v, err_msg=k)


def _fake_denoiser(g):
    wc, wf = T(g["wc"]), T(g["wf"])

    def fn(c, f, tt):
        s = (tt.float() / 1000.0).reshape(-1, 1, 1)
        return torch.einsum("ij,bjn->bin", wc, c) + s, torch.tanh(torch.einsum("ij,bjn->bin", wf, f)) - s
    return fn


def test_q_sample_and_p_losses(golden):
    g = golden("diffusion")
    tab = odf.schedule_tables()
    t = T(g["t"])
    close(odf.q_sample(tab, T(g["c0"]), t, T(g["cn"])), g["coords_t"], 0)
    close(odf.q_sample(tab, T(g["f0"]), t, T(g["fn"])), g["feats_t"], 0)
    loss, sub, pw = odf.p_losses(tab, _fake_denoiser(g), T(g["c0"]), T(g["f0"]), t, T(g["cn"]), T(g["fn"]))
    close(loss, g["loss"], 1e-7)
    close(sub["00_coords_loss"], g["coords_loss"], 1e-7)
    close(sub["01_feats_loss"], g["feats_loss"], 1e-7)
    close(pw["pointwise_coords_loss"], g["pw_coords"], 1e-6)
    close(pw["pointwise_feats_loss"], g["pw_feats"], 1e-6)


def test_reverse_step(golden):
    g = golden("diffusion")
    tab = odf.schedule_tables()
    t = T(g["t"])
    nc, rc = odf.p_sample_step(tab, T(g["c0"]), T(g["ps_eps_c"]), t, T(g["ps_noise_c"]),
                               (T(g["ps_clipc"])[0], T(g["ps_clipc"])[1]))
    nf, rf = odf.p_sample_step(tab, T(g["f0"]), T(g["ps_eps_f"]), t, T(g["ps_noise_f"]),
                               (T(g["ps_clipf"])[0], T(g["ps_clipf"])[1]))
    close(rc, g["ps_rec_c"], 1e-6); close(rf, g["ps_rec_f"], 1e-6)
    close(nc, g["ps_next_c"], 1e-6); close(nf, g["ps_next_f"], 1e-6)


def test_normalizers(golden):
    g = golden("normalizers")
    su = odf.unit_gaussian_stats(T(g["data_c"]))
    sm = odf.minus_one_to_one_stats(T(g["data_f"]))
    for k in ("shift", "scale", "min", "max"):
        close(su[k], g["un:" + k], 1e-6)
        close(sm[k], g["mm:" + k], 1e-6)
    close(odf.normalize(su, T(g["xc"]), True), g["c_train"], 1e-6)
    close(odf.normalize(sm, T(g["xf"]), True), g["f_train"], 1e-6)
    close(odf.normalize(su, T(g["xc"]), False), g["c_eval"], 1e-6)
    close(odf.normalize(sm, T(g["xf"]), False), g["f_eval"], 1e-6)


# ------------------------------------------------------------------ rays --------------------
def test_camera_rays(golden):
    g = golden("rays")
    extr, intr = T(g["extr"]), T(g["intr"])
    for res in (8, 128):
        o, d = orr.camera_rays(extr, intr, res)
        st = int(g[f"stride{res}"])
        close(o[:, ::st], g[f"o{res}"], 1e-6)
        close(d[:, ::st], g[f"d{res}"], 1e-6)
    o, d = orr.camera_rays(extr, T(g["intr_skew"]), 8)
    close(o, g["o8_skew"], 1e-6); close(d, g["d8_skew"], 1e-6)


def test_ray_limits_and_depth_samples(golden):
    g = golden("rays")
    extr, intr = T(g["extr"]), T(g["intr"])
    o, d = orr.camera_rays(extr, intr, 8)
    s, e = orr.ray_box_limits(o[None], d[None])
    close(s, g["lim_start"], 1e-6); close(e, g["lim_end"], 1e-6)
    # partial miss: substituted by global min/max of the hitting rays
    s2, e2 = orr.ray_box_limits(T(g["ow"])[None], T(g["dw"])[None])
    close(s2, g["limw_start"], 1e-6); close(e2, g["limw_end"], 1e-6)
    assert (g["limw_start"] == g["limw_start"].min()).sum() > 1
    # all rays miss: (-1, -2) kept
    s3, e3 = orr.ray_box_limits(T(g["om"])[None], T(g["dm"])[None])
    close(s3, g["limm_start"], 0); close(e3, g["limm_end"], 0)
    dep = orr.depth_samples(T(g["lim_start"]), T(g["lim_end"]), 16)
    close(dep.reshape(g["depths16"].shape), g["depths16"], 0)


# ------------------------------------------------------------------ renderer ----------------
def _field(g):
    p = orr.init_field_params(32, seed=int(g["field_seed"]))
    chk = float(sum(v.double().abs().sum() for v in p.values()))
    assert abs(chk - g["field_checksum"][0]) < 1e-6 * chk, "CPU RNG drifted: regenerate tests/golden"
    assert float(p["shape_net.0.weight"][17, 5]) == pytest.approx(g["field_checksum"][1], abs=0)
    return p


def test_brute_force_query_matches_reference(golden):
    g = golden("render_brute")
    x = g["x"]                                            # [1,2,R,S,3]
    B, Tn, R, S, _ = x.shape
    M, k, r = int(g["M"]), int(g["k"]), float(g["r"])
    idx, loc, nvalid = ovg.brute_force_query(x.reshape(B, Tn * R, S, 3), g["coords"], k, r, M)
    mask = np.arange(M)[None, None, :] < nvalid[..., None]
    ref_mask = g["mask"].reshape(B, Tn * R, M)
    # samples whose nearest-point distance is within 1e-6 of the radius may legitimately flip
    border = (np.abs(g["dist64_min"].reshape(B, Tn * R, S) - r) < 1e-6).any(axis=-1)
    assert border.sum() <= 2
    np.testing.assert_array_equal(mask[~border], ref_mask[~border])
    ok_rays = ~border.reshape(-1)
    mine_sets = np.sort(idx.reshape(-1, M, k)[ok_rays][mask.reshape(-1, M)[ok_rays]], axis=-1)
    ref_sets = np.sort(g["nb_idx"], axis=-1)
    if border.any():
        pytest.skip("radius-boundary sample present; set comparison needs per-ray alignment")
    assert mine_sets.shape == ref_sets.shape
    # the reference ranks by a matmul-based cdist (noisy), so the k-th neighbour may swap with the
    # (k+1)-th when they are nearly equidistant; require >= 99.9 % identical sets and no stranger
    same = (mine_sets == ref_sets).all(axis=-1)
    assert same.mean() > 0.999
    np.testing.assert_allclose(loc.reshape(-1, M, 3)[mask.reshape(-1, M)], g["shading_pts"], atol=0)


def test_shading_matches_reference(golden):
    g = golden("render_brute")
    p = _field(g)
    sigma, rgb, feat = orr.shade_points(p, T(g["nb_idx"]), T(g["shading_pts"]), T(g["coords"]), T(g["feats"]))
    close(feat, g["agg_feat"], 2e-5)
    close(sigma, g["sigma"], 2e-5)
    close(rgb, g["rgb"], 2e-5)


def test_depths_and_ray_march(golden):
    g = golden("raymarch")
    Nr, M = g["mask"].shape[2:4]
    mask = T(g["mask"]).reshape(Nr, M)
    dep = orr.depths_from_points(T(g["pts"]).reshape(Nr, M, 3), mask, T(g["o"]).reshape(Nr, 3),
                                 T(g["d"]).reshape(Nr, 3), T(g["ray_end"]).reshape(Nr, 1))
    close(dep, g["depths"].reshape(Nr, M), 1e-6)
    rgb = torch.zeros(Nr, M, 3)
    rgb[mask] = T(g["rgb_compact"])
    total, depth, chan = orr.ray_march(T(g["sigma"]).reshape(Nr, M), dep, rgb, mask)
    close(total, g["out_mask"].reshape(Nr, 1), 1e-6)
    close(depth, g["out_depth"].reshape(Nr, 1), 1e-5)
    close(chan, g["out_channels"].reshape(Nr, 3), 1e-6)
    assert np.isfinite(g["out_depth"]).all()


def test_render_end_to_end_brute(golden):
    g = golden("render_brute")
    p = _field(g)
    out = orr.render(p, T(g["coords"]), T(g["feats"]), T(g["extr"]), T(g["intr"]), res=int(g["res"]),
                     S=int(g["S"]), M=int(g["M"]), k=int(g["k"]), r=float(g["r"]), mode="brute")
    close(out["mask"], g["out_mask"], 5e-5)
    close(out["channels"], g["out_channels"], 5e-5)
    close(out["depth"], g["out_depth"], 5e-5)
    assert float(g["out_mask"].max()) > 0.1           # the object is actually visible (random weights: low density)


def test_unflatten_and_psnr(golden):
    g = golden("unflatten")
    close(orr.unflatten_image(T(g["channels"])), g["image"], 0)
    a = torch.rand(3, 8, 8)
    assert orr.psnr(a, a) == float("inf")
    assert orr.psnr(torch.zeros(4), torch.full((4,), 0.1)) == pytest.approx(20.0, abs=1e-4)


# ------------------------------------------------------------------ voxel grid spec ---------
@pytest.mark.parametrize("level", ["fine", "scaled"])
def test_voxel_grid_spec_properties(level):
    """The grid branch is parity-unpinned upstream; check the spec's own invariants and its
    relation to the exact radius query (grid result is a subset of the in-radius set), for both readings of the grid."""
    coords, _ = orr.synthetic_cloud(256, 4, seed=2)
    grid = ovg.VoxelGridOracle(grid_level=level)
    assert tuple(grid.cdims) == (25, 25, 25) and tuple(grid.dims) == ((50, 50, 50) if level == "fine" else (25, 25, 25))
    assert max(grid.vsize_tup) * 2 == pytest.approx(0.08)          # the radius always comes from the UNSCALED edge
    assert float(grid.vsize[0]) == pytest.approx(0.04 if level == "fine" else 0.08)
    grid.set_pointset(coords.numpy(), np.array([256], dtype=np.int32))
    rng = np.random.default_rng(0)
    base = coords.numpy()[0][rng.integers(0, 256, size=(40, 1))]             # near the surface
    x = (base + rng.normal(0, 0.03, size=(40, 24, 3))).astype(np.float32)[None]
    idx, loc, nsel, sel = grid.query_dense(x, 8, 2.0, 10)
    bidx, bloc, bn = ovg.brute_force_query(x, coords.numpy(), 256, 0.08, 24)
    assert (nsel <= 10).all() and nsel.max() == 10
    for r in range(40):
        for m in range(int(nsel[0, r])):
            s = sel[0, r, m]
            got = idx[0, r, m][idx[0, r, m] >= 0]
            d = np.linalg.norm(coords.numpy()[0][got] - x[0, r, s], axis=-1)
            assert (d < 0.08).all() and (np.diff(d) >= -1e-7).all()
            inball = np.nonzero(np.linalg.norm(coords.numpy()[0] - x[0, r, s], axis=-1) < 0.08)[0]
            assert set(got.tolist()) <= set(inball.tolist())
    # compaction contract of VoxelGrid.query
    sidx, sloc, ray_mask = grid.query(x, 8, 2.0, 10)
    assert sidx.shape == (int(ray_mask.sum()), 10, 8) and sloc.shape == (int(ray_mask.sum()), 10, 3)


def test_scaled_grid_with_uncapped_lists_is_the_reference_brute_force_branch():
    """Radius 0.08 = one cell of the scaled grid, so the 3^3 window of scaled cells CONTAINS the radius ball: with lists that
    drop nothing (max_points_per_voxel large) the scaled reading returns, for every sample that has a neighbour, exactly the
    list of the reference's pinned brute-force branch (aggregator.py:42-58) -- same indices in the same (dist^2, index) order.
    The fine reading cannot: its 3^3 window of 0.04 voxels does not cover the ball."""
    coords, _ = orr.synthetic_cloud(512, 4, seed=0)
    rng = np.random.default_rng(1)
    base = coords.numpy()[0][rng.integers(0, 512, size=(64, 1))]
    x = (base + rng.normal(0, 0.05, size=(64, 16, 3))).astype(np.float32)[None]
    bidx, _, _ = ovg.brute_force_query(x.reshape(1, 64 * 16, 1, 3), coords.numpy(), 8, 0.08, 1)      # one slot per sample
    bidx = bidx.reshape(64, 16, 8)
    rates = {}
    for level in ("fine", "scaled"):
        g = ovg.VoxelGridOracle(max_points_per_voxel=64, grid_level=level)
        g.set_pointset(coords.numpy(), np.array([512], dtype=np.int32))
        assert g.kept.all()
        idx, _, nsel, sel = g.query_dense(x, 8, 2.0, 16)
        same = total = 0
        for r in range(64):
            for m in range(int(nsel[0, r])):
                sm = sel[0, r, m]
                if bidx[r, sm, 0] < 0:
                    assert (idx[0, r, m] < 0).all()        # a selected sample without any in-radius point stays neighbour-less
                    continue
                total += 1
                same += int(np.array_equal(idx[0, r, m], bidx[r, sm]))
        rates[level] = same / total
        assert total > 300
    assert rates["scaled"] == 1.0 and rates["fine"] < 0.9, rates


def test_self_query_of_the_tv_loss_loses_keypoints_only_on_the_scaled_grid():
    """neural_point_cloud_tv_loss.py:41-43 queries every keypoint's own position and notes 'VoxelGrid looses keypoints sometimes'
    (a point comes back without any neighbour).  On the FINE grid that cannot happen for an in-range point: a kept point finds
    itself (distance 0), and a point dropped by the 4-per-voxel cap shares its 0.04 voxel (diagonal 0.069 < radius 0.08) with four
    kept ones.  On the SCALED grid (0.08 cells, diagonal 0.139) a dropped point can be farther than the radius from every kept
    point: the comment describes that reading.  Both are checked on the bench cloud; the scaled loss is forced on a built case."""
    coords, _ = orr.synthetic_cloud(512, 4, seed=0)
    c = coords.numpy()
    lost = {}
    for level in ("fine", "scaled"):
        g = ovg.VoxelGridOracle(grid_level=level)
        g.set_pointset(c, np.array([512], dtype=np.int32))
        idx, _, nsel, _ = g.query_dense(c.reshape(1, 512, 1, 3), 8, 2.0, 50)
        assert (nsel == 1).all()                            # every keypoint's own cell is occupied
        lost[level] = int((idx[0, :, 0, 0] < 0).sum())
        dropped = int((~g.kept).sum())
        print(f"[tv self-query, {level}] keypoints dropped by the cap {dropped} / 512, keypoints without any neighbour {lost[level]}")
    assert lost["fine"] == 0
    # built case: six points in one scaled cell, the last two > 0.08 away from the first four and from everything else
    # (cell edges sit at -1 + 0.08 i: [0.04, 0.12) is one scaled cell = eight fine voxels)
    pts = np.array([[[0.041, 0.041, 0.041], [0.042, 0.041, 0.041], [0.041, 0.042, 0.041], [0.041, 0.041, 0.042],
                     [0.119, 0.119, 0.119], [0.118, 0.119, 0.119]]], dtype=np.float32)
    g = ovg.VoxelGridOracle(grid_level="scaled")
    g.set_pointset(pts, np.array([6], dtype=np.int32))
    assert g.kept[0].tolist() == [True] * 4 + [False] * 2
    idx, _, _, _ = g.query_dense(pts.reshape(1, 6, 1, 3), 8, 2.0, 50)
    assert (idx[0, :4, 0, 0] >= 0).all() and (idx[0, 4:, 0] < 0).all()          # the two dropped keypoints are lost
    gf = ovg.VoxelGridOracle(grid_level="fine")
    gf.set_pointset(pts, np.array([6], dtype=np.int32))
    assert gf.kept.all() and (gf.query_dense(pts.reshape(1, 6, 1, 3), 8, 2.0, 50)[0][0, :, 0, 0] >= 0).all()


@pytest.mark.parametrize("level", ["fine", "scaled"])
def test_voxel_grid_capacity_limits(level):
    pts = np.zeros((1, 12, 3), dtype=np.float32)
    pts[0, :, 0] = 0.01                                                  # 12 points in ONE fine voxel
    pts[0, :, 1] = np.linspace(0.001, 0.03, 12)
    g = ovg.VoxelGridOracle(max_points_per_voxel=4, grid_level=level)
    g.set_pointset(pts, np.array([12], dtype=np.int32))
    assert g.kept[0].tolist() == [True] * 4 + [False] * 8
    x = np.array([[[[0.01, 0.01, 0.0]]]], dtype=np.float32)
    idx, _, nsel, _ = g.query_dense(x, 8, 2.0, 5)
    assert nsel[0, 0] == 1 and set(idx[0, 0, 0][idx[0, 0, 0] >= 0].tolist()) <= {0, 1, 2, 3}
    # voxel cap: keep only the 2 occupied voxels with the smallest linear id
    pts2 = np.array([[[0.5, 0.5, 0.5], [-0.5, -0.5, -0.5], [0.0, 0.0, 0.0], [1.5, 0, 0]]], dtype=np.float32)
    g2 = ovg.VoxelGridOracle(max_occ_voxels_per_example=2, grid_level=level)
    g2.set_pointset(pts2, np.array([4], dtype=np.int32))
    assert g2.kept[0].tolist() == [False, True, True, False]            # last point is out of range
