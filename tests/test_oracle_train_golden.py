"""The training-path oracle (oracle/train_render.py) against fixtures generated from the reference
(tests/golden/make_golden_train.py): training-mode rendering incl. gradients, stage-1 losses, variational embedding."""
import numpy as np
import torch

from oracle import train_render as tr
from oracle.renderer import init_field_params

T = torch.from_numpy


def _proj(t, seed):
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(t.numel(), generator=g, dtype=torch.float64)
    t = t.detach().double().flatten()
    return np.array([float(t.norm()), float(t @ v)])


def _render(g, stable):
    p = {k: v.clone().requires_grad_(True) for k, v in init_field_params(32, seed=int(g["field_seed"])).items()}
    feats = T(g["feats"]).clone().requires_grad_(True)
    out = tr.render_train(p, T(g["coords"]), feats, T(g["extr"]), T(g["intr"]), int(g["res"]), int(g["S"]), int(g["M"]), int(g["k"]),
                          float(g["r"]), "brute", int(g["renderer_ray_subsamples"]), int(g["aggregator_ray_subsamples"]),
                          T(g["ray_perm"]), T(g["jitter"]), T(g["valid_perm"]), stable_regroup=stable)
    return p, feats, out


def test_train_render_forward_and_gradients(golden):
    g = golden("train_render")
    p, feats, out = _render(g, stable=False)          # the reference's own (unstable) regrouping: bit-for-bit the fixture
    assert out["num_rays"] == g["out_mask"].shape[2]
    assert (out["ray_idx"].numpy() == g["out_ray_idx"]).all()
    for key in ("mask", "depth", "channels"):
        np.testing.assert_allclose(out[key].detach().numpy(), g["out_" + key], atol=1e-6)
    loss = (out["channels"] * T(g["g_channels"])).sum() + (out["mask"] * T(g["g_mask"])).sum()
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-6)
    loss.backward()
    np.testing.assert_allclose(feats.grad.numpy(), g["d_feats"], atol=1e-8, rtol=1e-4)
    for i, (name, v) in enumerate(sorted(p.items())):
        ref, mine = g["dparam:" + name], _proj(v.grad, 1000 + i)
        np.testing.assert_allclose(mine, ref, atol=1e-5 * max(1.0, abs(ref[0])), err_msg=name)


def test_stable_regrouping_selects_the_same_number_of_valid_rays(golden):
    """the HIP build's specification (stable regrouping): same count, every selected ray has a valid slot, rays ascending"""
    g = golden("train_render")
    _, _, out = _render(g, stable=True)
    assert out["num_rays"] == g["out_mask"].shape[2]
    idx = out["ray_idx"][0, :, :, 0]
    pos = {int(r): i for i, r in enumerate(T(g["ray_perm"])[:int(g["renderer_ray_subsamples"])].tolist())}
    for inst in idx:
        order = [pos[int(r)] for r in inst]
        assert order == sorted(order)                 # ascending position in the subsampled ray list (boolean selection)


def test_losses(golden):
    g = golden("losses")
    kl, pw = tr.kl_loss(T(g["kl_mean"]), T(g["kl_log_var"]), float(g["kl_weight"]))
    np.testing.assert_allclose(float(kl), float(g["kl_total"]), rtol=1e-6)
    np.testing.assert_allclose(pw.numpy(), g["kl_pointwise"], rtol=1e-6)
    np.testing.assert_allclose(float(tr.image_loss(T(g["img"]), T(g["pred_channels"]), T(g["ray_idx"]))), float(g["rec_total"]), rtol=1e-6)
    np.testing.assert_allclose(float(tr.image_loss(T(g["img"]), T(g["pred_full"]))), float(g["rec_full"]), rtol=1e-6)
    tv, pw = tr.tv_loss(T(g["tv_coords"]), T(g["tv_feats"]), int(g["tv_k"]), float(g["tv_r"]), float(g["tv_weight"]))
    np.testing.assert_allclose(float(tv), float(g["tv_total"]), rtol=1e-5)
    np.testing.assert_allclose(pw.numpy(), g["tv_pointwise"], rtol=1e-5, atol=1e-4)


def test_variational_embedding(golden):
    g = golden("variational_embedding")
    out = tr.variational_embedding(T(g["table"]), T(g["idx"]), 12, 8, T(g["eps"]))
    np.testing.assert_allclose(out.numpy(), g["out_train"], atol=1e-7)
    np.testing.assert_allclose(tr.variational_embedding(T(g["table"]), T(g["idx"]), 12, 8).numpy(), g["out_eval"], atol=0)
    np.testing.assert_allclose(g["mean"], g["out_eval"])
    np.testing.assert_allclose(np.exp(0.5 * g["log_var"]), g["std"], rtol=1e-6)
