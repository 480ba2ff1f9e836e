"""Checkpoint format of the whole model against a fixture produced by the reference's own NPCD
(tests/golden/make_golden_train.py): state_dict keys in order, shapes, the nested `_extra_state` of the two embedding
tables (flex_embedding.py:9-26, embedding.py:53-60), the duplicated field keys (pointnerf.field.* and
pointnerf.renderer.field.*), and loading such a checkpoint into this build's model."""
import numpy as np
import torch


def _cfg(g):
    cfg = {k: v for k, v in g["cfg"]}
    return dict(n_obj=int(cfg["n_obj"]), coords_dim=int(cfg["coords_dim"]), feats_dim=int(cfg["feats_dim"]), num_points=int(cfg["num_points"]),
                use_view_dir=cfg["use_view_dir"] == "True", width=int(cfg["width"]), layers=int(cfg["layers"]), heads=int(cfg["heads"]))


def test_state_dict_layout_matches_reference(golden):
    from npcd.models import NPCD
    g = golden("checkpoint_format")
    net = NPCD(**_cfg(g))
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    for k, v in sd.items():
        if torch.is_tensor(v):
            assert tuple(v.shape) == tuple(g["shape:" + k]), k
        else:
            assert set(v) == {"emb"} and set(v["emb"]) == {"weight"}, k
            assert tuple(v["emb"]["weight"].shape) == g["extra:" + k].shape, k
    dup = [k for k in sd if k.startswith("pointnerf.renderer.field.")]
    assert dup and all(k.replace("pointnerf.renderer.field.", "pointnerf.field.") in sd for k in dup)


def test_reference_format_checkpoint_loads_and_evaluates(golden):
    """A checkpoint with the reference's layout (values reconstructed from the fixture's tables + a deterministic fill) loads
    strictly, and the embeddings return what the reference returned for the same tables."""
    from npcd.models import NPCD
    g = golden("checkpoint_format")
    net = NPCD(**_cfg(g))
    gen = torch.Generator().manual_seed(123)
    ckpt = {}
    with torch.no_grad():
        # same fill order as the fixture: parameters() order, then the two tables
        ref_vals = {}
        for name, p in net.named_parameters():
            ref_vals[name] = torch.randn(p.shape, generator=gen) * 0.1
    for k in g["keys"]:
        k = str(k)
        if "extra:" + k in g:
            ckpt[k] = {"emb": {"weight": torch.nn.Parameter(torch.from_numpy(g["extra:" + k]).clone())}}
        elif k in ref_vals:
            ckpt[k] = ref_vals[k]
        elif k.replace("pointnerf.renderer.field.", "pointnerf.field.") in ref_vals:
            ckpt[k] = ref_vals[k.replace("pointnerf.renderer.field.", "pointnerf.field.")]
        else:
            ckpt[k] = net.state_dict()[k]                        # buffers (normaliser statistics)
    missing, unexpected = net.load_state_dict(ckpt, strict=True)
    assert not missing and not unexpected
    net.eval()
    idx = torch.from_numpy(g["idx"])
    np.testing.assert_allclose(net.pointnerf.feats(idx).detach().numpy(), g["feats_out"], atol=1e-7)
    np.testing.assert_allclose(net.pointnerf.coords(idx).detach().numpy(), g["coords_out"], atol=1e-7)
    assert net.pointnerf.get_all_coords().shape == (3, 16, 3)
    # a round trip through this build's own state_dict is loss-free, including the nested embedding state
    again = NPCD(**_cfg(g))
    again.load_state_dict(net.state_dict(), strict=True)
    for (ka, va), (kb, vb) in zip(net.state_dict().items(), again.state_dict().items()):
        assert ka == kb
        if torch.is_tensor(va):
            assert torch.equal(va, vb), ka
        else:
            assert torch.equal(va["emb"]["weight"], vb["emb"]["weight"]), ka
