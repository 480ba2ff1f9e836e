"""Train-state checkpoints against a checkpoint written by the reference's own `save_all`
(utils/checkpoint_utils.py:196-236; fixture `train_state_format.npz` from tests/golden/make_golden_train.py: tiny NPCD, two
AdamW + EMA steps on recorded gradients, then a third step by the reference):
  * a reference checkpoint loads into this build's trainer (flat buffers) and is written back unchanged -- top-level keys,
    optimizer state / param_groups, scheduler and EMA-scheduler dictionaries, EMA model keys, file name;
  * continuing from it reproduces the reference's third step (torch AdamW on the CPU path, the fused HIP AdamW + EMA
    kernel on the GPU path);
  * save -> resume in a fresh trainer continues bit for bit."""
import json
import os

import numpy as np
import pytest
import torch


def _cfg(g):
    cfg = {k: v for k, v in g["cfg"]}
    return dict(n_obj=int(cfg["n_obj"]), coords_dim=int(cfg["coords_dim"]), feats_dim=int(cfg["feats_dim"]), num_points=int(cfg["num_points"]),
                use_view_dir=cfg["use_view_dir"] == "True", width=int(cfg["width"]), layers=int(cfg["layers"]), heads=int(cfg["heads"]))


def _scalar(a):
    a = np.asarray(a)
    if a.dtype.kind in "US":
        return None if str(a) == "None" else str(a)
    return a.tolist()


def _reference_checkpoint(g, net):
    """the dictionary the reference's save_all wrote, rebuilt from the fixture"""
    def sub(path):
        return [str(k) for k in g["dictkeys:" + path]]
    ema_key = str(g["ema_key"])
    ckpt = {}
    for top in g["top_level_keys"]:
        top = str(top)
        if top == "model_state_dict":
            ckpt[top] = {k: torch.from_numpy(g[f"ckpt/{top}/{k}"]) for k in sub("ckpt/" + top)}
        elif top == "optimizer_state_dict":
            state = {int(i): {k: torch.from_numpy(np.asarray(g[f"ckpt/{top}/state/{i}/{k}"])) for k in sub(f"ckpt/{top}/state/{i}")}
                     for i in sub(f"ckpt/{top}/state")}
            groups = json.loads(str(g[f"json:ckpt/{top}/param_groups"]))
            for gr in groups:
                gr["betas"] = tuple(gr["betas"])
            ckpt[top] = {"state": state, "param_groups": groups}
        elif top == ema_key:
            own = net.state_dict()
            ckpt[top] = {str(k): (torch.from_numpy(g["ema_model/" + str(k)]) if "ema_model/" + str(k) in g else own[str(k)])
                         for k in g["ema_model_keys"]}
        else:
            ckpt[top] = {k: _scalar(g[f"ckpt/{top}/{k}"]) for k in sub("ckpt/" + top)}
    return ckpt


def _check_written_back(out, ref, g):
    assert list(out.keys()) == [str(k) for k in g["top_level_keys"]]
    assert list(out["model_state_dict"].keys()) == list(ref["model_state_dict"].keys())
    for k, v in ref["model_state_dict"].items():
        assert torch.equal(out["model_state_dict"][k].cpu(), v), k
    oo, ro = out["optimizer_state_dict"], ref["optimizer_state_dict"]
    assert json.dumps(oo["param_groups"]) == json.dumps(ro["param_groups"])
    assert list(oo["state"].keys()) == list(ro["state"].keys())
    for i, st in ro["state"].items():
        assert list(oo["state"][i].keys()) == list(st.keys())
        for k, v in st.items():
            assert oo["state"][i][k].dtype == v.dtype and torch.equal(oo["state"][i][k].cpu(), v), (i, k)
    assert oo["state"][0]["step"].device.type == "cpu"
    assert out["scheduler_state_dict"] == ref["scheduler_state_dict"]
    ema_key = str(g["ema_key"])
    sk = ema_key.replace("_model_state_dict", "_scheduler_state_dict")
    assert out[sk] == ref[sk]
    assert list(out[ema_key].keys()) == [str(k) for k in g["ema_model_keys"]]
    for k in out[ema_key]:
        if k.startswith("diffusion."):
            assert torch.equal(out[ema_key][k].cpu(), ref[ema_key][k]), k


def _third_step(tr, g, atol):
    names = [str(n) for n in g["param_names"]]
    params = dict(tr.model.named_parameters())
    with torch.no_grad():
        for i, n in enumerate(names):
            params[n].grad.copy_(torch.from_numpy(g[f"grad3/{i}"]))
    tr.apply_gradients()
    assert tr.iteration == 3
    ema = tr.ema_state_dict()
    for n in names:
        np.testing.assert_allclose(params[n].detach().cpu().numpy(), g["after3/" + n], rtol=0, atol=atol, err_msg=n)
        np.testing.assert_allclose(ema[n].cpu().numpy(), g["ema_after3/" + n], rtol=0, atol=atol, err_msg=n)


def _run(device, fused, tmp_path, golden, atol):
    from npcd.models import NPCD
    from npcd.train import DiffusionTrainer, resume_latest, save_train_state
    g = golden("train_state_format")
    net = NPCD(**_cfg(g)).to(device)
    ref = _reference_checkpoint(g, net.cpu())
    net.to(device)
    tr = DiffusionTrainer(net.diffusion, lr=1.0, weight_decay=0.5, dtype=None, fused=fused)
    tr.load_state_dict(ref)
    assert (tr.iteration, tr.finished_iterations, tr.lr, tr.weight_decay) == (2, 2, 7e-5, 0.01)
    _check_written_back(tr.state_dict(full_model=net), ref, g)
    # file naming, pruning and resume
    path = save_train_state(tr, str(tmp_path), full_model=net)
    assert os.path.basename(path) == str(g["file_name"])
    net2 = NPCD(**_cfg(g)).to(device)
    tr2 = DiffusionTrainer(net2.diffusion, dtype=None, fused=fused)
    assert resume_latest(tr2, str(tmp_path)) == path
    assert torch.equal(tr2.flat.flat, tr.flat.flat) and torch.equal(tr2.ema, tr.ema) and tr2.iteration == 2
    # the reference's third step, from both trainers
    _third_step(tr, g, atol)
    _third_step(tr2, g, atol)
    assert torch.equal(tr2.flat.flat, tr.flat.flat) and torch.equal(tr2.ema, tr.ema)
    tr.finished_iterations = 3
    save_train_state(tr, str(tmp_path), max_to_keep=1)
    assert sorted(os.listdir(tmp_path)) == ["diffusion_training-iter-000000003.pt"]
    return tr


def test_reference_train_state_roundtrip_cpu(tmp_path, golden):
    _run("cpu", False, tmp_path, golden, atol=1e-7)


@pytest.mark.gpu
def test_reference_train_state_roundtrip_gpu(tmp_path, golden):
    """flat buffers + fused HIP AdamW/EMA kernel continue a reference checkpoint like torch.optim.AdamW + EMAHandler do"""
    tr = _run("cuda", True, tmp_path, golden, atol=2e-7)
    assert tr.native


@pytest.mark.gpu
def test_resume_continues_bit_for_bit_gpu(tmp_path):
    """train 2 steps, checkpoint, 2 more -- vs a fresh trainer resumed from the checkpoint running the same 2 steps"""
    from npcd.models.diffusion import DiffusionModel
    from npcd.train import DiffusionTrainer, resume_latest, save_train_state
    def make():
        torch.manual_seed(0)
        m = DiffusionModel(3, 32, 48, 128, 2, 2, True).cuda()
        with torch.no_grad():
            m.denoiser.output_proj.weight.normal_(0, 0.05)
        return m
    g = torch.Generator().manual_seed(5)
    batches = [(torch.randn(4, 3, 48, generator=g).cuda(), torch.randn(4, 32, 48, generator=g).cuda(), torch.randint(0, 1000, (4,), generator=g).cuda(),
                torch.randn(4, 3, 48, generator=g).cuda(), torch.randn(4, 32, 48, generator=g).cuda()) for _ in range(4)]
    a = DiffusionTrainer(make())
    for c, f, t, cn, fn in batches[:2]:
        a.step(c, f, t=t, coords_noise=cn, feats_noise=fn)
    save_train_state(a, str(tmp_path))
    for c, f, t, cn, fn in batches[2:]:
        la, _ = a.step(c, f, t=t, coords_noise=cn, feats_noise=fn)
    b = DiffusionTrainer(make())
    with torch.no_grad():
        for p in b.model.parameters():
            p.add_(1.0)                           # whatever it held is replaced by the checkpoint
    assert resume_latest(b, str(tmp_path)) is not None and b.finished_iterations == 2
    for c, f, t, cn, fn in batches[2:]:
        lb, _ = b.step(c, f, t=t, coords_noise=cn, feats_noise=fn)
    assert float(la) == float(lb)
    assert torch.equal(a.flat.flat, b.flat.flat) and torch.equal(a.ema, b.ema) and torch.equal(a.exp_avg_sq, b.exp_avg_sq)


def test_stage1_train_state_roundtrip_cpu(tmp_path, golden):
    """PointNeRFTrainer: {model, optimizer, scheduler}_state_dict over the whole model / Adam on every pointnerf parameter
    (pointnerf_training.py:102,180-187), `pointnerf_training-iter-%09d.pt`, resume restores optimizer moments and iteration."""
    from npcd.models import NPCD
    from npcd.train import PointNeRFTrainer
    cfg = _cfg(golden("train_state_format"))
    torch.manual_seed(0)
    net = NPCD(**cfg)
    tr = PointNeRFTrainer(net)
    all_params = list(net.pointnerf.parameters())
    assert tr.optimizer.state_dict()["param_groups"][0]["params"] == list(range(len(all_params)))
    g = torch.Generator().manual_seed(1)
    for _ in range(2):                                       # two Adam steps on made-up gradients (no renderer on the CPU)
        for p in all_params:
            if p.requires_grad:
                p.grad = torch.randn(p.shape, generator=g) * 0.01
        tr.optimizer.step(); tr.scheduler.step(); tr.iteration += 1
    path = tr.save(str(tmp_path))
    assert os.path.basename(path) == "pointnerf_training-iter-000000002.pt"
    ckpt = torch.load(path, weights_only=False)
    assert list(ckpt.keys()) == ["model_state_dict", "optimizer_state_dict", "scheduler_state_dict"]
    assert list(ckpt["model_state_dict"].keys()) == list(net.state_dict().keys())
    torch.manual_seed(1)
    net2 = NPCD(**cfg)
    tr2 = PointNeRFTrainer(net2)
    assert tr2.resume_latest(str(tmp_path)) == path and tr2.iteration == 2
    for (ka, va), (kb, vb) in zip(net.state_dict().items(), net2.state_dict().items()):
        assert ka == kb
        if torch.is_tensor(va):
            assert torch.equal(va, vb), ka
        else:
            assert torch.equal(va["emb"]["weight"], vb["emb"]["weight"]), ka
    sa, sb = tr.optimizer.state_dict()["state"], tr2.optimizer.state_dict()["state"]
    assert list(sa.keys()) == list(sb.keys()) and len(sa) > 0
    for i in sa:
        assert torch.equal(sa[i]["exp_avg"], sb[i]["exp_avg"]) and float(sa[i]["step"]) == float(sb[i]["step"]) == 2.0
