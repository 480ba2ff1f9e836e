"""CPU-only checks of the host side: C-ABI library loads and exports every declared symbol,
module trees / state_dict keys mirror the reference, host math equals the oracle, and the product
path refuses to run without a GPU (no silent fallback)."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT


def test_library_exports_every_declared_symbol():
    from npcd import hip
    header = open(os.path.join(ROOT, "include", "npcd_hip.h")).read()
    declared = set(re.findall(r"\b(npcd_[a-z0-9_]+)\s*\(", header))
    declared -= {"npcd_grid_params"}
    assert declared, "no declarations parsed"
    L = hip.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"libnpcd_hip.so does not export {name}"
    assert declared == set(hip.SIGNATURES), (declared ^ set(hip.SIGNATURES))
    assert L.npcd_missing == (), f"stale library, missing {L.npcd_missing}"
    assert L.npcd_abi_version() == 9
    assert L.npcd_ray_march_ws_floats(16384) >= 2 and L.npcd_ray_march_ws_floats(0) == -1
    assert L.npcd_ray_gen_ws_floats(1, 128, 0) == 4 + 2 * 64 and L.npcd_ray_gen_ws_floats(2, 128, 100) == 4 + 2 and L.npcd_ray_gen_ws_floats(0, 128, 0) == -1
    assert L.npcd_error_string(-2).decode() == "unsupported shape or dtype"


def test_attention_backward_workspace_size():
    """Host-side sizing of the backward's scratch (include/npcd_hip.h): two row-constant planes over n padded to 64, plus 192
    floats per (batch, head, 32-row block) when the sequence is 128 j + 1 tokens long (the edge token's partial sums)."""
    from npcd import hip
    f = hip.lib().npcd_attn_bwd_workspace_floats
    assert f(2, 512, 3) == 2 * 2 * 3 * 512
    assert f(2, 513, 3) == 2 * 2 * 3 * 576 + 2 * 3 * 4 * 4 * 192
    assert f(2, 129, 3) == 2 * 2 * 3 * 192 + 2 * 3 * 1 * 4 * 192
    assert f(2, 65, 3) == 2 * 2 * 3 * 128 and f(2, 1, 3) == 2 * 2 * 3 * 64          # seeded / too short: no edge scratch
    assert f(0, 513, 3) == -1
    g = hip.lib().npcd_attn_fwd_fp8_workspace_bytes            # e4m3 k and v^T of the opt-in fp8 forward; 0 = length not covered
    assert g(2, 513, 3) == 2 * 3 * 64 * 2 * 512 and g(2, 512, 3) == 2 * 3 * 64 * 2 * 512 and g(2, 100, 3) == 0 and g(2, 1, 3) == 0


def test_no_cpu_fallback():
    from npcd.hip.attention import attention_qkvpacked, flash_attn_func
    with pytest.raises(RuntimeError, match="GPU"):
        attention_qkvpacked(torch.zeros(1, 4, 192, dtype=torch.bfloat16), 1)
    with pytest.raises(RuntimeError, match="GPU"):
        flash_attn_func(*(torch.zeros(1, 4, 1, 64, dtype=torch.bfloat16),) * 3)


def test_denoiser_state_dict_matches_reference(golden):
    from npcd.models.diffusion import NPCDTransformer
    g = golden("denoiser_f32_w64")
    ref = {k[2:]: tuple(v.shape) for k, v in g.items() if k.startswith("w:")}
    net = NPCDTransformer(coords_dim=3, feats_dim=32, width=64, layers=2, heads=1)
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == ref
    assert float(net.output_proj.weight.abs().sum()) == 0.0
    std = float(net.backbone.resblocks[0].attn.c_qkv.weight.std())
    assert std == pytest.approx(0.25 / 8.0, rel=0.1)


def test_timestep_embedding_host(golden):
    from npcd.models.diffusion.transformer import timestep_embedding
    g = golden("timestep_embedding")
    for dim in (128, 1024, 7):
        np.testing.assert_allclose(timestep_embedding(torch.from_numpy(g["t"]), dim).numpy(), g[f"dim{dim}"], atol=1e-6)


def test_gaussian_diffusion_host(golden):
    from npcd.models.diffusion import GaussianDiffusion
    g = golden("diffusion")
    gd = GaussianDiffusion()
    for k, v in g.items():
        if k.startswith("tab:"):
            np.testing.assert_array_equal(getattr(gd, k[4:]).numpy(), v, err_msg=k)
    T = torch.from_numpy
    wc, wf = T(g["wc"]), T(g["wf"])

    def fake(c, f, tt):
        s = (tt.float() / 1000.0).reshape(-1, 1, 1)
        return torch.einsum("ij,bjn->bin", wc, c) + s, torch.tanh(torch.einsum("ij,bjn->bin", wf, f)) - s

    loss, sub, pw = gd.p_losses(fake, T(g["c0"]), T(g["f0"]), T(g["t"]), T(g["cn"]), T(g["fn"]))
    assert float(loss) == pytest.approx(float(g["loss"]), abs=1e-7)
    np.testing.assert_allclose(pw["pointwise_feats_loss"].numpy(), g["pw_feats"], atol=1e-6)
    torch.manual_seed(1234)
    cn, cr, fn, fr = gd.p_sample(fake, T(g["c0"]), T(g["f0"]), T(g["t"]),
                                 (T(g["ps_clipc"])[:1], T(g["ps_clipc"])[1:]), (T(g["ps_clipf"])[:1], T(g["ps_clipf"])[1:]))
    np.testing.assert_allclose(cn.numpy(), g["ps_next_c"], atol=1e-6)
    np.testing.assert_allclose(fn.numpy(), g["ps_next_f"], atol=1e-6)
    np.testing.assert_allclose(cr.numpy(), g["ps_rec_c"], atol=1e-6)


def test_normalizers_host(golden):
    from npcd.models.diffusion import MinusOneToOneNormalization, UnitGaussianNormalization
    g = golden("normalizers")
    T = torch.from_numpy
    un, mm = UnitGaussianNormalization(3), MinusOneToOneNormalization(8)
    un.set_from_all_data(g["data_c"]); mm.set_from_all_data(g["data_f"])
    for k in ("shift", "scale", "min", "max"):
        np.testing.assert_allclose(getattr(un, k).numpy(), g["un:" + k], atol=1e-6)
        np.testing.assert_allclose(getattr(mm, k).numpy(), g["mm:" + k], atol=1e-6)
    assert set(un.state_dict()) == {"min", "max", "shift", "scale"}
    un.train(); mm.train()
    np.testing.assert_allclose(un(T(g["xc"])).numpy(), g["c_train"], atol=1e-6)
    un.eval(); mm.eval()
    np.testing.assert_allclose(mm(T(g["xf"])).numpy(), g["f_eval"], atol=1e-6)


def test_row_split_linear_matches_plain_linear():
    """train_path._RowSplitLinear (weight gradient summed over row slices, with and without a remainder slice) against
    F.linear + autograd in fp32 on the CPU."""
    import torch.nn.functional as F
    from npcd.models.pointnerf.train_path import _RowSplitLinear
    g = torch.Generator().manual_seed(0)
    old = _RowSplitLinear.SLICE
    _RowSplitLinear.SLICE = 64
    try:
        for rows in (64 * 3, 64 * 3 + 17, 100, 5):
            x = torch.randn(rows, 24, generator=g, requires_grad=True)
            w = torch.randn(16, 24, generator=g, requires_grad=True)
            b = torch.randn(16, generator=g, requires_grad=True)
            gy = torch.randn(rows, 16, generator=g)
            slope = 0.01 if rows % 2 else None                     # with and without the fused LeakyReLU
            y = _RowSplitLinear.apply(x, w, b, None, slope)
            y.backward(gy)
            got = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
            x.grad = w.grad = b.grad = None
            y2 = F.linear(x, w, b)
            if slope is not None:
                y2 = F.leaky_relu(y2, slope)
            y2.backward(gy)
            for a, r in zip(got, (y2.detach(), x.grad, w.grad, b.grad)):
                assert torch.allclose(a, r, atol=2e-5, rtol=1e-5), rows
            x.grad = w.grad = b.grad = None
    finally:
        _RowSplitLinear.SLICE = old


def test_bench_without_launcher_starts_ranks_as_a_child_and_returns_their_exit_code():
    """`python bench.py --gpus 2` with no WORLD_SIZE: bench.py must start torch.distributed.run itself (a child process, before
    any GPU call).  Without a GPU the ranks die at torch.cuda.set_device; what is checked here is the launcher: the command
    line it announces, that nothing but rank output reaches stdout, and that the child's failure is the script's exit code."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""          # (also on a GPU box: the ranks must not get as far as the benchmark)
    env["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=600)
    assert "starting 2 ranks" in out.stderr and "--nproc-per-node 2" in out.stderr and "--master-addr 127.0.0.1" in out.stderr
    assert out.returncode != 0                                   # no GPU -> the ranks fail -> so does bench.py, loudly
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_step_arena_hands_out_the_same_buffers_in_order():
    """npcd/hip/arena.py (round 6) on CPU tensors: the i-th request of a step gets the i-th buffer of the previous step when shape, dtype
    and device match; a changed shape replaces the slot; one step at a time (begin() refuses while a step is out, a StepGuard dropped with
    the graph ends it); paused() allocates outside and does not advance; outside an active arena empty() is torch.empty."""
    from npcd.hip import arena
    dev = torch.device("cpu")
    a = arena.StepArena()
    assert arena.empty((2, 3), torch.float32, dev).shape == (2, 3)           # no active arena: plain allocation
    tok = a.begin()
    assert tok > 0 and a.begin() == 0                                          # busy: a second step has to do without
    with a.active():
        x = arena.empty((4, 5), torch.float32, dev)
        y = arena.empty_like(x)
        with arena.paused():
            z = arena.empty((7,), torch.float32, dev)
        w = arena.empty(6, torch.int32, dev)
    a.end(tok)
    assert len(a.slots) == 3 and not a.busy and all(z is not s for s in a.slots)
    tok2 = a.begin()
    with a.active():
        x2, y2 = arena.empty((4, 5), torch.float32, dev), arena.empty((4, 5), torch.float32, dev)
        w2 = arena.empty((9,), torch.int32, dev)                               # another shape: the slot is replaced
    assert x2 is x and y2 is y and w2 is not w and w2.shape == (9,) and a.slots[2] is w2
    a.end(tok)                                                                 # a stale token does nothing
    assert a.busy
    a.end(tok2)
    assert not a.busy
    g = arena.StepGuard(a, a.begin())
    assert a.busy
    del g
    assert not a.busy                                                          # the guard ended the step its graph never ran
    tok3 = a.begin()
    with a.active():
        arena.empty((4, 5), torch.float32, dev)
    a.end(tok3)
    assert len(a.slots) == 1                                                   # a shorter step drops the tail


def test_stage1_numerics_selection_defaults_to_true_fp32():
    """PointNeRFTrainer's default must be the reference's fp32, the split-operand mode an explicit opt-in."""
    from npcd.hip import render as hr
    from npcd.models.pointnerf import PointNeRF, train_path as tp
    field = PointNeRF(1, 32, 64, False).field
    assert tp.fused_pair_mlp_precision(field, None) is None
    assert tp.fused_pair_mlp_precision(field, torch.float32) is None
    assert tp.fused_pair_mlp_precision(field, "library") is None
    assert tp.fused_pair_mlp_precision(field, "fp32_class") == hr.PAIR_MLP_X2
    assert tp.fused_pair_mlp_precision(field, torch.bfloat16) == hr.PAIR_MLP_BF16
    assert tp.point_layers_fused(field, "fp32_class") and not tp.point_layers_fused(field, None)
    assert tp.point_layers_fused(field, "fp32_class", 5000) and not tp.point_layers_fused(field, "fp32_class", 100)
    with pytest.raises(ValueError):
        tp.fused_pair_mlp_precision(field, "fp64")
    assert field.fp32_class_ok()


def test_weight_gradient_stream_switch_is_parsed_once():
    from npcd.models.diffusion import fused
    assert fused._parse_wgrad_stream(None) == (True, 20000) and fused._parse_wgrad_stream("") == (True, 20000)
    assert fused._parse_wgrad_stream("0") == (False, 0) and fused._parse_wgrad_stream("1") == (True, 1 << 62)
    assert fused._parse_wgrad_stream("12345") == (True, 12345)
