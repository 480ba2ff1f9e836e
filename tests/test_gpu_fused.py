"""GPU parity of the elementwise kernels and of the fused backbone forward/backward (explicit backward
over flat buffers) against plain PyTorch fp32 references / the module path."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


@pytest.mark.parametrize("T,W", [(37, 64), (130, 128), (513, 1024), (65, 2048)])
def test_add_ln_fwd_and_bwd(T, W):
    from npcd.hip import elementwise as ew
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, W, generator=g) * 2 + 0.3
    delta = torch.randn(T, W, generator=g).bfloat16()
    gamma, beta = torch.randn(W, generator=g) * 0.2 + 1, torch.randn(W, generator=g) * 0.1
    dy = torch.randn(T, W, generator=g).bfloat16()
    dres = torch.randn(T, W, generator=g)
    # reference (fp32)
    xr = (x + delta.float()).requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (W,), gr, br, 1e-5)
    yr.backward(dy.float())
    x_out, y, mean, rstd = ew.add_ln_fwd(x.cuda(), delta.cuda(), gamma.cuda(), beta.cuda())
    assert torch.allclose(x_out.cpu(), xr.detach(), atol=1e-6)
    assert rel(y, yr.detach()) < 4e-3                      # bf16 output rounding
    assert torch.allclose(mean.cpu(), xr.detach().mean(1), atol=1e-5)
    x2, y2, _, _ = ew.add_ln_fwd(x.cuda(), None, gamma.cuda(), beta.cuda())
    assert x2 is None and rel(y2, F.layer_norm(x, (W,), gamma, beta, 1e-5)) < 4e-3
    dgam, dbet, dcol = (torch.empty(W, device="cuda") for _ in range(3))
    dx, dxb = ew.ln_bwd(dy.cuda(), x_out, mean, rstd, gamma.cuda(), dres.cuda(), dgam, dbet, dcol)
    assert rel(dx, xr.grad + dres) < 1e-5
    assert rel(dxb, xr.grad + dres) < 4e-3
    assert rel(dgam, gr.grad) < 1e-5 and rel(dbet, br.grad) < 1e-5
    assert rel(dcol, (xr.grad + dres).sum(0)) < 1e-4
    dx2, none = ew.ln_bwd(dy.cuda(), x_out, mean, rstd, gamma.cuda(), None, dgam, dbet, None, want_bf16=False)
    assert none is None and rel(dx2, xr.grad) < 1e-5


@pytest.mark.parametrize("T,N", [(70, 256), (513, 4096)])
def test_gelu_and_colsum(T, N):
    from npcd.hip import elementwise as ew
    g = torch.Generator().manual_seed(N)
    h = (torch.randn(T, N, generator=g) * 2).bfloat16()
    dg = torch.randn(T, N, generator=g).bfloat16()
    hr = h.float().requires_grad_(True)
    yr = F.gelu(hr)
    yr.backward(dg.float())
    y = ew.gelu_fwd(h.cuda())
    assert rel(y, yr.detach()) < 4e-3
    db = torch.empty(N, device="cuda")
    dh = ew.gelu_bwd(dg.cuda(), h.cuda(), db)
    assert rel(dh, hr.grad) < 4e-3
    assert rel(db, dh.float().sum(0)) < 1e-5               # sums exactly what the GEMMs will see
    out = torch.empty(N, device="cuda")
    ew.colsum_bf16(dg.cuda(), out)
    assert rel(out, dg.float().sum(0)) < 1e-5


def test_adamw_ema_kernel_matches_torch():
    from npcd.hip import elementwise as ew
    g = torch.Generator().manual_seed(0)
    n = 4096 * 3 + 8
    p0 = torch.randn(n, generator=g)
    p = p0.clone().cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    ema = p.clone(); shadow = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=1e-2, weight_decay=0.01)
    ema_ref = p0.clone()
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        gdev = grad.clone().cuda()
        ew.adamw_ema(p, gdev, m, v, ema, shadow, 1e-2, 0.9, 0.999, 1e-8, 0.01, step, 0.9, zero_grad=True)
        assert float(gdev.abs().sum()) == 0.0
        ref.grad = grad
        opt.step()
        ema_ref.lerp_(ref.detach(), 0.1)
    assert torch.allclose(p.cpu(), ref.detach(), atol=2e-6)
    assert torch.allclose(ema.cpu(), ema_ref, atol=2e-6)
    assert torch.equal(shadow.cpu(), p.cpu().bfloat16())


def _models(L=2, W=128, H=2, F_=32, N=48):
    from npcd.models.diffusion import DiffusionModel
    torch.manual_seed(3)
    a = DiffusionModel(3, F_, N, W, L, H, True)
    with torch.no_grad():
        a.denoiser.output_proj.weight.normal_(0, 0.05)
        for mod in a.modules():
            if isinstance(mod, torch.nn.LayerNorm):
                mod.weight.normal_(1, 0.1); mod.bias.normal_(0, 0.1)
            if isinstance(mod, torch.nn.Linear):
                mod.bias.normal_(0, 0.05)
    import copy
    return a.cuda().train(), copy.deepcopy(a).cuda().train()


def test_fused_backbone_matches_module_path():
    """Same weights, same batch: explicit fused forward/backward vs the nn.Module/autograd path."""
    from npcd.train import DiffusionTrainer
    a, b = _models()
    ta = DiffusionTrainer(a, fused=True)
    tb = DiffusionTrainer(b, fused=False)
    assert a.denoiser.backbone.fused_engine is not None and b.denoiser.backbone.fused_engine is None
    g = torch.Generator().manual_seed(1)
    B, N, F_ = 3, 48, 32
    c0, f0 = torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()
    t = torch.tensor([5, 500, 990]).cuda()
    cn, fn = torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()
    # gradients after one backward (before the optimizer consumes them): run the pieces by hand
    for tr in (ta, tb):
        tr.flat.zero_grad(); tr.reducer.start_step()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss, _, _ = tr.model.compute_loss(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
        loss.backward()
        tr.loss_ = float(loss)
    assert abs(ta.loss_ - tb.loss_) < 2e-3 * abs(tb.loss_)
    names = [n for n, _ in a.named_parameters()]
    worst = ("", 0.0)
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if float(pb.grad.abs().max()) < 1e-6:
            continue
        e = rel(pa.grad, pb.grad)
        if e > worst[1]:
            worst = (n, e)
    assert worst[1] < 3e-2, worst
    # full steps stay together
    for _ in range(3):
        la, _ = ta.step(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
        lb, _ = tb.step(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
    assert abs(float(la) - float(lb)) < 5e-3 * abs(float(lb))
    assert rel(ta.flat.flat, tb.flat.flat) < 1e-3
    assert rel(ta.ema, tb.ema) < 1e-4


def test_weight_gradients_on_the_side_stream_are_the_same_bits():
    """The weight-gradient products of a block run on a second HIP stream (fused._WGRAD_STREAM, the default since round 5) and are joined
    before the block's gradients are handed on: every gradient after a backward, and the parameters after three optimizer steps, are
    bit-identical to the single-stream order."""
    from npcd.models.diffusion import fused
    from npcd.train import DiffusionTrainer
    assert fused._WGRAD_STREAM is True
    a, b = _models()
    ta, tb = DiffusionTrainer(a, fused=True), DiffusionTrainer(b, fused=True)
    g = torch.Generator().manual_seed(3)
    B, N, F_ = 5, 48, 32
    c0, f0 = torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()
    t = torch.tensor([5, 500, 990, 17, 640]).cuda()
    cn, fn = torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()
    try:
        for tr, side in ((ta, True), (tb, False)):
            fused._WGRAD_STREAM = side
            tr.flat.zero_grad(); tr.reducer.start_step()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss, _, _ = tr.model.compute_loss(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
            loss.backward()
            torch.cuda.synchronize()
        for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            assert torch.equal(pa.grad, pb.grad), n
        for tr, side in ((ta, True), (tb, False)):
            fused._WGRAD_STREAM = side
            for _ in range(3):
                tr.step(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
        torch.cuda.synchronize()
        assert torch.equal(ta.flat.flat, tb.flat.flat)
    finally:
        fused._WGRAD_STREAM = True


def test_step_arena_reuses_buffers_and_changes_no_bits():
    """npcd/hip/arena.py (round 6): the buffers a fused step allocates are handed out again, in order, by the next step.  Same kernels on
    the same values: parameters after four steps on four different batches are bit-identical to a trainer without the arena; the second
    step allocates nothing new; a forward whose graph is dropped without a backward releases the arena; two forwards before their
    backwards (the second one falls back to plain allocations) still give the gradients of two separate steps; what leaves the node
    (its output, the gradient of its input) is never an arena buffer."""
    import gc
    from npcd.models.diffusion import fused
    from npcd.train import DiffusionTrainer
    a, b = _models()
    ta = DiffusionTrainer(a, fused=True)
    eng = a.denoiser.backbone.fused_engine
    assert eng.arena is not None
    old = fused._STEP_ARENA
    fused._STEP_ARENA = False
    try:
        tb = DiffusionTrainer(b, fused=True)
    finally:
        fused._STEP_ARENA = old
    assert b.denoiser.backbone.fused_engine.arena is None
    g = torch.Generator().manual_seed(11)
    B, N, F_ = 3, 48, 32
    batches = [(torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda(), torch.randint(0, 1000, (B,), generator=g).cuda(),
                torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()) for _ in range(4)]
    for i, (c0, f0, t, cn, fn) in enumerate(batches):
        la, _ = ta.step(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
        lb, _ = tb.step(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
        assert torch.equal(la, lb), i
        if i == 0:
            recorded = len(eng.arena.slots)
            assert recorded > 20 and not eng.arena.busy
        else:
            assert len(eng.arena.slots) == recorded                  # nothing new after the first step
    torch.cuda.synchronize()
    assert torch.equal(ta.flat.flat, tb.flat.flat) and eng.arena.hits >= 3 * recorded
    # a dropped graph releases the arena
    c0, f0, t, cn, fn = batches[0]
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss, _, _ = ta.model.compute_loss(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
    assert eng.arena.busy
    del loss
    gc.collect()
    assert not eng.arena.busy
    # two forwards, then their backwards: gradients of each equal those of the same forward + backward on its own
    def grads(batch_list, interleaved):
        outs = []
        ta.flat.zero_grad(); ta.reducer.start_step()
        losses = []
        for c0, f0, t, cn, fn in batch_list:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                l, _, _ = ta.model.compute_loss(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
            if interleaved:
                losses.append(l)
            else:
                l.backward()
                outs.append(ta.flat.grad.clone())
        for l in losses:
            l.backward()
            outs.append(ta.flat.grad.clone())
        torch.cuda.synchronize()
        return outs
    sep = grads(batches[:2], interleaved=False)
    both = grads(batches[:2], interleaved=True)
    # (the fused node OVERWRITES its gradient ranges: after the second backward the block gradients are those of the second batch)
    lo, hi = eng.block_ranges[0][0] if eng.block_ranges else 0, eng.block_ranges[-1][1] if eng.block_ranges else ta.flat.numel
    assert torch.equal(sep[1][lo:hi], both[1][lo:hi]) and torch.equal(sep[0][lo:hi], both[0][lo:hi])
    assert not eng.arena.busy
    # the node's output and input gradient are not arena buffers
    x = torch.randn(2, N + 1, a.denoiser.backbone.width, device="cuda", requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = a.denoiser.backbone(x)
    y.float().sum().backward()
    ptrs = {s.data_ptr() for s in eng.arena.slots}
    assert y.data_ptr() not in ptrs and x.grad.data_ptr() not in ptrs
    ta.close(); tb.close()


def test_side_stream_weight_gradients_at_a_rank_sized_batch_are_the_same_bits():
    """The same bit-identity at the size the side stream is the default FOR (it was tested at T = 245 only): width 1,024, two
    blocks, 38 x 513 = 19,494 token rows -- just under fused._WGRAD_STREAM_MAX_T = 20,000, above _SPLIT_MIN, so the token split of the
    GEMMs (large call + left-over rows) and the row-split weight gradients run as in a rank's step.  Also: the environment variable is
    parsed once into (enabled, max rows)."""
    from npcd.models.diffusion import DiffusionModel, fused
    from npcd.train import DiffusionTrainer
    assert fused._parse_wgrad_stream(None) == (True, 20000) and fused._parse_wgrad_stream("0") == (False, 0)
    assert fused._parse_wgrad_stream("1") == (True, 1 << 62) and fused._parse_wgrad_stream("5000") == (True, 5000)
    B, N, F_, W = 38, 512, 32, 1024
    assert fused._SPLIT_MIN <= B * (N + 1) < fused._WGRAD_STREAM_MAX_T and fused._wgrad_side_ok(B * (N + 1))
    g = torch.Generator().manual_seed(7)
    c0, f0 = torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()
    t = torch.randint(0, 1000, (B,), generator=g).cuda()
    cn, fn = torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()
    grads = {}
    try:
        for side in (True, False):
            torch.manual_seed(9)
            m = DiffusionModel(3, F_, N, W, 2, 16, True)
            torch.nn.init.normal_(m.denoiser.output_proj.weight, std=0.02)
            tr = DiffusionTrainer(m.cuda().train(), fused=True)
            fused._WGRAD_STREAM = side
            tr.flat.zero_grad(); tr.reducer.start_step()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss, _, _ = tr.model.compute_loss(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
            loss.backward()
            torch.cuda.synchronize()
            grads[side] = tr.flat.grad.clone()
            tr.close()
            del tr, m
        assert float(grads[True].abs().sum()) > 0 and torch.equal(grads[True], grads[False])
    finally:
        fused._WGRAD_STREAM = True


@pytest.mark.parametrize("tag", ["f32_w64", "f128_w64", "f32_w128_h2"])
@pytest.mark.parametrize("half", [torch.bfloat16, torch.float16])
def test_fused_engine_matches_reference_golden_directly(golden, tag, half):
    """The BENCHMARKED code path (_BackboneFn: one autograd node, hand-written backward, flat buffers, 16-bit shadow) against
    the reference's own fp32 outputs and parameter gradients (fixtures denoiser_*.npz, generated by importing
    /root/reference) -- no detour over this package's module path.  Bars as for the module path: eps rel-L2 <= 2e-2,
    every parameter gradient rel-L2 <= 5e-2 (16-bit GEMM inputs against an fp32 reference).  Both activation types of the
    engine: bf16, and f16 (the reference's default --dtype; the golden loss is O(1), so no loss scaling is needed here)."""
    from npcd.hip import elementwise as ew
    from npcd.models.diffusion import NPCDTransformer
    from npcd.models.diffusion.fused import FusedBackboneEngine
    from npcd.train.engine import FlatBuffers
    g = golden("denoiser_" + tag)
    T = torch.from_numpy
    F_ = g["feats"].shape[1]
    net = NPCDTransformer(coords_dim=3, feats_dim=F_, width=int(g.get("width", 64)), layers=int(g.get("layers", 2 if tag == "f32_w64" else 1)),
                          heads=int(g["heads"]))
    net.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w:")})
    net = net.cuda()
    flat = FlatBuffers(net)
    shadow = torch.empty(flat.numel, dtype=half, device="cuda")
    ew.cast_f32_bf16(flat.flat, shadow)
    net.backbone.fused_engine = FusedBackboneEngine(net.backbone, flat, shadow)
    assert net.backbone.fused_engine.dtype == half
    calls = []
    import npcd.models.diffusion.fused as fused
    orig = fused._BackboneFn.apply
    fused._BackboneFn.apply = staticmethod(lambda *a: (calls.append(1), orig(*a))[1])
    try:
        with torch.autocast("cuda", dtype=half):
            ec, ef = net(T(g["coords"]).cuda(), T(g["feats"]).cuda(), T(g["t"]).cuda())
            loss = (ec.float() * T(g["gc"]).cuda()).sum() + (ef.float() * T(g["gf"]).cuda()).sum()
        loss.backward()
    finally:
        fused._BackboneFn.apply = orig
    assert calls, "the fused backbone node was not on the path"
    assert rel(ec, T(g["eps_coords"]).cuda()) < 2e-2 and rel(ef, T(g["eps_feats"]).cuda()) < 2e-2
    worst = ("", 0.0)
    named = dict(net.named_parameters())
    for k, v in g.items():
        if k.startswith("g:") and np.abs(v).max() > 1e-3:
            e = rel(named[k[2:]].grad, T(v).cuda())
            if e > worst[1]:
                worst = (k[2:], e)
    assert worst[1] < 5e-2, worst


def test_shadow_follows_parameter_writes_outside_the_optimizer():
    """The fused paths read GEMM weights from the trainer's bf16 shadow; load_state_dict / EMA copies write the fp32 masters
    behind its back.  The engine notices (parameter version stamps) and re-casts before the next forward: training forward
    and no-grad sampling forward both see the new weights, LayerNorm and Linear consistently."""
    from npcd.train import DiffusionTrainer
    a, b = _models()
    ta = DiffusionTrainer(a, fused=True)
    g = torch.Generator().manual_seed(5)
    B, N, F_ = 2, 48, 32
    c0, f0 = torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()
    t = torch.tensor([7, 700]).cuda()
    with torch.no_grad():
        for p in b.parameters():
            p.add_(torch.randn_like(p) * 0.05)                     # a different set of weights
    a.load_state_dict(b.state_dict())                              # written into the flat buffer, shadow now stale
    tb = DiffusionTrainer(b, fused=True)                           # fresh shadow of the same weights
    for mode in ("train", "sample"):
        outs = []
        for m in (a, b):
            if mode == "train":
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    outs.append(m.denoiser(c0, f0, t)[1].float())
            else:
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                    outs.append(m.denoiser(c0, f0, t)[1].float())
        assert torch.equal(outs[0], outs[1]), mode
    assert torch.equal(ta.shadow, tb.shadow)
    # sampling from the EMA weights and switching back
    keep = {k: v.clone() for k, v in a.state_dict().items()}
    ta.step(c0, f0, t=t)
    a.load_state_dict(ta.ema_state_dict())
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        e1 = a.denoiser(c0, f0, t)[1].float()
    a.load_state_dict(keep)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        e2 = a.denoiser(c0, f0, t)[1].float()
        e3 = b.denoiser(c0, f0, t)[1].float()
    assert torch.equal(e2, e3) and not torch.equal(e1, e2)


def test_forward_only_backbone_for_sampling():
    """no_grad + bf16 autocast (the sampler / evaluation): fused forward-only kernels vs the module path and vs fp32; the
    bf16 weight copies follow in-place parameter updates; with a trainer attached the trainer's shadow weights are used."""
    a, b = _models()
    a.eval(); b.eval()
    g = torch.Generator().manual_seed(2)
    B, N, F_ = 3, 48, 32
    c, f = torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()
    t = torch.tensor([3, 400, 999]).cuda()
    bb = a.denoiser.backbone
    with torch.no_grad():
        ref32 = torch.cat(b.denoiser(c, f, t), dim=1)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got = torch.cat(a.denoiser(c, f, t), dim=1)
            assert bb._infer_weights is not None                  # the forward-only path ran
            h = torch.randn(B, N + 1, bb.width, generator=g).cuda()
            fused_out = bb(h)
            x = h
            for blk in bb.resblocks:                              # the nn.Module path under the same autocast
                x = blk(x)
            assert rel(fused_out, x) < 1e-2
    assert rel(got, ref32) < 2e-2
    # in-place update of a weight -> the bf16 copy is refreshed
    with torch.no_grad():
        bb.resblocks[0].mlp.c_fc.weight.mul_(1.5)
        b.denoiser.backbone.resblocks[0].mlp.c_fc.weight.mul_(1.5)
        ref2 = torch.cat(b.denoiser(c, f, t), dim=1)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got2 = torch.cat(a.denoiser(c, f, t), dim=1)
    assert rel(got2, ref2) < 2e-2 and rel(got2, got) > 1e-3
    # attached to a trainer: evaluation uses the trainer's bf16 shadow, which follows the optimizer steps
    from npcd.train import DiffusionTrainer
    a.train()
    tr = DiffusionTrainer(a, fused=True)
    tr.step(c, f, t=t, coords_noise=torch.randn_like(c), feats_noise=torch.randn_like(f))
    a.eval()
    with torch.no_grad():
        ref3 = torch.cat(a.denoiser(c, f, t), dim=1)              # fp32 module path on the updated master weights
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got3 = torch.cat(a.denoiser(c, f, t), dim=1)
    assert rel(got3, ref3) < 2e-2


def test_stress_config_shape_step():
    """BASELINE configs[4] shape: 2048 points x 256-d latents, 8-layer denoiser (width 1024, 16 heads -> sequence 2049),
    a small batch of it: the fused engine's step against the nn.Module / autograd path with the same weights and draws."""
    import copy
    from npcd.models.diffusion import DiffusionModel
    from npcd.train import DiffusionTrainer
    torch.manual_seed(0)
    a = DiffusionModel(3, 256, 2048, 1024, 8, 16, True)
    with torch.no_grad():
        a.denoiser.output_proj.weight.normal_(0, 0.02)
    b = copy.deepcopy(a)
    ta, tb = DiffusionTrainer(a.cuda(), fused=True), DiffusionTrainer(b.cuda(), fused=False)
    assert a.denoiser.backbone.fused_engine is not None
    g = torch.Generator().manual_seed(1)
    B = 2
    c, f = torch.randn(B, 3, 2048, generator=g).cuda(), torch.randn(B, 256, 2048, generator=g).cuda()
    t = torch.tensor([10, 900]).cuda()
    cn, fn = torch.randn(B, 3, 2048, generator=g).cuda(), torch.randn(B, 256, 2048, generator=g).cuda()
    for _ in range(2):
        la, _ = ta.step(c, f, t=t, coords_noise=cn, feats_noise=fn)
        lb, _ = tb.step(c, f, t=t, coords_noise=cn, feats_noise=fn)
        assert math.isfinite(float(la)) and abs(float(la) - float(lb)) < 5e-3 * abs(float(lb))
    assert rel(ta.flat.flat, tb.flat.flat) < 1e-3


def test_stress_config_step_at_per_gpu_batch_32_fp8_and_bf16_attention():
    """BASELINE configs[4] as one rank of its 8-GPU job runs it: 2048 points x 256-d latents, 8-layer denoiser (width 1024, 16 heads,
    sequence 2049), PER-GPU batch 32 of the global 256; one DiffusionTrainer.step with the attention forward on the fp8 (e4m3,
    block-scaled MFMA) kernel and one on the bf16 kernel, same weights and draws.  Bars: both losses finite, |loss(fp8) - loss(bf16)|
    <= 2e-2 relative (measured 1e-3), eps-hat of sample 0 against the fp32 CPU oracle (oracle.denoiser, the same weights): rel-L2
    <= 3e-2 with bf16 attention (measured 6e-3) and <= 8e-2 with fp8 attention (the attention-level bar of test_attention_fp8_forward)."""
    import copy
    from npcd.hip import attention as hattn
    from npcd.models.diffusion import DiffusionModel
    from npcd.train import DiffusionTrainer
    from oracle import denoiser as od
    torch.manual_seed(0)
    base = DiffusionModel(3, 256, 2048, 1024, 8, 16, True)
    with torch.no_grad():
        base.denoiser.output_proj.weight.normal_(0, 0.02)
    g = torch.Generator().manual_seed(1)
    B = 32
    c, f = torch.randn(B, 3, 2048, generator=g), torch.randn(B, 256, 2048, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    cn, fn = torch.randn(B, 3, 2048, generator=g), torch.randn(B, 256, 2048, generator=g)
    params = {k: v.detach().clone() for k, v in base.denoiser.state_dict().items()}
    x_c = base.diffusion_process.q_sample(c, t, cn)
    x_f = base.diffusion_process.q_sample(f, t, fn)
    with torch.no_grad():
        ref_c, ref_f = od.denoiser_forward(params, x_c[:1], x_f[:1], t[:1], 16)      # one sample through the fp32 oracle
    ref = torch.cat((ref_c, ref_f), dim=1)
    losses, errs = {}, {}
    saved = hattn.FWD_FP8
    try:
        for mode in ("bf16", "fp8"):
            hattn.FWD_FP8 = mode == "fp8"
            m = copy.deepcopy(base).cuda().train()
            tr = DiffusionTrainer(m)
            assert tr.native and m.denoiser.backbone.fused_engine is not None
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                e_c, e_f = m.denoiser(x_c[:1].cuda(), x_f[:1].cuda(), t[:1].cuda())
            errs[mode] = rel(torch.cat((e_c, e_f), dim=1).float().cpu(), ref)
            loss, _ = tr.step(c.cuda(), f.cuda(), t=t.cuda(), coords_noise=cn.cuda(), feats_noise=fn.cuda())
            losses[mode] = float(loss)
            assert math.isfinite(losses[mode])
            loss2, _ = tr.step(c.cuda(), f.cuda(), t=t.cuda(), coords_noise=cn.cuda(), feats_noise=fn.cuda())
            assert math.isfinite(float(loss2))
            del tr, m
            torch.cuda.empty_cache()
    finally:
        hattn.FWD_FP8 = saved
    print(f"cfg-5 per-GPU batch 32: loss bf16 {losses['bf16']:.5f} fp8 {losses['fp8']:.5f}; eps-hat rel-L2 vs oracle bf16 {errs['bf16']:.2e} fp8 {errs['fp8']:.2e}")
    assert abs(losses["fp8"] - losses["bf16"]) <= 2e-2 * abs(losses["bf16"])
    assert errs["bf16"] <= 3e-2 and errs["fp8"] <= 8e-2


@pytest.mark.parametrize("T,N,K", [(513 * 8, 1024, 1024), (2052, 768, 256), (700, 256, 512), (32, 256, 256)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_own_weight_gradient_kernel(T, N, K, dtype):
    """npcd_wgrad (csrc/gemm.hip: dW = dy^T x, split over the token range, slabs added in slice order) against the fp64 product of
    the same 16-bit operands: fp32 accumulation -> 1e-5 of the largest entry; token counts that are no multiple of the 32-token
    ring stage (rows past T come from a zero page); bitwise reproducible."""
    from npcd.hip import elementwise as ew
    g = torch.Generator().manual_seed(T + N)
    dy = torch.randn(T, N, generator=g).to(dtype).cuda()
    x = torch.randn(T, K, generator=g).to(dtype).cuda()
    out = torch.full((N, K), float("nan"), device="cuda")
    assert ew.wgrad(dy, x, out)
    ref = dy.double().t() @ x.double()
    assert float((out.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    out2 = torch.empty_like(out)
    assert ew.wgrad(dy, x, out2) and torch.equal(out, out2)
    assert not ew.wgrad(dy[:, :100].contiguous(), x, torch.empty(100, K, device="cuda"))      # shapes it does not cover: the caller's fallback


@pytest.mark.parametrize("T", [513 * 8, 700])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_grouped_weight_gradient_launch(T, dtype):
    """npcd_wgrad_group (round 6): the four weight gradients of a residual block at width 1,024 -- (3072, 1024), (1024, 1024), (4096, 1024),
    (1024, 4096) -- over one token range in ONE launch, a workgroup per 256 x 256 tile over all T tokens.  T = 4,104 is one rank's token
    count at per-GPU batch 8.  Against the fp64 product of the same 16-bit operands: 1e-5 of the largest entry;
    bitwise reproducible; equal to the single-product launch's bits wherever that one does not split the token range."""
    from npcd.hip import elementwise as ew
    g = torch.Generator().manual_seed(T)
    shapes = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]
    trip = []
    for N, K in shapes:
        dy = torch.randn(T, N, generator=g).to(dtype).cuda()
        x = torch.randn(T, K, generator=g).to(dtype).cuda()
        trip.append((dy, x, torch.full((N, K), float("nan"), device="cuda")))
    assert ew.wgrad_group(trip)
    for dy, x, out in trip:
        ref = dy.double().t() @ x.double()
        assert float((out.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    again = [(dy, x, torch.empty_like(out)) for dy, x, out in trip]
    assert ew.wgrad_group(again)
    assert all(torch.equal(a[2], b[2]) for a, b in zip(trip, again))
    if ew.lib().npcd_wgrad_slices(T, 1024, 1024) == 1:
        one = torch.empty_like(trip[1][2])
        assert ew.wgrad(trip[1][0], trip[1][1], one) and torch.equal(one, trip[1][2])
    assert not ew.wgrad_group([(trip[0][0][:, :100].contiguous(), trip[0][1], torch.empty(100, 1024, device="cuda"))])
    assert not ew.wgrad_group(trip + trip + [trip[0]])           # more than eight products: the caller's per-product path


def test_float16_training_with_dynamic_loss_scale():
    """dtype=float16 (the reference's default --dtype, train_diffusion.py:78): the fused backbone engine IS engaged (f16 shadow,
    f16 activations through the same kernels), scaled backward, overflow -> skipped step and halved scale."""
    from npcd.train import DiffusionTrainer
    import npcd.models.diffusion.fused as fused
    a, b = _models()
    tr = DiffusionTrainer(a, dtype=torch.float16)
    ref = DiffusionTrainer(b, dtype=torch.bfloat16, fused=False)
    eng = a.denoiser.backbone.fused_engine
    assert tr.loss_scale == 65536.0 and tr.native and eng is not None and eng.dtype == torch.float16 and tr.shadow.dtype == torch.float16
    calls = []
    orig = fused._BackboneFn.apply
    fused._BackboneFn.apply = staticmethod(lambda *x: (calls.append(1), orig(*x))[1])
    g = torch.Generator().manual_seed(2)
    B, N, F_ = 3, 48, 32
    c0, f0 = torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()
    t = torch.tensor([7, 400, 900]).cuda()
    cn, fn = torch.randn(B, 3, N, generator=g).cuda(), torch.randn(B, F_, N, generator=g).cuda()
    try:
        l16, _ = tr.step(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
    finally:
        fused._BackboneFn.apply = orig
    assert calls, "float16 training did not go through the fused backbone node"
    lref, _ = ref.step(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
    assert torch.isfinite(l16) and abs(float(l16) - float(lref)) < 2e-2 * abs(float(lref))
    assert tr.skipped_steps == 0 and tr.iteration == 1
    # after one applied step the two trainers' parameters moved the same way (AdamW's first step is +-lr per element)
    moved = [(pa - pb).abs().max() for pa, pb in zip(a.parameters(), b.parameters())]
    assert float(max(moved)) < 3 * tr.lr
    # force an overflow: the step must be skipped, the scale halved, parameters and step count untouched
    before = tr.flat.flat.clone()
    tr.loss_scale = 2.0 ** 40
    tr.step(c0, f0, t=t, coords_noise=cn, feats_noise=fn)
    assert tr.skipped_steps == 1 and tr.loss_scale == 2.0 ** 39 and tr.iteration == 1
    assert torch.equal(tr.flat.flat, before) and float(tr.flat.grad.abs().max()) == 0.0


def test_fused_q_sample_and_eps_mse_match_reference_golden(golden):
    """npcd_q_sample / npcd_eps_mse_{fwd,bwd} against the reference's own q_sample and p_losses outputs (fixture diffusion.npz,
    generated by importing the reference): x_t BIT-EXACT (the kernel rounds the two products and the sum separately, like the
    reference's eager ops), loss terms to 1e-6 relative, pointwise losses to 1e-7 absolute; the backward against torch autograd
    of the same expression, fp32 and bf16 predictions."""
    from npcd.hip import elementwise as ew
    from npcd.models.diffusion import GaussianDiffusion
    g = golden("diffusion")
    T = lambda k: torch.from_numpy(g[k]).cuda()
    gd = GaussianDiffusion().cuda()
    # the schedule tables are built on the HOST (float64 linspace / cumprod): their last bits depend on the host's numpy build
    # (observed: this container's Xeon and the GPU box's EPYC differ in a few entries), exactly as the reference's would --
    # the fixture's tables are loaded so that the comparison is about the kernel
    gd.sqrt_alphas_cumprod.copy_(T("tab:sqrt_alphas_cumprod"))
    gd.sqrt_one_minus_alphas_cumprod.copy_(T("tab:sqrt_one_minus_alphas_cumprod"))
    t = T("t")
    for x0, nz, xt in (("c0", "cn", "coords_t"), ("f0", "fn", "feats_t")):
        out = gd.q_sample(T(x0), t, T(nz))
        assert torch.equal(out, T(xt)), x0
    # loss: the fixture's "denoiser" is a fixed function of (x_t, t) (make_golden.py: a channel-mixing matrix, + t / 1000 for the
    # coordinates, tanh(.) - t / 1000 for the features), so eps_hat can be rebuilt here (on the CPU, in the fixture's own op order)
    sft = (torch.from_numpy(g["t"]).float() / 1000.0).reshape(-1, 1, 1)
    eps_of = {"coords_t": torch.einsum("ij,bjn->bin", torch.from_numpy(g["wc"]), torch.from_numpy(g["coords_t"])) + sft,
              "feats_t": torch.tanh(torch.einsum("ij,bjn->bin", torch.from_numpy(g["wf"]), torch.from_numpy(g["feats_t"]))) - sft}
    for nz, xt, wk, lk, pk in (("cn", "coords_t", "wc", "coords_loss", "pw_coords"), ("fn", "feats_t", "wf", "feats_loss", "pw_feats")):
        eps = eps_of[xt].cuda().requires_grad_(True)
        loss, pw = ew.eps_mse(eps, T(nz), want_pointwise=True)
        assert abs(float(loss) - float(g[lk])) < 1e-6 * abs(float(g[lk]))
        assert float((pw - T(pk)).abs().max()) < 1e-7 * max(1.0, float(T(pk).abs().max()))
        (loss * 3.0).backward()
        e2 = eps.detach().clone().requires_grad_(True)
        (((T(nz) - e2) ** 2 / 2.0).mean() * 3.0).backward()
        assert torch.allclose(eps.grad, e2.grad, rtol=1e-6, atol=1e-12)
        eb = eps.detach().bfloat16().requires_grad_(True)
        lb, none = ew.eps_mse(eb, T(nz), want_pointwise=False)
        assert none is None
        lb.backward()
        e3 = eb.detach().clone().requires_grad_(True)
        ((T(nz) - e3.float()) ** 2 / 2.0).mean().backward()
        assert abs(float(lb) - float(((T(nz) - eb.detach().float()) ** 2 / 2).mean())) < 1e-6 * float(lb)
        assert rel(eb.grad, e3.grad) < 4e-3 and eb.grad.dtype == torch.bfloat16


@pytest.mark.parametrize("S", [2, 4, 8])
def test_sum_slices_is_the_ordered_sum(S):
    """csrc/elementwise.hip sum_slices_kernel (the weight-gradient partials of the row-split GEMMs): slices added in order, bitwise."""
    from npcd.hip import elementwise as ew
    g = torch.Generator().manual_seed(S)
    part = torch.randn(S, 1024, 772, generator=g).cuda()
    out = torch.empty(1024, 772, device="cuda")
    assert ew.sum_slices(part, out)
    ref = part[0].clone()
    for s in range(1, S):
        ref = ref + part[s]
    assert torch.equal(out, ref)
    assert not ew.sum_slices(part[:, :3, :5].contiguous(), torch.empty(3, 5, device="cuda"))             # numel % 4 != 0: declined


@pytest.mark.gpu
@pytest.mark.parametrize("T,J,K", [(1, 1, 256), (257, 3, 256), (70001, 1, 256), (200003, 3, 256), (5000, 4, 64), (3000, 2, 2048)])
def test_small_wgrad_matches_the_fp32_product(T, J, K):
    """csrc/elementwise.hip small_wgrad_kernel (weight gradient of the field heads' last layers, fields/mlp.py:38-72): dy^T x with
    fp32 accumulation of the bf16 operands; tolerance 1e-5 relative to the column's sum of magnitudes (fp32 summation order only)."""
    from npcd.hip import elementwise as ew
    g = torch.Generator(device="cuda").manual_seed(T + J)
    dy = torch.randn(T, J, device="cuda", generator=g).bfloat16()
    x = torch.randn(T, K, device="cuda", generator=g).bfloat16()
    out = ew.small_wgrad(dy, x)
    assert out is not None and out.shape == (J, K) and out.dtype == torch.float32
    ref = dy.double().t() @ x.double()
    mag = dy.double().abs().t() @ x.double().abs()
    assert ((out.double() - ref).abs() <= 1e-5 * mag + 1e-30).all()
    assert torch.equal(out, ew.small_wgrad(dy, x))                                      # fixed summation order
    assert ew.small_wgrad(dy.float(), x) is None and ew.small_wgrad(dy, x[:, :48].contiguous()) is None   # declined, caller falls back


@pytest.mark.parametrize("own_dgelu", [False, True])
def test_full_width_two_blocks_elementwise_gradients_vs_oracle(own_dgelu, monkeypatch):
    """Engine-level parity at the BENCHMARK width: W 1024 / H 16 / n = 513 (N = 512 points + the time token) / L 2 / B 2 on the
    fused engine under bf16 autocast against the CPU oracle (oracle/denoiser.py, pinned to the reference by the denoiser_*.npz
    fixtures) on the same weights and inputs -- ELEMENTWISE rel-L2 of the eps prediction and of named parameter gradients (a
    head permutation, a wrong column-sum by-product or a mis-handled edge token would pass a comparison of gradient NORMS).
    Non-trivial biases and LayerNorm affines.  Bars: eps 2e-2, every checked gradient 5e-2 (bf16 GEMM operands and attention
    against fp32)."""
    from oracle import denoiser as od
    from npcd.models.diffusion import NPCDTransformer
    from npcd.train import DiffusionTrainer
    from npcd.models.diffusion import DiffusionModel
    import npcd.models.diffusion.fused as fused
    # own_dgelu: the opt-in data gradient of mlp.c_proj on the own NT GEMM with the GELU backward + bias column sums in its epilogue
    # (csrc/gemm_nt.hip); T = 1026 = 4 full 256-row tiles + 2 remainder rows, so both of its row ranges run
    monkeypatch.setattr(fused, "_OWN_DGELU", own_dgelu)
    W, H, L, N, Fd, B = 1024, 16, 2, 512, 128, 2
    params = od.init_params(3, Fd, W, L, H, seed=3)
    g = torch.Generator().manual_seed(17)
    for k in params:                                   # the synthetic init has zero biases and unit LayerNorms: perturb them
        if k.endswith(".bias"):
            params[k] = params[k] + torch.randn(params[k].shape, generator=g) * 0.05
        elif ".ln_" in k or k.startswith("ln_"):
            params[k] = params[k] + torch.randn(params[k].shape, generator=g) * 0.1
    coords, feats = torch.randn(B, 3, N, generator=g), torch.rand(B, Fd, N, generator=g) * 2 - 1
    t = torch.tensor([11, 873])
    gc, gf = torch.randn(B, 3, N, generator=g), torch.randn(B, Fd, N, generator=g)
    torch.set_num_threads(min(16, torch.get_num_threads() or 1))
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ec_r, ef_r = od.denoiser_forward(leaves, coords, feats, t, H)
    ((ec_r * gc).sum() + (ef_r * gf).sum()).backward()

    model = DiffusionModel(3, Fd, N, W, L, H, True)
    model.denoiser.load_state_dict(params)
    model = model.cuda().train()
    tr = DiffusionTrainer(model, dtype=torch.bfloat16)
    assert tr.native and model.denoiser.backbone.fused_engine is not None
    tr.flat.zero_grad()
    tr.reducer.start_step()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ec, ef = model.denoiser(coords.cuda(), feats.cuda(), t.cuda())
        loss = (ec.float() * gc.cuda()).sum() + (ef.float() * gf.cuda()).sum()
    loss.backward()
    torch.cuda.synchronize()
    assert rel(ec, ec_r.detach()) < 2e-2 and rel(ef, ef_r.detach()) < 2e-2
    named = dict(model.denoiser.named_parameters())
    checked = ["input_proj.weight", "time_embed.c_fc.weight", "ln_pre.weight",
               "backbone.resblocks.0.ln_1.weight", "backbone.resblocks.0.ln_1.bias",
               "backbone.resblocks.0.attn.c_qkv.weight", "backbone.resblocks.0.attn.c_qkv.bias",
               "backbone.resblocks.0.attn.c_proj.weight", "backbone.resblocks.0.attn.c_proj.bias",
               "backbone.resblocks.0.ln_2.weight", "backbone.resblocks.0.mlp.c_fc.weight", "backbone.resblocks.0.mlp.c_fc.bias",
               "backbone.resblocks.0.mlp.c_proj.weight", "backbone.resblocks.0.mlp.c_proj.bias",
               "backbone.resblocks.1.attn.c_qkv.weight", "backbone.resblocks.1.attn.c_qkv.bias",
               "backbone.resblocks.1.mlp.c_fc.bias", "backbone.resblocks.1.mlp.c_proj.weight",
               "ln_post.weight", "output_proj.weight"]
    errs = {k: rel(named[k].grad, leaves[k].grad) for k in checked}
    worst = max(errs.items(), key=lambda kv: kv[1])
    assert worst[1] < 5e-2, errs


def test_trainer_close_removes_its_hooks_and_engine():
    """A second DiffusionTrainer on the same model must not stack forward / state_dict hooks on top of the first one's:
    close() removes them and the fused engine; the hooks hold the trainer strongly and a new trainer closes the attached one."""
    import gc
    import weakref
    from npcd.train import DiffusionTrainer
    a, _ = _models()
    den = a.denoiser
    n_fwd, n_sd_model, n_sd_den = len(den._forward_pre_hooks), len(a._state_dict_pre_hooks), len(den._state_dict_pre_hooks)
    t1 = DiffusionTrainer(a, fused=True)
    assert len(den._forward_pre_hooks) == n_fwd + 1 and len(a._state_dict_pre_hooks) == n_sd_model + 1
    assert len(den._state_dict_pre_hooks) == n_sd_den + 1
    t1.close()
    assert len(den._forward_pre_hooks) == n_fwd and len(a._state_dict_pre_hooks) == n_sd_model and len(den._state_dict_pre_hooks) == n_sd_den
    assert den.backbone.fused_engine is None
    t2 = DiffusionTrainer(a, fused=True)
    assert len(den._forward_pre_hooks) == n_fwd + 1 and den.backbone.fused_engine is not None
    g = torch.Generator().manual_seed(2)
    c0, f0 = torch.randn(2, 3, 48, generator=g).cuda(), torch.randn(2, 32, 48, generator=g).cuda()
    loss, _ = t2.step(c0, f0)
    assert torch.isfinite(loss)
    # a trainer whose last outside reference is dropped WITHOUT close() stays attached (the model's hooks hold it strongly, ADVICE
    # r4): its waits keep working for every later forward / state_dict(), nothing is left half-gathered
    r = weakref.ref(t2)
    del t2
    gc.collect()
    assert r() is not None and a.__dict__["_npcd_trainer"] is r()
    waits = []
    orig = r().wait_params
    r().wait_params = lambda *x, **k: (waits.append(x), orig(*x, **k))[1]
    a.state_dict()
    assert waits, "state_dict() of a model with an attached trainer must complete the parameter gathers first"
    del r().wait_params, orig
    # a third trainer on the same model detaches the second one by itself: hooks do not stack, the old one becomes collectable
    eng2 = den.backbone.fused_engine
    t3 = DiffusionTrainer(a, fused=True)
    assert len(den._forward_pre_hooks) == n_fwd + 1 and len(a._state_dict_pre_hooks) == n_sd_model + 1
    assert den.backbone.fused_engine is not eng2 and eng2.wait_range is None and a.__dict__["_npcd_trainer"] is t3
    del eng2
    gc.collect()
    assert r() is None, "a replaced trainer must not be kept alive by the model"
    loss, _ = t3.step(c0, f0)
    assert torch.isfinite(loss)
    t3.close()
    assert "_npcd_trainer" not in a.__dict__ and den.backbone.fused_engine is None


def test_model_with_attached_trainer_copies_and_pickles_without_it(tmp_path):
    """copy.deepcopy(model) (the common EMA-snapshot pattern) and torch.save(model) on a model with an attached trainer must not try to
    copy or pickle the trainer -- flat buffers, streams, ctypes handles, process groups: the copy carries no trainer, no fused
    engine and inert hooks, owns its parameters (not views of the trainer's flat buffer) and runs on the module path."""
    import copy
    from npcd.train import DiffusionTrainer
    a, _ = _models()
    tr = DiffusionTrainer(a, fused=True)
    g = torch.Generator().manual_seed(4)
    c0, f0 = torch.randn(2, 3, 48, generator=g).cuda(), torch.randn(2, 32, 48, generator=g).cuda()
    tr.step(c0, f0)
    snap = copy.deepcopy(a)
    assert snap.__dict__.get("_npcd_trainer") is None and snap.denoiser.backbone.fused_engine is None
    assert a.__dict__["_npcd_trainer"] is tr and a.denoiser.backbone.fused_engine is not None          # the original keeps its trainer
    lo, hi = tr.flat.flat.data_ptr(), tr.flat.flat.data_ptr() + tr.flat.flat.numel() * 4
    for (n, p), (_, q) in zip(snap.named_parameters(), a.named_parameters()):
        assert torch.equal(p, q) and not (lo <= p.data_ptr() < hi), n
    before = {n: p.detach().clone() for n, p in snap.named_parameters()}
    tr.step(c0, f0)                                                   # the original moves on, the snapshot does not
    assert all(torch.equal(p, before[n]) for n, p in snap.named_parameters())
    assert any(not torch.equal(p, before[n]) for n, p in a.named_parameters())
    t = torch.tensor([3, 700]).cuda()
    with torch.no_grad():                                             # the snapshot is a working model (module path, inert hooks)
        ec, ef = snap.denoiser(c0, f0, t)
    assert torch.isfinite(ec).all() and torch.isfinite(ef).all()
    snap.state_dict()
    path = str(tmp_path / "whole_model.pt")
    torch.save(a, path)
    back = torch.load(path, weights_only=False)
    assert back.__dict__.get("_npcd_trainer") is None and back.denoiser.backbone.fused_engine is None
    for (n, p), (_, q) in zip(back.named_parameters(), a.named_parameters()):
        assert torch.equal(p.cuda(), q), n
    tr.close()
