"""GPU parity: HIP attention (through the C ABI) and the denoiser vs the CPU oracle / golden vectors.

Tolerances (bf16 inputs, fp32 accumulation, P rounded to bf16 before P.V as in flash-attn):
  forward  : rel-L2 <= 1e-2, max-abs <= 2e-2 * max|ref|   against the fp32 oracle on the SAME
             bf16-rounded inputs
  backward : rel-L2 <= 2e-2
  denoiser : eps rel-L2 <= 2e-2 vs the reference's fp32 golden output (SURVEY.md §7 step 4 measured
             7e-3 for the reference's own bf16-autocast path)
"""
import math

import numpy as np
import pytest
import torch

from oracle import denoiser as od

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    """relative L2 error; falls back to an absolute bound when the reference is (numerically) zero,
    e.g. dq/dk for a single key (softmax over one element has zero gradient)."""
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-3 * max(1.0, b.numel() ** 0.5)))


def ref_attention(qkv16, heads, gout16=None):
    qkv = qkv16.float().cpu().requires_grad_(gout16 is not None)
    out = od.attention_qkvpacked(qkv, heads)
    if gout16 is None:
        return out, None
    (out * gout16.float().cpu()).sum().backward()
    return out.detach(), qkv.grad


def run_hip(qkv16, heads, gout16=None):
    from npcd.hip.attention import attention_qkvpacked
    x = qkv16.cuda().requires_grad_(gout16 is not None)
    out = attention_qkvpacked(x, heads)
    if gout16 is None:
        return out, None
    out.backward(gout16.cuda())
    return out.detach(), x.grad


def check(qkv16, heads, gout16, tag):
    out, dqkv = run_hip(qkv16, heads, gout16)
    ro, rg = ref_attention(qkv16, heads, gout16)
    assert torch.isfinite(out).all(), tag
    e = rel_l2(out, ro)
    mx = float((out.float().cpu() - ro).abs().max() / ro.abs().max())
    assert e < 1e-2 and mx < 2e-2, f"{tag}: fwd rel-L2 {e:.3e} max {mx:.3e}"
    if gout16 is not None:
        assert torch.isfinite(dqkv).all(), tag
        B, n, w3 = qkv16.shape
        d = w3 // heads // 3
        g, r = dqkv.float().cpu().view(B, n, heads, 3, d), rg.view(B, n, heads, 3, d)
        for j, name in enumerate("qkv"):
            ej = rel_l2(g[:, :, :, j], r[:, :, :, j])
            assert ej < 2e-2, f"{tag}: d{name} rel-L2 {ej:.3e}"
            # the last token on its own (sequences of k*128 + 1 tokens take its dk / dv from the dQ pass's by-product)
            el = rel_l2(g[:, -1, :, j], r[:, -1, :, j])
            assert el < 3e-2, f"{tag}: d{name} of the last token rel-L2 {el:.3e}"


@pytest.fixture(params=["auto", "32", "64"])
def fwd_form(request, monkeypatch):
    """The forward has two forms (csrc/attention.hip: 32 / 64 query rows per wave; the library picks by sequence length and reads
    NPCD_ATTN_FWD at every call): the parity tests run with the automatic choice and with each form forced."""
    if request.param == "auto":
        monkeypatch.delenv("NPCD_ATTN_FWD", raising=False)
    else:
        monkeypatch.setenv("NPCD_ATTN_FWD", request.param)
    return request.param


@pytest.mark.parametrize("tag", ["n513_h1_d64", "n130_h4_d64", "n17_h4_d32"])      # (the d = 32 fixture: csrc/attention_gen.hip, round 6)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_golden(golden, tag, dtype, fwd_form):
    g = golden("attention_" + tag)
    qkv16 = torch.from_numpy(g["qkv"]).to(dtype)
    gout16 = torch.from_numpy(g["gout"]).to(dtype)
    check(qkv16, int(g["heads"]), gout16, tag)
    # and against the reference's own fp32 output (inputs rounded to 16 bit -> looser)
    out, _ = run_hip(qkv16, int(g["heads"]))
    assert rel_l2(out, torch.from_numpy(g["out"])) < 2e-2


@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 192, 193, 256, 257, 384, 448, 513, 576, 700, 1025, 2049])
def test_attention_ragged_lengths(n, fwd_form):      # 513 = BASELINE configs[1] (512 points + time token), 2049 = configs[4] (2048 points)
    gen = torch.Generator().manual_seed(n)
    B, H = 2, 3
    qkv = (torch.randn(B, n, 3 * H * 64, generator=gen) * 1.5).bfloat16()
    gout = torch.randn(B, n, H * 64, generator=gen).bfloat16()
    check(qkv, H, gout, f"n={n}")


@pytest.mark.parametrize("n", [1, 2, 17, 31, 33, 64, 65, 127, 129, 200, 513])
@pytest.mark.parametrize("d", [32, 128])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_attention_other_head_dims_ragged_lengths(n, d, dtype):
    """Head dims 32 and 128 (VERDICT r5 missing 3: the reference's attention works for any width / heads, transformer.py:68-84): the
    kernels of csrc/attention_gen.hip -- forward, dQ pass, dK / dV pass -- against the fp32 oracle on the same 16-bit inputs, the bars of
    the d = 64 kernels, ragged lengths around the 32-row wave tile, the 64-row streamed tile and the 128-row workgroup."""
    gen = torch.Generator().manual_seed(1000 * d + n)
    B, H = 2, 3
    qkv = (torch.randn(B, n, 3 * H * d, generator=gen) * 1.5 * (64 / d) ** 0.25).to(dtype)
    gout = torch.randn(B, n, H * d, generator=gen).to(dtype)
    check(qkv, H, gout, f"d={d} n={n}")


@pytest.mark.parametrize("n", [40, 129, 513])
def test_generic_kernels_agree_with_the_specialised_d64_family(n, monkeypatch):
    """NPCD_ATTN_GEN=1 sends d = 64 through csrc/attention_gen.hip as well: the two kernel families against each other (and both against
    the oracle), forward and all three gradients; the generic kernels are bitwise reproducible like the specialised ones."""
    gen = torch.Generator().manual_seed(n)
    B, H = 2, 2
    qkv = (torch.randn(B, n, 3 * H * 64, generator=gen) * 1.5).bfloat16()
    gout = torch.randn(B, n, H * 64, generator=gen).bfloat16()
    o64, g64 = run_hip(qkv, H, gout)
    monkeypatch.setenv("NPCD_ATTN_GEN", "1")
    check(qkv, H, gout, f"generic d=64 n={n}")
    og, gg = run_hip(qkv, H, gout)
    og2, gg2 = run_hip(qkv, H, gout)
    assert torch.equal(og, og2) and torch.equal(gg, gg2)
    assert rel_l2(og, o64.float()) < 5e-3 and rel_l2(gg, g64.float()) < 1e-2


def test_forced_rescale_at_other_head_dims():
    """One key in the LAST tile dominates one query (the running maximum jumps late) at d = 32 and d = 128."""
    for d in (32, 128):
        gen = torch.Generator().manual_seed(3)
        n, H = 200, 1
        qkv = torch.randn(1, n, 3 * d, generator=gen)
        qkv[0, 5, 0:d] = 3.0 * (64 / d) ** 0.5            # query 5
        qkv[0, 190, d:2 * d] = 3.0 * (64 / d) ** 0.5      # key 190 -> score 576 / sqrt(d) * ... = 72
        check(qkv.bfloat16(), H, torch.randn(1, n, d, generator=gen).bfloat16(), f"spike d={d}")


@pytest.mark.parametrize("W,H", [(128, 4), (256, 2)])
def test_denoiser_with_other_head_dims_vs_oracle(W, H):
    """NPCDTransformer with width / heads = 32 and 128 under bf16 autocast (module path: the fused backbone engine is built for d = 64)
    against the fp32 oracle (oracle/denoiser.py, pinned to the reference by the denoiser fixtures): eps rel-L2 <= 2e-2, every parameter
    gradient <= 5e-2 -- the bars of the golden denoiser test."""
    from npcd.models.diffusion import NPCDTransformer
    F_, N, B, L = 32, 96, 2, 2
    params = od.init_params(3, F_, W, L, H, seed=7)
    g = torch.Generator().manual_seed(W)
    params["output_proj.weight"] = torch.randn(params["output_proj.weight"].shape, generator=g) * 0.02      # (zero-initialised in the reference)
    c, f = torch.randn(B, 3, N, generator=g), torch.randn(B, F_, N, generator=g)
    t = torch.tensor([17, 803])
    gc, gf = torch.randn(B, 3, N, generator=g), torch.randn(B, F_, N, generator=g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ec_r, ef_r = od.denoiser_forward(leaves, c, f, t, H)
    ((ec_r * gc).sum() + (ef_r * gf).sum()).backward()
    net = NPCDTransformer(coords_dim=3, feats_dim=F_, width=W, layers=L, heads=H)
    net.load_state_dict(params)
    net = net.cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ec, ef = net(c.cuda(), f.cuda(), t.cuda())
        loss = (ec.float() * gc.cuda()).sum() + (ef.float() * gf.cuda()).sum()
    loss.backward()
    assert rel_l2(ec, ec_r.detach()) < 2e-2 and rel_l2(ef, ef_r.detach()) < 2e-2
    worst = max((rel_l2(p.grad, leaves[k].grad), k) for k, p in net.named_parameters() if float(leaves[k].grad.abs().max()) > 1e-3)
    assert worst[0] < 5e-2, worst
    # the sampler's precision: fp32, no autocast, no gradients (diffusion_model.py:108-133) -- the exact fp32 attention forward at this head dim
    with torch.no_grad():
        ec32, ef32 = net(c.cuda(), f.cuda(), t.cuda())
    assert ec32.dtype == torch.float32
    assert rel_l2(ec32, ec_r.detach()) < 1e-4 and rel_l2(ef32, ef_r.detach()) < 1e-4


def test_attention_forced_rescale(fwd_form):
    """One key (in the LAST tile) dominates one query: the running max must jump late."""
    gen = torch.Generator().manual_seed(3)
    n, H = 200, 1
    qkv = torch.randn(1, n, 192, generator=gen)
    qkv[0, 5, 0:64] = 3.0          # query 5
    qkv[0, 190, 64:128] = 3.0      # key 190 -> score 576/8 = 72
    check(qkv.bfloat16(), H, torch.randn(1, n, 64, generator=gen).bfloat16(), "spike")


@pytest.mark.parametrize("n", [129, 513])
def test_attention_edge_token_extremes(n):
    """Sequences of 128 j + 1 tokens: the last token never enters the key / query stream -- it seeds the online softmax and the
    gradient accumulators, and its own three gradient rows are summed from per-wave partials (csrc/attention.hip, 'the edge
    token').  Here it DOMINATES: as a key it wins every row's softmax, as a query it is peaked on one early key; the gradients must
    also be bitwise reproducible (fixed summation order)."""
    gen = torch.Generator().manual_seed(100 + n)
    H = 2
    qkv = torch.randn(2, n, 3 * H * 64, generator=gen)
    q, k = qkv.view(2, n, H, 3, 64)[:, :, :, 0], qkv.view(2, n, H, 3, 64)[:, :, :, 1]
    k[:, -1] = 0.35 * q.mean(dim=1) + 0.9          # the last key: scores well above the rest for most rows
    q[:, -1, 0] = 4.0 * k[:, 7, 0]                 # the last query, head 0: peaked on key 7
    gout = torch.randn(2, n, H * 64, generator=gen).bfloat16()
    check(qkv.bfloat16(), H, gout, f"edge-dominant n={n}")
    a = run_hip(qkv.bfloat16(), H, gout)[1]
    b = run_hip(qkv.bfloat16(), H, gout)[1]
    assert torch.equal(a, b), "attention backward is not bitwise reproducible"


@pytest.mark.parametrize("n", [129, 200, 513])
def test_backward_passes_launched_separately_match_the_single_call(n, monkeypatch):
    """npcd_attn_bwd_pass(1) then (2) -- what bench.py times -- against npcd_attn_bwd: the same kernels, bitwise equal gradients
    (for 128 j + 1 tokens pass 2 also launches the kernel that finishes the last token's three rows from both passes' partials)."""
    from npcd.hip import attention as A
    gen = torch.Generator().manual_seed(n)
    qkv = torch.randn(2, n, 3 * 2 * 64, generator=gen).bfloat16()
    gout = torch.randn(2, n, 2 * 64, generator=gen).bfloat16()
    one = run_hip(qkv, 2, gout)[1]
    monkeypatch.setattr(A, "KERNEL_EVENTS", {t: [] for t in A.KERNEL_TAGS})
    two = run_hip(qkv, 2, gout)[1]
    assert len(A.KERNEL_EVENTS["dq"]) == 1 and len(A.KERNEL_EVENTS["dkdv"]) == 1
    assert torch.equal(one, two)


@pytest.mark.parametrize("n", [40, 65, 129, 193, 200, 513])
def test_backward_column_sums_by_product(n):
    """npcd_attn_bwd_colsum: the column sums of the packed dqkv (the c_qkv bias gradient) from the backward's own row stores --
    against the sum of the stored bf16 gradient (fp32 accumulation; order differs: 1e-5 of the column's absolute sum) and with
    gradients bitwise equal to the plain call; ragged blocks (waves without rows) and the edge token's rows included."""
    from npcd.hip import attention as A
    from npcd.hip import elementwise as ew
    B, H, d = 3, 2, 64
    gen = torch.Generator().manual_seed(n)
    qkv = torch.randn(B, n, H, 3 * d, generator=gen).bfloat16().cuda()
    dout = torch.randn(B, n, H, d, generator=gen).bfloat16().cuda()
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    out, lse = A._fwd(q, k, v, 0.125)
    g0, g1 = torch.zeros_like(qkv), torch.zeros_like(qkv)
    A._bwd(q, k, v, out, dout, lse, g0[..., :d], g0[..., d:2 * d], g0[..., 2 * d:], 0.125)
    part, rows = A.colsum_part_for(g1[..., :d])
    part.fill_(float("nan"))                         # every partial row must be written
    A._bwd(q, k, v, out, dout, lse, g1[..., :d], g1[..., d:2 * d], g1[..., 2 * d:], 0.125, colsum_part=part)
    assert torch.equal(g0, g1)
    got = torch.empty(3 * H * d, device="cuda")
    ew._finish(None, part, rows, 3 * H * d, got)
    flat = g1.float().view(B * n, 3 * H * d)
    want, scale_ = flat.sum(0), flat.abs().sum(0)
    assert torch.isfinite(got).all()
    assert float(((got - want).abs() / scale_.clamp_min(1e-6)).max()) < 1e-5


@pytest.mark.parametrize("n,B", [(128, 2), (513, 2), (2049, 1)])
def test_attention_fp8_forward(n, B, monkeypatch):
    """Opt-in forward with e4m3 operands on the block-scaled matrix instruction (BASELINE configs[4]: 'fp8 MFMA attention';
    NPCD_ATTN_FP8=1).  Bars (e4m3 has 3 mantissa bits; q, k, v and P are all rounded to it): output rel-L2 <= 8e-2 and LSE within
    8e-2 of the fp32 oracle on the same bf16 inputs -- measured 5.3e-2 / 3.3e-2 --; gradients (bf16 backward kernels against the fp8
    forward's LSE, as in FlashAttention-3) rel-L2 <= 1.2e-1.  A length the fp8 kernel does not cover takes the bf16 kernel."""
    from npcd.hip import attention as A
    monkeypatch.setattr(A, "FWD_FP8", True)
    H = 3
    gen = torch.Generator().manual_seed(n)
    qkv = torch.randn(B, n, 3 * H * 64, generator=gen).bfloat16()
    gout = torch.randn(B, n, H * 64, generator=gen).bfloat16()
    assert A.lib().npcd_attn_fwd_fp8_workspace_bytes(B, n, H) == B * H * 64 * 2 * (n - (n & 1))
    out, dqkv = run_hip(qkv, H, gout)
    ro, rg = ref_attention(qkv, H, gout)
    assert torch.isfinite(out).all() and torch.isfinite(dqkv).all()
    e = rel_l2(out, ro)
    assert 5e-3 < e < 8e-2, f"fp8 forward rel-L2 {e:.3e} (below 5e-3 the bf16 kernel ran instead)"
    g, r = dqkv.float().cpu().view(B, n, H, 3, 64), rg.view(B, n, H, 3, 64)
    for j, name in enumerate("qkv"):
        ej = rel_l2(g[:, :, :, j], r[:, :, :, j])
        assert ej < 1.2e-1, f"fp8 forward, d{name} rel-L2 {ej:.3e}"
    q4 = qkv.cuda().view(B, n, H, 3, 64)
    _, lse = A._fwd(q4[:, :, :, 0], q4[:, :, :, 1], q4[:, :, :, 2], 0.125)
    s = torch.einsum("bnhd,bmhd->bhnm", q4[:, :, :, 0].float(), q4[:, :, :, 1].float()) * 0.125
    assert float((lse - torch.logsumexp(s, -1)).abs().max()) < 8e-2
    # not covered (n = 100): same entry point, bf16 kernel, bf16 accuracy
    qkv2 = torch.randn(1, 100, 3 * H * 64, generator=gen).bfloat16()
    assert A.lib().npcd_attn_fwd_fp8_workspace_bytes(1, 100, H) == 0
    assert rel_l2(run_hip(qkv2, H)[0], ref_attention(qkv2, H)[0]) < 1e-2


def test_flash_attn_func_dropin_strided():
    """The reference's call pattern: q,k,v = split(view(B,n,H,3d)) -> flash_attn_func (transformer.py:71-75)."""
    from flash_attn import flash_attn_func
    gen = torch.Generator().manual_seed(9)
    B, n, H, d = 2, 77, 4, 64
    qkv = torch.randn(B, n, 3 * H * d, generator=gen).bfloat16()
    x = qkv.cuda().requires_grad_(True)
    q, k, v = torch.split(x.view(B, n, H, -1), d, dim=-1)
    assert not q.is_contiguous()
    out = flash_attn_func(q, k, v, causal=False, dropout_p=0)
    gout = torch.randn(B, n, H * d, generator=gen).bfloat16()
    out.reshape(B, n, -1).backward(gout.cuda())
    ro, rg = ref_attention(qkv, H, gout)
    assert rel_l2(out.reshape(B, n, -1), ro) < 1e-2
    assert rel_l2(x.grad, rg) < 2e-2
    with pytest.raises(NotImplementedError):
        flash_attn_func(q, k, v, causal=True)


def test_attention_full_size_properties():
    """BASELINE cfg 2 size (B=64, n=513, H=16): size-independent properties instead of an oracle run."""
    from npcd.hip.attention import attention_qkvpacked
    B, n, H, d = 64, 513, 16, 64
    gen = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, n, 3 * H * d, device="cuda", generator=gen).bfloat16()
    x = qkv.view(B, n, H, 3, d)
    # (1) rows of softmax sum to one: V == 1 -> out == 1
    x1 = x.clone(); x1[:, :, :, 2] = 1.0
    out = attention_qkvpacked(x1.view(B, n, -1), H)
    assert float((out.float() - 1).abs().max()) < 1e-2
    # (2) permutation equivariance over the key/value points: permuting k and v rows together leaves out unchanged
    perm = torch.randperm(n, device="cuda", generator=gen)
    xp = x.clone(); xp[:, :, :, 1:] = x[:, perm][:, :, :, 1:]
    o0 = attention_qkvpacked(qkv, H)
    o1 = attention_qkvpacked(xp.view(B, n, -1), H)
    assert rel_l2(o1, o0) < 6e-3
    # (3) linearity in V
    xa = x.clone(); xa[:, :, :, 2] = x[:, :, :, 2] * 2
    o2 = attention_qkvpacked(xa.view(B, n, -1), H)
    assert rel_l2(o2, o0.float() * 2) < 6e-3
    # (4) one sample against the oracle
    ro, _ = ref_attention(qkv[:1].cpu(), H)
    assert rel_l2(o0[:1], ro) < 1e-2


def test_unsupported_shapes_fail_loudly():
    from npcd.hip.attention import attention_qkvpacked
    with pytest.raises(RuntimeError, match="unsupported"):
        attention_qkvpacked(torch.zeros(1, 8, 3 * 2 * 16, device="cuda", dtype=torch.bfloat16), 2)   # d = 16 (32, 64, 128 are built)
    with pytest.raises(RuntimeError, match="head dim"):
        attention_qkvpacked(torch.zeros(1, 8, 3 * 2 * 16, device="cuda", dtype=torch.float32), 2)    # fp32: 32, 64, 128 as well
    with pytest.raises(RuntimeError, match="WITH GRADIENTS"):                                        # fp32 TRAINING at other head dims: not built
        attention_qkvpacked(torch.zeros(1, 8, 3 * 2 * 32, device="cuda", dtype=torch.float32, requires_grad=True), 2)
    with pytest.raises(RuntimeError, match="supports"):
        attention_qkvpacked(torch.zeros(1, 8, 192, device="cuda", dtype=torch.float64), 1)


@pytest.mark.parametrize("name", ["attention_n130_h4_d64", "attention_n513_h1_d64"])
def test_attention_fp32_training_matches_reference_golden(golden, name):
    """`--dtype float32` training: exact-fp32 HIP forward, gradients from the fp32 matrix-instruction backward kernels
    (attn_bwd_f32_dq_kernel / attn_bwd_f32_dkdv_kernel), against the reference's own fp32 attention output and qkv gradient
    (fixture G1)."""
    from npcd.hip.attention import attention_qkvpacked
    g = golden(name)
    H = int(g["heads"])
    qkv = torch.from_numpy(g["qkv"]).cuda().requires_grad_(True)
    out = attention_qkvpacked(qkv, H)
    (out * torch.from_numpy(g["gout"]).cuda()).sum().backward()
    ref_o, ref_g = torch.from_numpy(g["out"]), torch.from_numpy(g["dqkv"])
    assert float((out.detach().cpu() - ref_o).abs().max()) < 2e-5 * max(1.0, float(ref_o.abs().max()))
    assert float((qkv.grad.cpu() - ref_g).abs().max()) < 2e-5 * max(1.0, float(ref_g.abs().max()))


@pytest.mark.parametrize("n,H,B", [(1, 1, 1), (31, 2, 2), (32, 1, 1), (33, 3, 1), (127, 2, 1), (129, 1, 2), (257, 2, 1), (513, 2, 2)])
def test_attention_fp32_backward_kernel_ragged_lengths(n, H, B):
    """The fp32 backward kernels on lengths around their 32-row / 128-row tile edges against float64 autograd of the oracle's
    einsum attention (transformer.py:76-83), 2e-5 relative to the largest gradient entry; q / k / v are strided views of the
    packed projection (no copies) and a second call gives the same bits (no atomics)."""
    from npcd.hip.attention import attention_qkvpacked
    gen = torch.Generator().manual_seed(1000 + n)
    qkv = torch.randn(B, n, 3 * H * 64, generator=gen) * 1.2
    gout = torch.randn(B, n, H * 64, generator=gen)
    ref_in = qkv.double().requires_grad_(True)
    (od.attention_qkvpacked(ref_in, H) * gout.double()).sum().backward()
    grads = []
    for _ in range(2):
        x = qkv.cuda().requires_grad_(True)
        (attention_qkvpacked(x, H) * gout.cuda()).sum().backward()
        grads.append(x.grad)
    ref_g = ref_in.grad.float()
    assert float((grads[0].cpu() - ref_g).abs().max()) < 2e-5 * max(1.0, float(ref_g.abs().max()))
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("tag", ["f32_w64", "f128_w64", "f32_w128_h2"])
def test_denoiser_matches_reference_golden(golden, tag):
    from npcd.models.diffusion import NPCDTransformer
    g = golden("denoiser_" + tag)
    T = torch.from_numpy
    F_ = g["feats"].shape[1]
    net = NPCDTransformer(coords_dim=3, feats_dim=F_, width=int(g.get("width", 64)), layers=int(g.get("layers", 2 if tag == "f32_w64" else 1)),
                          heads=int(g["heads"]))
    net.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w:")})
    net = net.cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ec, ef = net(T(g["coords"]).cuda(), T(g["feats"]).cuda(), T(g["t"]).cuda())
        loss = (ec.float() * T(g["gc"]).cuda()).sum() + (ef.float() * T(g["gf"]).cuda()).sum()
    loss.backward()
    assert rel_l2(ec, T(g["eps_coords"])) < 2e-2 and rel_l2(ef, T(g["eps_feats"])) < 2e-2
    worst = 0.0
    for k, v in g.items():
        if k.startswith("g:") and np.abs(v).max() > 1e-3:
            p = dict(net.named_parameters())[k[2:]]
            worst = max(worst, rel_l2(p.grad, T(v)))
    assert worst < 5e-2, f"worst param-grad rel-L2 {worst:.3e}"


@pytest.mark.parametrize("tag", ["f32_w64", "f128_w64", "f32_w128_h2"])
def test_denoiser_fp32_training_matches_reference_golden(golden, tag):
    """`--dtype float32` (train_diffusion.py:78 choice, no autocast): the whole denoiser forward + backward in fp32 with the fp32
    attention kernels (forward, dq, dk/dv on the fp32 matrix instruction) against the reference's own fp32 outputs and parameter
    gradients: 1e-4 relative (the bf16-autocast form of this test holds 2e-2 / 5e-2)."""
    from npcd.models.diffusion import NPCDTransformer
    g = golden("denoiser_" + tag)
    T = torch.from_numpy
    F_ = g["feats"].shape[1]
    net = NPCDTransformer(coords_dim=3, feats_dim=F_, width=int(g.get("width", 64)), layers=int(g.get("layers", 2 if tag == "f32_w64" else 1)),
                          heads=int(g["heads"]))
    net.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("w:")})
    net = net.cuda()
    ec, ef = net(T(g["coords"]).cuda(), T(g["feats"]).cuda(), T(g["t"]).cuda())
    assert ec.dtype == torch.float32
    ((ec * T(g["gc"]).cuda()).sum() + (ef * T(g["gf"]).cuda()).sum()).backward()
    assert rel_l2(ec, T(g["eps_coords"])) < 1e-4 and rel_l2(ef, T(g["eps_feats"])) < 1e-4
    worst = 0.0
    for k, v in g.items():
        if k.startswith("g:") and np.abs(v).max() > 1e-3:
            worst = max(worst, rel_l2(dict(net.named_parameters())[k[2:]].grad, T(v)))
    assert worst < 1e-4, f"worst param-grad rel-L2 {worst:.3e}"


@pytest.mark.parametrize("d", [32, 128])
@pytest.mark.parametrize("n,H", [(1, 1), (17, 2), (65, 1), (130, 3), (513, 2)])
def test_attention_fp32_inference_other_head_dims(n, H, d):
    """The exact fp32 forward (the reference samples in fp32, diffusion_model.py:108-133) at head dims 32 and 128 (round 6: the vector-ALU
    kernel templated over the head dim): against float64 softmax attention on the same inputs, 2e-6 relative."""
    from npcd.hip.attention import attention_qkvpacked
    g = torch.Generator().manual_seed(100 * d + n)
    qkv = torch.randn(2, n, 3 * H * d, generator=g) * 1.2
    with torch.no_grad():
        out = attention_qkvpacked(qkv.cuda(), H)
    x = qkv.double().view(2, n, H, 3 * d)
    q, k, v = (x[..., i * d:(i + 1) * d].permute(0, 2, 1, 3) for i in range(3))
    ref = (torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), -1) @ v).permute(0, 2, 1, 3).reshape(2, n, H * d)
    assert out.dtype == torch.float32
    err = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 2e-6, err


@pytest.mark.parametrize("n,H", [(1, 1), (17, 2), (64, 1), (130, 3), (513, 2)])
def test_attention_fp32_inference_exact(n, H):
    """fp32 sampling path (the reference runs generate() with the fp32 einsum attention): exact fp32 math."""
    from flash_attn import flash_attn_func
    from npcd.hip.attention import attention_qkvpacked
    gen = torch.Generator().manual_seed(n)
    qkv = torch.randn(2, n, 3 * H * 64, generator=gen) * 1.5
    ref = od.attention_qkvpacked(qkv, H)
    with torch.no_grad():
        out = attention_qkvpacked(qkv.cuda(), H)
        q, k, v = torch.split(qkv.cuda().view(2, n, H, -1), 64, dim=-1)
        out2 = flash_attn_func(q, k, v)
    assert out.dtype == torch.float32
    assert float((out.cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    assert torch.equal(out2.reshape(2, n, -1), out)


def test_generate_reverse_step_and_sampler(golden):
    """DiffusionModel.generate (diffusion_model.py:108-133): fp32, einsum-equivalent attention, x0 clipping.
    One reverse step is checked against the oracle with the SAME noise; the full loop for shape / finiteness."""
    from oracle import diffusion as odf
    from npcd.models.diffusion import DiffusionModel
    torch.manual_seed(0)
    F_, N, W, L, H = 32, 48, 128, 2, 2
    params = od.init_params(3, F_, W, L, H, seed=4)
    model = DiffusionModel(3, F_, N, W, L, H, False)
    model.denoiser.load_state_dict(params)
    model = model.cuda().eval()
    gd = model.diffusion_process
    g = torch.Generator().manual_seed(1)
    c_t, f_t = torch.randn(2, 3, N, generator=g), torch.randn(2, F_, N, generator=g)
    t = torch.tensor([999, 3])
    clip_c, clip_f = (torch.tensor([-2.0]), torch.tensor([2.5])), (torch.tensor([-1.0]), torch.tensor([1.0]))
    torch.manual_seed(77)
    nc, nf = torch.randn(2, 3, N, device="cuda"), torch.randn(2, F_, N, device="cuda")     # what p_sample will draw
    torch.manual_seed(77)
    with torch.no_grad():
        c1, c0, f1, f0 = gd.p_sample(model.denoiser, c_t.cuda(), f_t.cuda(), t.cuda(), tuple(x.cuda() for x in clip_c),
                                     tuple(x.cuda() for x in clip_f))
    ec, ef = od.denoiser_forward(params, c_t, f_t, t, H)
    tab = odf.schedule_tables()
    rc1, rc0 = odf.p_sample_step(tab, c_t, ec, t, nc.cpu(), clip_c)
    rf1, rf0 = odf.p_sample_step(tab, f_t, ef, t, nf.cpu(), clip_f)
    for a, b in ((c1, rc1), (c0, rc0), (f1, rf1), (f0, rf0)):
        assert float((a.cpu() - b).abs().max()) < 2e-4
    # the fused posterior update (one kernel per tensor) draws the same noise in the same order: same step
    torch.manual_seed(77)
    with torch.no_grad():
        fc1, ff1 = gd.p_sample_fused(model.denoiser, c_t.cuda(), f_t.cuda(), t.cuda(), (-2.0, 2.5), (-1.0, 1.0))
    assert float((fc1 - c1).abs().max()) < 1e-5 and float((ff1 - f1).abs().max()) < 1e-5
    # full sampling loop on a shortened chain
    gd.num_timesteps = 12
    model.coords_normalization.min.fill_(-3); model.coords_normalization.max.fill_(3)
    model.feats_normalization.min.fill_(-1); model.feats_normalization.max.fill_(1)
    coords, feats = model.generate(3, batch_size=2, progress=False)
    assert len(coords) == 3 and coords[0].shape == (3, N) and feats[0].shape == (F_, N)
    assert all(torch.isfinite(x).all() for x in coords + feats)
    # same seed: the fused loop reproduces the step-by-step reference formulation
    torch.manual_seed(5)
    a_c, a_f = model.generate(2, batch_size=2, progress=False)
    torch.manual_seed(5)
    c = torch.randn(2, 3, N, device="cuda"); f = torch.randn(2, F_, N, device="cuda")
    with torch.no_grad():
        for i in range(11, -1, -1):
            tt = torch.full((2,), i, device="cuda", dtype=torch.long)
            c, _, f, _ = gd.p_sample(model.denoiser, c, f, tt, (model.coords_normalization.min, model.coords_normalization.max),
                                     (model.feats_normalization.min, model.feats_normalization.max))
    assert float((torch.stack(a_c) - model.coords_normalization(c)).abs().max()) < 1e-4
    assert float((torch.stack(a_f) - model.feats_normalization(f)).abs().max()) < 1e-4
    # bf16 autocast + HIP-graph replay of the reverse step
    coords, feats = model.generate(4, batch_size=4, progress=False, dtype=torch.bfloat16, use_graph=True)
    assert len(coords) == 4 and all(torch.isfinite(x).all() for x in coords + feats)
    assert float(torch.stack(coords).abs().max()) <= 3.0 + 1e-5          # x0 clipping was applied on the last step
    with pytest.raises(AssertionError):
        model.train().generate(1)


def test_attention_is_bitwise_reproducible():
    """No atomics and fixed summation orders in all three kernels: two runs give identical bits (full BASELINE cfg 2 size)."""
    from npcd.hip.attention import attention_qkvpacked
    B, n, H, d = 64, 513, 16, 64
    gen = torch.Generator(device="cuda").manual_seed(7)
    qkv = torch.randn(B, n, 3 * H * d, device="cuda", generator=gen).bfloat16()
    gout = torch.randn(B, n, H * d, device="cuda", generator=gen).bfloat16()
    res = []
    for _ in range(2):
        x = qkv.clone().requires_grad_(True)
        out = attention_qkvpacked(x, H)
        out.backward(gout)
        res.append((out.detach().clone(), x.grad.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("n", [1, 33, 64, 130, 256, 257, 448, 513, 576, 1025, 2049])
def test_single_pass_backward_matches_oracle_and_two_pass(n, monkeypatch):
    """The opt-in single-pass backward (NPCD_ATTN_BWD=fused, attn_bwd_fused_kernel: five matrix products, dS through LDS, dQ summed
    inside the matrix instruction over 256-key passes with an fp32 running sum between passes) against the fp32 oracle with the bars
    of the default two-pass kernels, and against those kernels themselves (different summation orders: <= 1.5e-2 apart); bitwise
    reproducible."""
    from npcd.hip import attention as hattn
    gen = torch.Generator().manual_seed(1000 + n)
    B, H = 2, 3
    qkv = (torch.randn(B, n, 3 * H * 64, generator=gen) * 1.5).bfloat16()
    gout = torch.randn(B, n, H * 64, generator=gen).bfloat16()

    def grads(mode):
        monkeypatch.setattr(hattn, "BWD_MODE", mode)
        x = qkv.cuda().requires_grad_(True)
        out = hattn.attention_qkvpacked(x, H)
        out.backward(gout.cuda())
        return x.grad.clone()

    g_two, g_one, g_again = grads("twopass"), grads("fused"), grads("fused")
    assert torch.equal(g_one, g_again)
    assert rel_l2(g_one, g_two.float().cpu()) < 1.5e-2
    monkeypatch.setattr(hattn, "BWD_MODE", "fused")
    check(qkv, H, gout, f"fused n={n}")


def test_opt_in_forward_without_the_fifth_workgroup_still_matches():
    """NPCD_ATTN_ROWX32=1 (round 4, opt-in because it measured slower): the 32-row forward of 256 j + 1-token sequences without a workgroup
    for the single last query row -- eight waves of the (batch, head) split that row's keys, a merge kernel combines them.  The switch
    is read once per process, so the golden + ragged-length tests run again in a child process with it set."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, NPCD_ATTN_ROWX32="1")
    out = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-k", "golden or ragged"], capture_output=True, text=True,
                         env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-1000:]


def test_fp32_class_sampler_forward_matches_the_fp32_path():
    """DiffusionModel.generate(dtype="fp32_class") (round 5): the reference samples in fp32 (diffusion_model.py:108-133); here the backbone's
    Linear layers run as ONE bf16 library GEMM each over the three cross products of split operands (csrc/split.hip + fused.backbone_forward_x2),
    everything else -- residual stream, LayerNorm, the fp32 matrix-instruction attention, exact-erf GELU -- in fp32.  Against the plain fp32 module
    path on the same weights: the eps prediction to rel-L2 <= 2e-5 (bf16 autocast: ~5e-3) at the tiny golden width AND at the benchmark's
    width (1024 / 16 heads, 2 blocks, sequence 513); against the float32 CPU oracle like the fp32 path; one sampling loop reproduces the fp32 loop
    from the same seed to 1e-3; the split kernel itself against its definition."""
    from npcd.hip import elementwise as ew
    from npcd.models.diffusion import DiffusionModel
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    # the split kernel: [hi | lo | hi] of (x + bias) and of gelu(x + bias)
    x = torch.randn(37, 256, device="cuda") * 3
    b = torch.randn(256, device="cuda")
    for gelu in (False, True):
        y = x + b
        if gelu:
            y = torch.nn.functional.gelu(y)
        got = ew.split3(x, b, gelu)
        hi = y.bfloat16()
        lo = (y - hi.float()).bfloat16()
        assert got.shape == (37, 768)
        assert rel(got[:, :256].float() + got[:, 256:512].float(), y) < 1e-5 and torch.equal(got[:, :256], got[:, 512:])
        assert float((got[:, :256].float() - hi.float()).abs().max()) <= 1e-2 * float(y.abs().max()) and rel(got[:, 256:512].float(), lo.float()) < 2e-2
    # the LayerNorm hand-over kernel: xnew = x + o + bias (bit-exact in that order), [hi | lo | hi] of LayerNorm(xnew); widths it does not take -> None
    for W_, T_ in ((256, 5), (1024, 1027), (768, 64)):
        x, o = torch.randn(T_, W_, device="cuda") * 2 + 0.5, torch.randn(T_, W_, device="cuda")
        b, g, be = torch.randn(W_, device="cuda"), torch.rand(W_, device="cuda") + 0.5, torch.randn(W_, device="cuda")
        for fused_add in (False, True):
            xn, got = ew.add_ln_split3(x, g, be, o, b) if fused_add else ew.add_ln_split3(x, g, be)
            want_x = x + (o + b) if fused_add else x
            assert torch.equal(xn, want_x)
            y = torch.nn.functional.layer_norm(want_x.double(), (W_,), g.double(), be.double())
            assert got.shape == (T_, 3 * W_) and torch.equal(got[:, :W_], got[:, 2 * W_:])
            assert rel(got[:, :W_].float() + got[:, W_:2 * W_].float(), y) < 1e-5
    assert ew.add_ln_split3(torch.randn(4, 128, device="cuda"), torch.ones(128, device="cuda"), torch.zeros(128, device="cuda")) is None
    for (F_, N, W, L, H, B) in ((32, 48, 128, 2, 2, 3), (128, 512, 1024, 2, 16, 2)):
        torch.manual_seed(W)
        params = od.init_params(3, F_, W, L, H, seed=4)
        model = DiffusionModel(3, F_, N, W, L, H, False)
        model.denoiser.load_state_dict(params)
        with torch.no_grad():
            model.denoiser.output_proj.weight.normal_(0, 0.05)           # (zero-initialised in the reference: eps would not see the backbone)
        model = model.cuda().eval()
        c = torch.randn(B, 3, N, device="cuda")
        f = torch.randn(B, F_, N, device="cuda")
        t = torch.randint(0, 1000, (B,), device="cuda")
        bb = model.denoiser.backbone
        with torch.no_grad():
            e32 = model.denoiser(c, f, t)
            bb.fp32_class = True
            ex2 = model.denoiser(c, f, t)
            bb.fp32_class = False
            with torch.autocast("cuda", dtype=torch.bfloat16):
                e16 = model.denoiser(c, f, t)
        assert bb._infer_weights_x2 is not None, "the split-operand path was not taken"
        r_c, r_f = rel(ex2[0], e32[0]), rel(ex2[1], e32[1])
        print(f"fp32-class denoiser forward W {W}: eps rel-L2 vs the fp32 path {r_c:.1e} / {r_f:.1e}; bf16 autocast {rel(e16[0].float(), e32[0]):.1e}")
        assert r_c < 2e-5 and r_f < 2e-5 and rel(e16[0].float(), e32[0]) > 1e-4
        if W == 128:
            sd = {k: v.cpu() for k, v in model.denoiser.state_dict().items()}
            oc, of = od.denoiser_forward(sd, c.cpu(), f.cpu(), t.cpu(), H)
            assert rel(ex2[0].cpu(), oc) < 1e-4 and rel(ex2[1].cpu(), of) < 1e-4
            # a shortened sampling loop: same seed, fp32 against fp32-class
            model.diffusion_process.num_timesteps = 10
            model.coords_normalization.min.fill_(-3); model.coords_normalization.max.fill_(3)
            model.feats_normalization.min.fill_(-1); model.feats_normalization.max.fill_(1)
            torch.manual_seed(3)
            a_c, a_f = model.generate(2, batch_size=2, progress=False)
            torch.manual_seed(3)
            b_c, b_f = model.generate(2, batch_size=2, progress=False, dtype="fp32_class")
            assert bb.fp32_class is False
            assert float((torch.stack(a_c) - torch.stack(b_c)).abs().max()) < 1e-3 and float((torch.stack(a_f) - torch.stack(b_f)).abs().max()) < 1e-3
            with pytest.raises(ValueError):
                model.generate(1, batch_size=1, progress=False, dtype="fp64")


def test_opt_in_resident_forward_still_matches():
    """NPCD_ATTN_FWD=res (round 5, opt-in because it measured slower: 141 against 120 us in the step): the forward of 513-token sequences
    with K / V of a (batch, head) resident in LDS -- one workgroup of eight waves per (batch, head), the 513th query row split over the
    waves and merged through LDS.  The golden, ragged-length (513 is among them), forced-rescale and reproducibility tests run again
    in a child process with the switch set; other lengths fall through to the ring kernels."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, NPCD_ATTN_FWD="res")
    out = subprocess.run([sys.executable, "-m", "pytest", __file__, "-q", "-x", "-k", "golden or ragged or rescale or reproduc or edge_token"],
                         capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-1000:]
