"""GPU parity of the own NT GEMMs with fused epilogues (csrc/gemm_nt.hip) against plain PyTorch fp32 references on the same
16-bit operands: y = x W^T + b, (h, gelu(h)), and the data gradient of the layer behind a GELU fused with the GELU backward and
the bias-gradient column sums.  Reference semantics: nn.Linear / nn.GELU() (exact erf) of transformer.py:118-137 under autocast
(16-bit GEMM outputs, fp32 accumulation)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-12))


# ragged M (not a multiple of 256, of 32, of 8), one tile, several rounds of the persistent grid, both K extremes
SHAPES = [(513, 1024, 1024), (1, 1024, 64), (255, 1024, 128), (257, 2048, 192), (4104, 1024, 1024), (2052, 3072, 1024),
          (1026, 4096, 1024), (1539, 1024, 4096)]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_fwd_matches_fp32_reference(M, N, K, dtype):
    from npcd.hip import linear as hl
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 1.5).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).cuda()
    b = (torch.randn(N, generator=g) * 0.5).to(dtype).cuda()
    ref = x.float() @ w.float().t() + b.float()
    y = hl.linear_fwd(x, w, b)
    assert y.dtype == dtype and torch.isfinite(y).all()
    assert rel(y, ref) < (4e-3 if dtype == torch.bfloat16 else 6e-4)          # one rounding of the output
    # elementwise: within one 16-bit ulp of the fp32 result
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    assert float(((y.float() - ref).abs() / ref.abs().clamp_min(1.0)).max()) < 1.01 * ulp
    y0 = hl.linear_fwd(x, w, None)
    assert rel(y0, x.float() @ w.float().t()) < (4e-3 if dtype == torch.bfloat16 else 6e-4)
    # bitwise reproducible
    assert torch.equal(y, hl.linear_fwd(x, w, b))


# the 128 x 128 form: the token counts of one rank of the 8- / 4- / 2-GPU job (left-over rows 8 / 16 / 32 ride on the last row tile),
# one row, exactly one tile, a ragged end above 32 rows (an own row tile), N of one tile, both K extremes
SHAPES128 = [(4104, 1024, 1024), (8208, 1024, 1024), (4104, 1024, 4096), (4104, 1024, 3072), (16416, 1024, 1024), (1, 128, 64), (128, 128, 64),
             (129, 256, 128), (160, 128, 192), (161, 384, 64), (300, 1024, 256), (4104, 4096, 1024)]


@pytest.mark.parametrize("M,N,K", SHAPES128)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear128_fwd_matches_fp32_reference(M, N, K, dtype):
    from npcd.hip import linear as hl
    g = torch.Generator().manual_seed(M + N + K + 1)
    x = (torch.randn(M, K, generator=g) * 1.5).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).cuda()
    b = (torch.randn(N, generator=g) * 0.5).to(dtype).cuda()
    ref = x.float() @ w.float().t() + b.float()
    y = torch.full((M + 1, N), 7.0, dtype=dtype, device="cuda")             # (+ a guard row behind the output)
    hl.linear128_fwd(x, w, b, out=y[:M])
    assert torch.isfinite(y).all() and bool((y[M] == 7.0).all())
    assert rel(y[:M], ref) < (4e-3 if dtype == torch.bfloat16 else 6e-4)
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    assert float(((y[:M].float() - ref).abs() / ref.abs().clamp_min(1.0)).max()) < 1.01 * ulp
    assert rel(hl.linear128_fwd(x, w, None), x.float() @ w.float().t()) < (4e-3 if dtype == torch.bfloat16 else 6e-4)
    assert torch.equal(y[:M], hl.linear128_fwd(x, w, b))                    # bitwise reproducible


def test_linear128_catches_a_transposed_or_permuted_output():
    from npcd.hip import linear as hl
    M, N, K = 300, 1024, 256
    x = torch.zeros(M, K)
    for m in range(M):
        x[m, (7 * m + 3) % K] = 1.0
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125) / 16.0           # exact in bf16
    y = hl.linear128_fwd(x.bfloat16().cuda(), w.bfloat16().cuda(), None)
    assert torch.equal(y.float().cpu(), (x @ w.t()).bfloat16().float())


def test_linear_catches_a_transposed_or_permuted_output():
    """A = I-style check with ASYMMETRIC operands (a swapped row / column map or a wrong column run order would pass a random
    tolerance test only by luck): x selects single rows of w."""
    from npcd.hip import linear as hl
    M, N, K = 300, 1024, 256
    x = torch.zeros(M, K)
    for m in range(M):
        x[m, (7 * m + 3) % K] = 1.0
    w = (torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125) / 16.0           # exact in bf16
    y = hl.linear_fwd(x.bfloat16().cuda(), w.bfloat16().cuda(), None)
    ref = x @ w.t()
    assert torch.equal(y.float().cpu(), ref.bfloat16().float())


@pytest.mark.parametrize("M,N,K", [(513, 4096, 1024), (1030, 1024, 256)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_gelu_fwd(M, N, K, dtype):
    from npcd.hip import linear as hl
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * 2).to(dtype).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dtype).cuda()
    b = (torch.randn(N, generator=g) * 0.5).to(dtype).cuda()
    h, gl = hl.linear_gelu_fwd(x, w, b)
    h_ref = (x.float() @ w.float().t() + b.float())
    assert rel(h, h_ref) < (4e-3 if dtype == torch.bfloat16 else 6e-4)
    assert torch.equal(h, hl.linear_fwd(x, w, b))                              # the same pre-activation as the plain product
    g_ref = F.gelu(h.float())                                                  # GELU of the ROUNDED pre-activation, like the reference
    assert rel(gl, g_ref) < (3e-3 if dtype == torch.bfloat16 else 4e-4)
    ulp = 2.0 ** (-8 if dtype == torch.bfloat16 else -11)
    assert float(((gl.float() - g_ref).abs() / g_ref.abs().clamp_min(2.0 ** -6)).max()) < 1.01 * ulp


@pytest.mark.parametrize("M,N,K", [(513, 4096, 1024), (700, 1024, 128)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_linear_dgelu_bwd(M, N, K, dtype):
    """dx-side of c_proj fused with the GELU backward: N = the hidden width (4 W), K = W."""
    from npcd.hip import elementwise as ew
    from npcd.hip import linear as hl
    from npcd.hip import check, lib, ptr, stream_ptr
    g = torch.Generator().manual_seed(M + N)
    dy = torch.randn(M, K, generator=g).to(dtype).cuda()
    w = (torch.randn(K, N, generator=g) / K ** 0.5).to(dtype).cuda()          # the c_proj weight [W, 4 W]
    h = (torch.randn(M, N, generator=g) * 2).to(dtype).cuda()
    wt = hl.transpose16(w)
    assert torch.equal(wt, w.t().contiguous())
    dh, part, rows = hl.linear_dgelu_bwd(dy, wt, h)
    dg = (dy.float() @ w.float()).to(dtype)                                   # the library's 16-bit data gradient
    hr = h.float().requires_grad_(True)
    F.gelu(hr).backward(dg.float())
    assert rel(dh, hr.grad) < (5e-3 if dtype == torch.bfloat16 else 8e-4)
    db = torch.empty(N, device="cuda")
    check(lib().npcd_colsum_finalize(ptr(part), rows, N, ptr(db), 0, stream_ptr()), "npcd_colsum_finalize")
    assert rel(db, dh.float().sum(0)) < 1e-5                                  # the sums of the SAME rounded values the GEMMs see
    # against the separate kernels of the previous rounds on the same 16-bit dg
    db2 = torch.empty(N, device="cuda")
    dh2 = ew.gelu_bwd(dg, h, db2)
    assert rel(dh, dh2) < (3e-3 if dtype == torch.bfloat16 else 5e-4)
