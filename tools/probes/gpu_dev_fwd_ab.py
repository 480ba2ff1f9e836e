"""Dev probe: the forms of the attention forward (NPCD_ATTN_FWD=32: 32 query rows per wave; 64: 64 rows per wave, one item per
workgroup; default: persistent workgroups for 256 j / 256 j + 1 tokens) in ONE
process, interleaved rounds: error against an fp32 reference (output rel-L2, LSE max abs) and HIP-event time per launch.
usage: python3 tools/probes/gpu_dev_fwd_ab.py [rounds] [n ...]"""
import sys, os, math
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "neural-point-cloud-diffusion_amd"))
import torch
from npcd.hip import attention as A
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ns = [int(x) for x in sys.argv[2:]] or [513]
H, d = 16, 64
scale = 1 / math.sqrt(d)
FORMS = (("32", "32 rows/wave"), ("64", "64 rows/wave"), ("p", "64 rows/wave, persistent (where the shape fits)"))


def run(form, q, k, v):
    os.environ["NPCD_ATTN_FWD"] = form
    return A._fwd(q, k, v, scale)


for n in ns:
    B = int(os.environ.get("NPCD_B", "0")) or (64 if n <= 600 else 32)
    torch.manual_seed(0)
    mult = float(os.environ.get("NPCD_QK_MULT", "1"))
    qkv = torch.randn(B, n, H, 3 * d, device="cuda")
    qkv[..., :2 * d] *= mult
    qkv = qkv.bfloat16()
    if os.environ.get("NPCD_ZERO_DATA"):
        qkv.zero_()
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    qs = qkv[:2].float()
    qq, kk, vv = (qs[..., i * d:(i + 1) * d].permute(0, 2, 1, 3) for i in range(3))
    sc = qq @ kk.transpose(-1, -2) * scale
    ref = (torch.softmax(sc, -1) @ vv).permute(0, 2, 1, 3)
    ref_lse = torch.logsumexp(sc, -1)
    rel = lambda a, b: ((a.float() - b).norm() / b.norm()).item()
    for form, name in FORMS:
        out, lse = run(form, q, k, v)
        torch.cuda.synchronize()
        print(f"n={n} {name}: rel-L2 out {rel(out[:2], ref):.2e}  max|lse diff| {float((lse[:2] - ref_lse).abs().max()):.2e}  finite {bool(torch.isfinite(out).all())}", flush=True)
    times = {f: [] for f, _ in FORMS}
    for r in range(rounds):
        for form, _ in FORMS:
            for _ in range(3):
                run(form, q, k, v)
            evs = []
            for _ in range(20):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(form, q, k, v); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            times[form].append(ts[len(ts) // 2] * 1e3)
    fl = 4 * B * H * n * n * d
    for form, name in FORMS:
        t = sorted(times[form])
        print(f"n={n} B={B} {name}: median-of-rounds {t[len(t) // 2]:.1f} us  min {t[0]:.1f}  max {t[-1]:.1f}  -> {fl / t[len(t) // 2] / 1e6:.0f} TFLOP/s", flush=True)
