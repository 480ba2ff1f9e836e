"""Dev probe: can an HBM-bound elementwise kernel run BESIDE a library GEMM when each gets its own CUs?  Streams with CU masks
(hipExtStreamCreateWithCUMask): (1) rate of an elementwise pass on n CUs, contiguous and strided masks; (2) rate of a bf16 GEMM on
the complementary CUs; (3) both at once against back to back."""
import ctypes, os, sys, time
import torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
NCU = torch.cuda.get_device_properties(0).multi_processor_count
print("CUs:", NCU)


def masked_stream(bits):
    words = (NCU + 31) // 32
    arr = (ctypes.c_uint32 * words)()
    for b in bits:
        arr[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), words, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def timed(fn, stream, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


T, W = 32832, 1024
x = torch.randn(T, 4 * W, device=dev, dtype=torch.bfloat16); y = torch.randn_like(x); z = torch.empty_like(x)
ew = lambda: torch.add(x, y, out=z)                 # 3 x 269 MB per call
ew_bytes = 3 * x.numel() * 2
a = torch.randn(T, W, device=dev, dtype=torch.bfloat16); w = torch.randn(4 * W, W, device=dev, dtype=torch.bfloat16)
out = torch.empty(T, 4 * W, device=dev, dtype=torch.bfloat16)
gemm = lambda: torch.mm(a, w.t(), out=out)
gf = 2.0 * T * W * 4 * W
torch.cuda.synchronize()
full = torch.cuda.current_stream()
print(f"all CUs: elementwise {ew_bytes / timed(ew, full) / 1e6:.2f} TB/s, GEMM {gf / timed(gemm, full) / 1e6:.0f} TF/s")
for n in (32, 64, 96):
    for name, bits in (("contiguous", list(range(n))), ("strided", [i for i in range(NCU) if i % (NCU // n) == 0][:n])):
        s = masked_stream(bits)
        rest = masked_stream([i for i in range(NCU) if i not in set(bits)])
        t_e = timed(ew, s); t_g = timed(gemm, rest)
        # both at once: k GEMMs on `rest` while elementwise passes run on `s`
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(rest):
            for _ in range(20): gemm()
        with torch.cuda.stream(s):
            for _ in range(20): ew()
        torch.cuda.synchronize()
        both = (time.perf_counter() - t0) / 20 * 1e6
        print(f"{n:3d} CUs {name:10s}: elementwise alone {ew_bytes / t_e / 1e6:.2f} TB/s ({t_e:.0f} us), GEMM on the other {NCU - n} alone {gf / t_g / 1e6:.0f} TF/s ({t_g:.0f} us); "
              f"both at once {both:.0f} us per pair (back to back on all CUs: see first line)")
