// Dev probe: would the n = 513 attention forward gain from the 16x16x32 matrix instruction?
// One software-pipelined attention STAGE (32 queries x 32 keys, d = 64) in a loop, in both instruction shapes, with the stage
// structure of csrc/attention.hip (fwd_stage): [12 LDS fragment reads] wait [score products + the PV products of the previous P]
// [softmax vector work on the new scores: max3 x 8, (fma + exp2) x 16, add x 16, cvt_pk x 8] -- the SAME vector instruction mix in
// both shapes, dependent on the accumulators like in the kernel.  Three waves per SIMD (768 workgroups of 256 threads,
// __launch_bounds__(256, 3)), random operands in LDS (the clock the chip holds depends on the data: MI355X_MICROARCH.md, DVFS
// give-back), wall time by HIP events and the in-kernel clock by s_memtime / s_memrealtime.
//   shape A: 8 x v_mfma_f32_32x32x16_bf16 per stage (4 chained score products, 4 PV products on two accumulators)
//   shape B: 16 x v_mfma_f32_16x16x32_bf16 per stage (4 score tiles x 2 chained, 8 PV tiles), the layout of a 16x16 port:
//            same accumulator registers (16 + 32), same operand registers, same LDS bytes
// Build + run: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o /tmp/asp tools/probes/attn_shape_probe.hip && /tmp/asp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, b2));
}
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// the softmax vector work of one stage on 16 scores of a lane: returns the packed P (8 dwords), updates l
__device__ __forceinline__ void softmax16(const float (&s)[16], float m, float c, float& l, unsigned (&p)[8], float& mx_out) {
    float mx = fmaxf(s[0], s[1]);
#pragma unroll
    for (int i = 2; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s[i]), s[i + 1]);
    mx_out = mx;
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * j], c, -m)), b = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * j + 1], c, -m));
        rs += a + b;
        p[j] = pack2(a, b);
    }
    l += rs;
}

__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    const unsigned long long v = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    const void* sb = (const void*)(((unsigned long long)hi << 32) | lo);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(lds_dst) : "memory");
}
// what the shipped kernels do once per 64-key tile (kv_mid): ring = 1: wait for the tile in flight + barrier + this wave's four 1-KiB pieces of the tile after next;
// 2: the wait + barrier only; 3: the four pieces only (no barrier: timing only); 0: nothing (the bare stage loop)
__device__ __forceinline__ void ring_step(int ring, int tile, const unsigned* src, unsigned char* smem, int wave, int lane) {
    if (ring == 0) return;
    if (ring != 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (ring != 2) {
        const char* sb = reinterpret_cast<const char*>(src) + (((size_t)(blockIdx.x >> 2) * 131072 + (size_t)tile * 16384 + wave * 4096) & 0xFFFFF);
        const unsigned dst = lds_addr_of(smem) + ((tile + 2) % 3) * 16384 + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(sb + i * 1024, (unsigned)lane * 16u, __builtin_amdgcn_readfirstlane(dst + i * 1024));
    }
}

template <int SHAPE>
__global__ __launch_bounds__(256, 3) void stage_loop(const unsigned* __restrict__ src, float* __restrict__ out, long long* __restrict__ clk, int iters, int ring) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 16384];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 3 * 16384 / 16; i += 256) reinterpret_cast<u32x4*>(smem)[i] = reinterpret_cast<const u32x4*>(src)[(blockIdx.x * 131 + i) % 65536];
    __syncthreads();
    // fragment addresses: a conflict-free linear pattern per instruction (the kernels' swizzled images are conflict-free too)
    const unsigned char* kbase = smem + lane * 16;            // ds_read_b128: 1 KiB per wave-instruction
    const unsigned char* vbase = smem + 8192 + lane * 8;      // 8-byte reads (stand-in for ds_read_b64_tr_b16: same bytes, same issue)
    bf16x8 qf[4];
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(smem + 32768 + s * 1024 + lane * 16);
    float m = 4.f, l = 0.f;
    const float c = 0.18f;
    unsigned pw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t0 = 0, r0 = 0;
    if (SHAPE == 0) {
        f32x16 o0 = {0}, o1 = {0};
        for (int it = 0; it < iters; ++it) {
            if (it == 8) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
            const int slot = (it % 3) * 16384 & 8191;          // (stays inside the first 8 KiB: the data does not matter, the bytes do)
            u32x4 kr[4];
            u32x2 vr[8];
#pragma unroll
            for (int s = 0; s < 4; ++s) kr[s] = *reinterpret_cast<const u32x4*>(kbase + ((slot + s * 1024) & 8191));
#pragma unroll
            for (int s = 0; s < 8; ++s) vr[s] = *reinterpret_cast<const u32x2*>(vbase + ((slot + s * 512) & 8191));
            lds_wait();
            f32x16 s0 = {0};
#pragma unroll
            for (int s = 0; s < 4; ++s) s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kr[s]), qf[s], s0, 0, 0, 0);
            const u32x4 p0 = {pw[0], pw[1], pw[2], pw[3]}, p1 = {pw[4], pw[5], pw[6], pw[7]};
            const u32x4 v0 = {vr[0][0], vr[0][1], vr[1][0], vr[1][1]}, v1 = {vr[2][0], vr[2][1], vr[3][0], vr[3][1]};
            const u32x4 v2 = {vr[4][0], vr[4][1], vr[5][0], vr[5][1]}, v3 = {vr[6][0], vr[6][1], vr[7][0], vr[7][1]};
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v0), __builtin_bit_cast(bf16x8, p0), o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v1), __builtin_bit_cast(bf16x8, p0), o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v2), __builtin_bit_cast(bf16x8, p1), o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v3), __builtin_bit_cast(bf16x8, p1), o1, 0, 0, 0);
            float sv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[i] = s0[i];
            float mx;
            softmax16(sv, m, c, l, pw, mx);
            // the kernels' lane-half exchange + deferred-maximum test (never taken here: m stays put)
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
            mx = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])) * c;
            if (__any(mx > m + 1e30f)) m = mx;
            if (it & 1) ring_step(ring, it >> 1, src, smem, __builtin_amdgcn_readfirstlane(tid >> 6), lane);
        }
        float acc = l + m;
        for (int i = 0; i < 16; ++i) acc += o0[i] + o1[i];
        out[blockIdx.x * 256 + tid] = acc;
    } else {
        f32x4 o[8];
        for (int t = 0; t < 8; ++t) o[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            if (it == 8) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
            const int slot = (it % 3) * 16384 & 8191;
            u32x4 kr[4];
            u32x2 vr[8];
#pragma unroll
            for (int s = 0; s < 4; ++s) kr[s] = *reinterpret_cast<const u32x4*>(kbase + ((slot + s * 1024) & 8191));
#pragma unroll
            for (int s = 0; s < 8; ++s) vr[s] = *reinterpret_cast<const u32x2*>(vbase + ((slot + s * 512) & 8191));
            lds_wait();
            // scores: key tile kt (A = kr[2 kt + s]) x query tile qt (B = qf[2 qt + s]), two chained k-steps of 32
            f32x4 st[2][2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kr[2 * kt]), qf[2 * qt], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kr[2 * kt + 1]), qf[2 * qt + 1], a, 0, 0, 0);
                    st[kt][qt] = a;
                }
            // PV of the previous P: d tile dt (A = two transposed reads) x query tile qt (B = pw[4 qt .. 4 qt + 3]), one k-step of 32 keys
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const u32x4 va = {vr[2 * dt][0], vr[2 * dt][1], vr[2 * dt + 1][0], vr[2 * dt + 1][1]};
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    const u32x4 pb = {pw[4 * qt], pw[4 * qt + 1], pw[4 * qt + 2], pw[4 * qt + 3]};
                    o[2 * dt + qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, va), __builtin_bit_cast(bf16x8, pb), o[2 * dt + qt], 0, 0, 0);
                }
            }
            // a lane's 16 scores: query tile qt -> keys of both key tiles (8 values), the order in which P is handed over
            float sv[16];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) sv[8 * qt + 4 * kt + i] = st[kt][qt][i];
            float mx;
            softmax16(sv, m, c, l, pw, mx);
            // deferred-maximum test on the lane's own scores (the four lanes of a query only meet when it fires)
            if (__any(mx * c > m + 1e30f)) m = mx * c;
            if (it & 1) ring_step(ring, it >> 1, src, smem, __builtin_amdgcn_readfirstlane(tid >> 6), lane);
        }
        float acc = l + m;
        for (int t = 0; t < 8; ++t) acc += o[t][0] + o[t][1] + o[t][2] + o[t][3];
        out[blockIdx.x * 256 + tid] = acc;
    }
    if (tid == 0 && blockIdx.x < 768) {
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

static int g_wgs = 768, g_ring = 0;          // resident workgroups: 768 = three waves per SIMD, 512 = two, 256 = one
template <int SHAPE>
static void run(const unsigned* src, float* out, long long* clk, int iters, const char* name) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) stage_loop<SHAPE><<<g_wgs, 256>>>(src, out, clk, iters, g_ring);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        stage_loop<SHAPE><<<g_wgs, 256>>>(src, out, clk, iters, g_ring);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    std::vector<long long> h(2 * 768);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int b = 0; b < g_wgs; ++b) { cyc += (double)h[2 * b]; rt += (double)h[2 * b + 1]; }
    // a SIMD runs 3 waves: (iters - 8) stages per wave
    const double stages = (double)(iters - 8);
    printf("%-28s wall %.3f ms (best %.3f)  in-kernel clock %.2f GHz  %.0f cycles per stage and wave = %.0f per stage and SIMD  %.1f TFLOP/s\n", name, sum / reps, best,
           cyc / rt / 10.0, cyc / g_wgs / stages, cyc / g_wgs / stages / (g_wgs / 256.0), g_wgs * 4.0 * iters * 2.0 * 32 * 32 * 64 * 2 / (sum / reps * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const bool zeros = argc > 2 && atoi(argv[2]) != 0;
    if (argc > 3) g_wgs = atoi(argv[3]);
    if (argc > 4) g_ring = atoi(argv[4]);
    unsigned* src; float* out; long long* clk;
    hipMalloc(&src, 65536 * 16 + 3 * 16384); hipMalloc(&out, 768 * 256 * 4); hipMalloc(&clk, 2 * 768 * 8);
    std::vector<unsigned> h(65536 * 4 + 3 * 4096);
    srand(1);
    for (auto& x : h) {
        // two random bf16 values in [-1, 1): sign, exponent 118..126, 7 mantissa bits
        auto one = [&]() { return (unsigned)(((rand() & 1) << 15) | ((118 + rand() % 9) << 7) | (rand() & 127)); };
        x = zeros ? 0u : (one() << 16) | one();
    }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("stage loop, %d stages per wave, %d workgroups = %d waves per SIMD, %s operands, per-tile ring step %d (0 none, 1 wait + barrier + 4 DMA pieces, 2 wait + barrier, 3 pieces only)\n", iters, g_wgs, g_wgs / 256, zeros ? "ZERO" : "random", g_ring);
    for (int round = 0; round < 2; ++round) {
        run<0>(src, out, clk, iters, "32x32x16 (8 per stage)");
        run<1>(src, out, clk, iters, "16x16x32 (16 per stage)");
    }
    return 0;
}
