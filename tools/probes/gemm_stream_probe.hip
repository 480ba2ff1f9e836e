// Dev probe: what can ANY 256 x 256-tile bf16 NT GEMM kernel reach on this chip at the c_fc shape
// (M = 32,768 tokens, N = 4,096, K = 1,024), in the structure the verdict names -- one wave per SIMD, 128 x 128 wave tiles on
// v_mfma_f32_16x16x32_bf16, both operands streamed through LDS by LDS-DMA (global_load_lds_dwordx4)?
// Three arms of ONE persistent kernel (256 workgroups of 256 threads, 8 tiles of 256 x 256 each, XCD-contiguous tile order, K-steps of 64
// through two 64-KB LDS stages, one counted wait + barrier per K-step):
//   1  the STREAM alone: every K-step's 64 KB (x rows + w rows of the tile) requested by 16 DMA instructions per wave, nothing else;
//   2  the MATRIX INSTRUCTIONS alone: 128 per wave and K-step on register operands (random data), no memory traffic;
//   3  both, the DMA instructions spread between the matrix instructions (one per eight) -- still WITHOUT the 32 LDS fragment reads per
//      wave and K-step and without an epilogue: an upper bound of what a real kernel of this structure can do.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/gsp tools/probes/gemm_stream_probe.hip && /tmp/gsp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    const unsigned long long v = (unsigned long long)sbase;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    const void* sb = (const void*)(((unsigned long long)hi << 32) | lo);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(lds_dst) : "memory");
}

constexpr int M = 32768, N = 4096, K = 1024, KSTEPS = K / 64;

template <int ARM>
__global__ __launch_bounds__(256, 1) void probe(const __bf16* __restrict__ x, const __bf16* __restrict__ w, float* __restrict__ out, const unsigned* __restrict__ rnd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];        // 2 stages x (x half 32 KB | w half 32 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile order: each XCD (blockIdx & 7) takes a contiguous band of row tiles, 8 x 4 super-tiles (the order of csrc/gemm_nt.hip)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;                    // 32 workgroups per XCD
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x4 ra = reinterpret_cast<const u32x4*>(rnd)[(tid * 8 + i) & 4095], rb = reinterpret_cast<const u32x4*>(rnd)[(tid * 8 + 4 + i) & 4095];
        a[i] = __builtin_bit_cast(bf16x8, ra);
        b[i] = __builtin_bit_cast(bf16x8, rb);
    }
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // per-lane source offset of a 1-KiB piece: 8 rows x 128 B, row stride K * 2 B (XOR swizzle of the chunk on the source side)
    const unsigned voff = (unsigned)((lane >> 3) * (K * 2) + (((lane & 7) ^ (lane >> 3)) << 4));
    const unsigned lds0 = lds_addr(smem);
    for (int it = 0; it < 8; ++it) {
        // tile (rt, ct): M / 256 = 128 row tiles, N / 256 = 16 column tiles; XCD x takes row tiles 16 x .. 16 x + 15
        // the XCD's band of 16 x 16 tiles in 8 super-tiles of 8 row tiles x 4 column tiles (4 MB of x + 2 MB of w per round and L2)
        const int rt = xcd * 16 + (it >> 2) * 8 + (slot >> 2), ct = (it & 3) * 4 + (slot & 3);
        const char* xs = reinterpret_cast<const char*>(x + (size_t)rt * 256 * K);
        const char* ws = reinterpret_cast<const char*>(w + (size_t)ct * 256 * K);
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const unsigned stage = lds0 + (ks & 1) * 65536;
            // this wave's 16 pieces of the K-step: 8 of the x half (rows wave * 64 .. + 63), 8 of the w half
            if (ARM != 2) {
#pragma unroll
                for (int pc = 0; pc < 16; ++pc) {
                    const bool second = pc >= 8;
                    const int piece = wave * 8 + (pc & 7);                      // 32 pieces of 8 rows per half
                    const char* src = (second ? ws : xs) + (size_t)piece * 8 * (K * 2) + ks * 128;
                    dma16(src, voff, __builtin_amdgcn_readfirstlane(stage + (second ? 32768 : 0) + piece * 1024));
                    if (ARM == 3) {
#pragma unroll
                        for (int m = 0; m < 8; ++m) {
                            const int t = pc * 8 + m;
                            acc[t & 63] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[(t >> 2) & 3], acc[t & 63], 0, 0, 0);
                        }
                    }
                }
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");               // the PREVIOUS K-step's pieces have landed (one stage of prefetch)
                __builtin_amdgcn_s_barrier();
            } else {
#pragma unroll
                for (int t = 0; t < 128; ++t) acc[t & 63] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[(t >> 2) & 3], acc[t & 63], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (ARM != 2) s += (float)smem[(tid * 16) & 65535];
    out[blockIdx.x * 256 + tid] = s;
}

// Arm 10: arm 3 with REGISTER staging instead of LDS-DMA: 16 x global_load_dwordx4 per wave and K-step (the same bytes through the same
// vector-memory path) into one of two register sets, spread between the matrix instructions; the set loaded one K-step earlier is written
// to LDS with 16 x ds_write_b128 behind the loads.  Does an ordinary load hold its wave at issue the way the LDS-DMA instruction does?
template <int CUR>
__device__ __forceinline__ void regstage_step(const char* xs, const char* ws, int ks, int wave, unsigned voff, unsigned char* smem, int lane,
                                              u32x4 (&cur)[16], const u32x4 (&prev)[16], bool have_prev, const bf16x8 (&a)[4], const bf16x8 (&b)[4], f32x4 (&acc)[64]) {
#pragma unroll
    for (int pc = 0; pc < 16; ++pc) {
        const bool second = pc >= 8;
        const int piece = wave * 8 + (pc & 7);
        const char* src = (second ? ws : xs) + (size_t)piece * 8 * (K * 2) + ks * 128;
        cur[pc] = *reinterpret_cast<const u32x4*>(src + voff);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int t = pc * 8 + m;
            acc[t & 63] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[(t >> 2) & 3], acc[t & 63], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (have_prev) {
        unsigned char* stage = smem + ((ks + 1) & 1) * 65536;
#pragma unroll
        for (int pc = 0; pc < 16; ++pc) {
            const bool second = pc >= 8;
            const int piece = wave * 8 + (pc & 7);
            *reinterpret_cast<u32x4*>(stage + (second ? 32768 : 0) + piece * 1024 + lane * 16) = prev[pc];
        }
    }
    __builtin_amdgcn_s_barrier();
}
__global__ __launch_bounds__(256, 1) void probe_regstage(const __bf16* __restrict__ x, const __bf16* __restrict__ w, float* __restrict__ out, const unsigned* __restrict__ rnd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x4 ra = reinterpret_cast<const u32x4*>(rnd)[(tid * 8 + i) & 4095], rb = reinterpret_cast<const u32x4*>(rnd)[(tid * 8 + 4 + i) & 4095];
        a[i] = __builtin_bit_cast(bf16x8, ra);
        b[i] = __builtin_bit_cast(bf16x8, rb);
    }
    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned voff = (unsigned)((lane >> 3) * (K * 2) + (((lane & 7) ^ (lane >> 3)) << 4));
    u32x4 r0[16], r1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r0[i] = r1[i] = u32x4{0, 0, 0, 0};
    for (int it = 0; it < 8; ++it) {
        const int rt = xcd * 16 + (it >> 2) * 8 + (slot >> 2), ct = (it & 3) * 4 + (slot & 3);
        const char* xs = reinterpret_cast<const char*>(x + (size_t)rt * 256 * K);
        const char* ws = reinterpret_cast<const char*>(w + (size_t)ct * 256 * K);
        for (int ks = 0; ks < KSTEPS; ks += 2) {
            regstage_step<0>(xs, ws, ks, wave, voff, smem, lane, r0, r1, ks > 0 || it > 0, a, b, acc);
            regstage_step<1>(xs, ws, ks + 1, wave, voff, smem, lane, r1, r0, true, a, b, acc);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += (float)smem[(tid * 16) & 65535] + (float)(r0[0][0] & 1u) + (float)(r1[0][0] & 1u);
    out[blockIdx.x * 256 + tid] = s;
}
static void run_regstage(const __bf16* x, const __bf16* w, float* out, const unsigned* rnd) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_regstage), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) probe_regstage<<<256, 256, 131072>>>(x, w, out, rnd);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < 10; ++r) {
        hipEventRecord(e0);
        probe_regstage<<<256, 256, 131072>>>(x, w, out, rnd);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double flop = 2.0 * M * N * K, bytes = 2048.0 * KSTEPS * 65536;
    printf("%-44s %.1f us (best %.1f)   %.0f TFLOP/s   stream %.2f TB/s\n", "10 = 3 with register staging (load + ds_write)", sum / 10 * 1e3, best * 1e3,
           flop / (sum / 10 * 1e-3) / 1e12, bytes / (sum / 10 * 1e-3) / 1e12);
}

// Arms 4 / 5: the same tile with EIGHT waves (two per SIMD, 128 x 64 wave tiles = 32 accumulators of 16 x 16, 64 matrix instructions and 8 DMA
// pieces per wave and K-step).  4: every wave [8 DMA | 64 MFMA] in the same order (the two waves of a SIMD request at the same time and
// compute at the same time);  5: STAGGERED roles -- waves 0-3 request first and then compute, waves 4-7 compute first and then request,
// so that on every SIMD one wave issues matrix instructions while the other is held in DMA issue.
template <int ARM>
__global__ __launch_bounds__(512, 2) void probe8(const __bf16* __restrict__ x, const __bf16* __restrict__ w, float* __restrict__ out, const unsigned* __restrict__ rnd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x4 ra = reinterpret_cast<const u32x4*>(rnd)[(tid * 8 + i) & 4095], rb = reinterpret_cast<const u32x4*>(rnd)[(tid * 8 + 4 + i) & 4095];
        a[i] = __builtin_bit_cast(bf16x8, ra);
        b[i] = __builtin_bit_cast(bf16x8, rb);
    }
    f32x4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned voff = (unsigned)((lane >> 3) * (K * 2) + (((lane & 7) ^ (lane >> 3)) << 4));
    const unsigned lds0 = lds_addr(smem);
    const bool late = (ARM == 5 || ARM == 9) && wave >= 4;                      // (wave-uniform) requests AFTER its matrix instructions
    if (ARM == 8 && wave < 4) __builtin_amdgcn_s_setprio(3);                    // arm 8 = arm 7 with the requesting waves at raised priority
    for (int it = 0; it < 8; ++it) {
        const int rt = xcd * 16 + (it >> 2) * 8 + (slot >> 2), ct = (it & 3) * 4 + (slot & 3);
        const char* xs = reinterpret_cast<const char*>(x + (size_t)rt * 256 * K);
        const char* ws = reinterpret_cast<const char*>(w + (size_t)ct * 256 * K);
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const unsigned stage = lds0 + (ks & 1) * 65536;
            auto request = [&]() {
                if (ARM == 9) __builtin_amdgcn_s_setprio(3);                      // arm 9 = arm 5 with the priority raised around the requests
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) {                                  // waves 0-3: the x half, waves 4-7: the w half (8 pieces each)
                    const bool second = wave >= 4;
                    const int piece = (wave & 3) * 8 + pc;
                    const char* src = (second ? ws : xs) + (size_t)piece * 8 * (K * 2) + ks * 128;
                    dma16(src, voff, __builtin_amdgcn_readfirstlane(stage + (second ? 32768 : 0) + piece * 1024));
                }
                if (ARM == 9) __builtin_amdgcn_s_setprio(0);
            };
            auto compute = [&]() {
#pragma unroll
                for (int t = 0; t < 64; ++t) acc[t & 31] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[(t >> 2) & 3], acc[t & 31], 0, 0, 0);
            };
            if (ARM == 6) {              // fine interleave: one DMA piece, eight matrix instructions, ... in every wave
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) {
                    const bool second = wave >= 4;
                    const int piece = (wave & 3) * 8 + pc;
                    const char* src = (second ? ws : xs) + (size_t)piece * 8 * (K * 2) + ks * 128;
                    dma16(src, voff, __builtin_amdgcn_readfirstlane(stage + (second ? 32768 : 0) + piece * 1024));
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int t = pc * 8 + m;
                        acc[t & 31] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[(t >> 2) & 3], acc[t & 31], 0, 0, 0);
                    }
                }
            } else if (ARM == 7 || ARM == 8) {       // only waves 0-3 request (16 pieces each), waves 4-7 only compute (128 matrix instructions): does a wave
                                         // held in DMA issue stop its SIMD partner's matrix instructions?
                if (wave < 4) {
#pragma unroll
                    for (int pc = 0; pc < 16; ++pc) {
                        const bool second = pc >= 8;
                        const int piece = wave * 8 + (pc & 7);
                        const char* src = (second ? ws : xs) + (size_t)piece * 8 * (K * 2) + ks * 128;
                        dma16(src, voff, __builtin_amdgcn_readfirstlane(stage + (second ? 32768 : 0) + piece * 1024));
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 128; ++t) acc[t & 31] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t & 3], b[(t >> 2) & 3], acc[t & 31], 0, 0, 0);
                }
            } else if (!late) { request(); __builtin_amdgcn_sched_barrier(0); compute(); }
            else { compute(); __builtin_amdgcn_sched_barrier(0); request(); }
            if (ARM == 7 || ARM == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");               // the previous K-step's pieces of this wave have landed
            __builtin_amdgcn_s_barrier();
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    s += (float)smem[(tid * 16) & 65535];
    out[blockIdx.x * 512 + tid] = s;
}
template <int ARM>
static void run8(const __bf16* x, const __bf16* w, float* out, const unsigned* rnd, const char* name) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe8<ARM>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) probe8<ARM><<<256, 512, 131072>>>(x, w, out, rnd);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < 10; ++r) {
        hipEventRecord(e0);
        probe8<ARM><<<256, 512, 131072>>>(x, w, out, rnd);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double flop = 2.0 * M * N * K, bytes = 2048.0 * KSTEPS * 65536;
    printf("%-44s %.1f us (best %.1f)   %.0f TFLOP/s   stream %.2f TB/s\n", name, sum / 10 * 1e3, best * 1e3, flop / (sum / 10 * 1e-3) / 1e12,
           bytes / (sum / 10 * 1e-3) / 1e12);
}

template <int ARM>
static void run(const __bf16* x, const __bf16* w, float* out, const unsigned* rnd, const char* name) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<ARM>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) probe<ARM><<<256, 256, 131072>>>(x, w, out, rnd);
    hipDeviceSynchronize();
    float best = 1e30f, sum = 0.f;
    for (int r = 0; r < 10; ++r) {
        hipEventRecord(e0);
        probe<ARM><<<256, 256, 131072>>>(x, w, out, rnd);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sum += ms;
    }
    const double flop = 2.0 * M * N * K, bytes = 2048.0 * KSTEPS * 65536;
    printf("%-44s %.1f us (best %.1f)", name, sum / 10 * 1e3, best * 1e3);
    if (ARM != 1) printf("   %.0f TFLOP/s", flop / (sum / 10 * 1e-3) / 1e12);
    if (ARM != 2) printf("   stream %.2f TB/s = %.1f GB/s per CU", bytes / (sum / 10 * 1e-3) / 1e12, bytes / (sum / 10 * 1e-3) / 1e9 / 256);
    printf("\n");
}

int main() {
    __bf16 *x, *w; float* out; unsigned* rnd;
    hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&rnd, 4096 * 16);
    std::vector<unsigned> h((size_t)M * K / 2);
    srand(2);
    auto one = [&]() { return (unsigned)(((rand() & 1) << 15) | ((118 + rand() % 9) << 7) | (rand() & 127)); };
    for (auto& v : h) v = (one() << 16) | one();
    hipMemcpy(x, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
    hipMemcpy(rnd, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    printf("c_fc shape: M %d x N %d x K %d bf16, 256 x 256 tiles, one wave per SIMD; library kernel at this shape: 211-237 us (R4.1)\n", M, N, K);
    for (int round = 0; round < 2; ++round) {
        run<1>(x, w, out, rnd, "1 stream alone (64 KB per K-step by LDS-DMA)");
        run<2>(x, w, out, rnd, "2 matrix instructions alone (16x16x32)");
        run<3>(x, w, out, rnd, "3 stream + matrix instructions interleaved");
        run_regstage(x, w, out, rnd);
        run8<4>(x, w, out, rnd, "4 eight waves, same order in every wave");
        run8<5>(x, w, out, rnd, "5 eight waves, staggered roles");
        run8<6>(x, w, out, rnd, "6 eight waves, 1 DMA : 8 MFMA in every wave");
        run8<7>(x, w, out, rnd, "7 waves 0-3 request, waves 4-7 compute");
        run8<8>(x, w, out, rnd, "8 = 7, requesting waves at s_setprio 3");
        run8<9>(x, w, out, rnd, "9 = 5, s_setprio 3 around the requests");
    }
    return 0;
}
