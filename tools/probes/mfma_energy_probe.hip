// Dev probe (experiments R5.12/R5.13): the shading kernels are power-bound on this pool -- does the matrix-instruction SHAPE change the clock
// the chip holds?  The inner loop of a shading layer (a wave's 64 output channels x 128 rows: weights from registers, activation
// fragments from LDS, fp32 accumulation in 128 registers) in two shapes at equal FLOPs and equal LDS bytes:
//   A: per 32 input channels 2 x [4 ds_read_b128, 8 x v_mfma_f32_32x32x16_f16]       (the shipped form)
//   B: per 32 input channels      8 ds_read_b128, 32 x v_mfma_f32_16x16x32_f16
// two waves per SIMD (512 workgroups of 256 threads), random or zero operands; prints wall time, shader clock, matrix rate.
// Build + run: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form -o /tmp/mep tools/probes/mfma_energy_probe.hip && /tmp/mep [iters] [zero]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ long long g_clk[4];

template <int SHAPE, bool LDS>
__global__ __launch_bounds__(256, 2) void loop_kernel(const f16x8* src, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[128 * 528];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 128 * 528 / 16; i += 256) reinterpret_cast<f16x8*>(smem)[i] = src[(i * 7 + blockIdx.x) & 4095];
    __syncthreads();
    f16x8 a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = src[(tid * 4 + i + 17 * blockIdx.x) & 4095];
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 0) {
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const unsigned char* hb = smem + (lane & 31) * 528 + (lane >> 5) * 16;
        f16x8 b[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) b[cb] = *reinterpret_cast<const f16x8*>(hb + cb * 32 * 528);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                f16x8 bn[4];
                if (LDS) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) bn[cb] = *reinterpret_cast<const f16x8*>(hb + cb * 32 * 528 + ((s + 1) & 15) * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    acc[0][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s & 1], b[cb], acc[0][cb], 0, 0, 0);
                    acc[1][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 + (s & 1)], b[cb], acc[1][cb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (LDS) {
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) b[cb] = bn[cb];
                }
            }
        }
        float x = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) x += acc[i][j][e];
        sink[blockIdx.x * 256 + tid] = x;
    } else {
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
        // B operand of 16x16x32: lane = (row lane & 15, k-group lane >> 4): 16 bytes of row (rb * 16 + lane & 15)
        const unsigned char* hb = smem + (lane & 15) * 528 + (lane >> 4) * 16;
        f16x8 b[8];
#pragma unroll
        for (int rb = 0; rb < 8; ++rb) b[rb] = *reinterpret_cast<const f16x8*>(hb + rb * 16 * 528);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                f16x8 bn[8];
                if (LDS) {
#pragma unroll
                    for (int rb = 0; rb < 8; ++rb) bn[rb] = *reinterpret_cast<const f16x8*>(hb + rb * 16 * 528 + ((s + 1) & 7) * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rb = 0; rb < 8; ++rb)
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) acc[mb][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mb], b[rb], acc[mb][rb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (LDS) {
#pragma unroll
                    for (int rb = 0; rb < 8; ++rb) b[rb] = bn[rb];
                }
            }
        }
        float x = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) x += acc[i][j][e];
        sink[blockIdx.x * 256 + tid] = x;
    }
    if (blockIdx.x == 7 && tid == 0) {
        g_clk[0] = __builtin_amdgcn_s_memtime() - t0;
        g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int SHAPE, bool LDS>
static void run(const char* name, const f16x8* src, float* sink, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL((loop_kernel<SHAPE, LDS>), dim3(512), dim3(256), 0, 0, src, sink, iters);      // warm
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<SHAPE, LDS>), dim3(512), dim3(256), 0, 0, src, sink, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        long long clk[4];
        hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk));
        const double flops = 512.0 * 4 * iters * 16 * 8 * 32768.0;           // per wave and 16-channel step: 8 x 32x32x16
        printf("%-34s %8.3f ms  %6.3f PFLOP/s  clock %.3f GHz  (%lld clocks per wave, %.1f per 32 input channels)\n", name, ms, flops / ms / 1e12,
               clk[0] / (clk[1] * 10.0), clk[0], (double)clk[0] / iters / 8);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const bool zero = argc > 2 && atoi(argv[2]);
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto& v : h) v = zero ? (_Float16)0.f : (_Float16)((rand() % 2001 - 1000) / 1000.f);
    f16x8* src; float* sink;
    hipMalloc(&src, h.size() * 2); hipMalloc(&sink, 512 * 256 * 4);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("%s operands, %d iterations of 16 x 16 input channels\n", zero ? "zero" : "random", iters);
    for (int round = 0; round < 2; ++round) {
        run<0, true>("32x32x16, LDS activation reads", src, sink, iters);
        run<1, true>("16x16x32, LDS activation reads", src, sink, iters);
        run<0, false>("32x32x16, operands in registers", src, sink, iters);
        run<1, false>("16x16x32, operands in registers", src, sink, iters);
    }
    return 0;
}
