// Dev probe: how many vector instructions of the SAME wave hide behind one v_mfma_f32_32x32x16_f16 when a SIMD runs a single wave
// (256-thread workgroup, one per CU)?  Loop of 8 independent matrix instructions, each followed by K fillers; cycles per matrix
// instruction (s_memtime) for K = 0..8, accumulators in VGPRs ("v") or AGPRs ("a"), fillers = v_pk_mul_f16 / v_cvt_pk_f16_f32 /
// v_accvgpr_write.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/probes/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FILL_PK(n)  asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(f[n]) : "v"(one));
#define FILL_CVT(n) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(f[n]) : "v"(x));
#define FILL_ACC(n) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(g[n]) : "v"(x));

template <int K, int KIND, bool AG>
__global__ __launch_bounds__(256, 1) void probe(long long* out, int iters) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(threadIdx.x * 0.001f); b[j] = (_Float16)0.5f; }
    f32x16 c[8];
    for (int t = 0; t < 8; ++t) for (int j = 0; j < 16; ++j) c[t][j] = 0.f;
    unsigned f[8] = {1, 2, 3, 4, 5, 6, 7, 8}, one = 0x3c003c00u; float g[8]; float x = 1.5f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[t]) : "v"(a), "v"(b));
            else if (KIND == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[t]) : "v"(a), "a"(b));       // B operand in AGPRs
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[t]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0 || KIND == 3) { FILL_PK(k) } else if (KIND == 1) { FILL_CVT(k) } else { FILL_ACC(k) }
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += c[t][0];
    for (int k = 0; k < 8; ++k) s += (float)f[k];
    if (KIND == 2) for (int k = 0; k < K; ++k) s += g[k];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; }
}
template <int K, int KIND, bool AG>
static void run(long long* d) {
    const int iters = 2000;
    long long h[2];
    probe<K, KIND, AG><<<256, 256>>>(d, iters);
    probe<K, KIND, AG><<<256, 256>>>(d, iters);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("%s acc, filler %s, K = %d: %.1f cycles per matrix instruction\n", AG ? "AGPR" : "VGPR", KIND == 0 ? "v_pk_mul_f16" : KIND == 3 ? "v_pk_mul_f16 (B operand in AGPRs)" : KIND == 1 ? "v_cvt_pk_f16_f32" : "v_accvgpr_write", K,
           (double)h[0] / (iters * 8.0));
}
int main() {
    long long* d; hipMalloc(&d, 16);
    run<0, 0, false>(d); run<2, 0, false>(d); run<4, 0, false>(d); run<5, 0, false>(d); run<6, 0, false>(d); run<8, 0, false>(d);
    run<0, 0, true>(d); run<2, 0, true>(d); run<4, 0, true>(d); run<6, 0, true>(d); run<8, 0, true>(d);
    run<0, 3, false>(d); run<2, 3, false>(d); run<4, 3, false>(d); run<4, 1, false>(d); run<6, 1, false>(d); run<4, 2, false>(d); run<6, 2, false>(d);
    return 0;
}
