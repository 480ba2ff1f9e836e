// Split-operand form of a Linear layer's input for the fp32-CLASS forward of the denoiser (sampling in the reference's numerics class,
// diffusion_model.py:108-133 / transformer.py:118-137): x W^T with fp32 operands is computed as ONE bf16 library GEMM over the three
// cross products of the split operands,  [xh | xl | xh] [Wh | Wh | Wl]^T  with  x = xh + xl, W = Wh + Wl  (bf16 halves: 16 mantissa
// bits, fp32's exponent range; the lo x lo term, 2^-18 relative, is dropped; fp32 accumulation and output): 3.2e-6 relative against
// float64 at the sampler's shapes (an fp32 GEMM: 4e-7, a bf16 GEMM: 2e-3) at 2.3-3.5 x the rate of the fp32 library GEMM, which runs on
// the fp32 matrix instruction (1/16 of the bf16 rate).
// This file: the activation side.  One pass: y = x (+ bias) (-> exact-erf GELU)  ->  out [T, 3K] bf16 = [hi(y) | lo(y) | hi(y)].
// HBM-bound: 4 K bytes read + 6 K bytes written per row.
#include <math.h>

#include "common.h"

namespace npcd {

template <bool GELU>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, const float* __restrict__ bias, __bf16* __restrict__ out, int64_t T, int K) {
    const int64_t n8 = T * (int64_t)(K / 8);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / (K / 8);
        const int c8 = (int)(i - row * (K / 8));
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + row * K + c8 * 8), b = *reinterpret_cast<const f32x4*>(x + row * K + c8 * 8 + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        if (bias) {
            const f32x4 ba = *reinterpret_cast<const f32x4*>(bias + c8 * 8), bb = *reinterpret_cast<const f32x4*>(bias + c8 * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += ba[j]; v[4 + j] += bb[j]; }
        }
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y = v[j];
            if (GELU) y = 0.5f * y * (1.f + erff(y * 0.70710678118654752f));        // nn.GELU() (exact erf), transformer.py:131
            hi[j] = (__bf16)y;
            lo[j] = (__bf16)(y - (float)hi[j]);
        }
        __bf16* o = out + row * (3 * (int64_t)K) + c8 * 8;
        *reinterpret_cast<bf16x8*>(o) = hi;
        *reinterpret_cast<bf16x8*>(o + K) = lo;
        *reinterpret_cast<bf16x8*>(o + 2 * (int64_t)K) = hi;
    }
}

// Residual add + bias + LayerNorm + split in one pass (the two places of a block where the fp32-class forward hands a LayerNorm output to a
// Linear layer):  xnew = x (+ o + bias);  y = LayerNorm(xnew) * gamma + beta (fp32 statistics, biased variance, like nn.LayerNorm);
// out [T, 3 W] = [hi(y) | lo(y) | hi(y)].  One wave per row, the row in registers (W / 64 values per lane, W a multiple of 256, <= 4096).
template <int VPL>      // float4 groups per lane: W = 256 * VPL
__global__ __launch_bounds__(256) void add_ln_split3_kernel(const float* __restrict__ x, const float* __restrict__ o, const float* __restrict__ bias,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ xnew,
                                                            __bf16* __restrict__ out, int64_t T, float eps) {
    constexpr int W = 256 * VPL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < T; row += (int64_t)gridDim.x * 4) {
        f32x4 v[VPL];
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < VPL; ++g) {
            const int col = (g * 64 + lane) * 4;
            v[g] = *reinterpret_cast<const f32x4*>(x + row * W + col);
            if (o) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(o + row * W + col), b = *reinterpret_cast<const f32x4*>(bias + col);
                v[g] += a + b;
                *reinterpret_cast<f32x4*>(xnew + row * W + col) = v[g];
            }
            sum += (v[g][0] + v[g][1]) + (v[g][2] + v[g][3]);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
        const float mean = sum * (1.f / W);
        float sq = 0.f;
#pragma unroll
        for (int g = 0; g < VPL; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[g][j] - mean; sq += d * d; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m, 64);
        const float rstd = rsqrtf(sq * (1.f / W) + eps);
#pragma unroll
        for (int g = 0; g < VPL; ++g) {
            const int col = (g * 64 + lane) * 4;
            const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + col), be = *reinterpret_cast<const f32x4*>(beta + col);
            bf16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y = (v[g][j] - mean) * rstd * ga[j] + be[j];
                hi[j] = (__bf16)y;
                lo[j] = (__bf16)(y - (float)hi[j]);
            }
            __bf16* op = out + row * (3 * (int64_t)W) + col;
            *reinterpret_cast<bf16x4*>(op) = hi;
            *reinterpret_cast<bf16x4*>(op + W) = lo;
            *reinterpret_cast<bf16x4*>(op + 2 * W) = hi;
        }
    }
}

}  // namespace npcd

using namespace npcd;

extern "C" int npcd_add_ln_split3_bf16(const float* x, const float* o, const float* bias, const float* gamma, const float* beta, float* xnew,
                                       void* out, int64_t rows, int W, float eps, void* stream) {
    if (!x || !gamma || !beta || !out || rows <= 0 || W <= 0) return NPCD_ERR_ARG;
    if ((o != nullptr) != (bias != nullptr) || (o != nullptr) != (xnew != nullptr)) return NPCD_ERR_ARG;
    if (W % 256 != 0 || W > 4096) return NPCD_ERR_UNSUPPORTED;
    for (const void* p : {(const void*)x, (const void*)o, (const void*)bias, (const void*)gamma, (const void*)beta, (const void*)xnew, (const void*)out})
        if (reinterpret_cast<uintptr_t>(p) & 15) return NPCD_ERR_UNSUPPORTED;
    const int grid = (int)((rows + 3) / 4 < 8192 ? (rows + 3) / 4 : 8192);
    hipStream_t st = static_cast<hipStream_t>(stream);
    __bf16* ob = static_cast<__bf16*>(out);
#define NPCD_LAUNCH_ALS(V) hipLaunchKernelGGL(add_ln_split3_kernel<V>, dim3(grid), dim3(256), 0, st, x, o, bias, gamma, beta, xnew, ob, rows, eps)
    switch (W / 256) {
        case 1: NPCD_LAUNCH_ALS(1); break;
        case 2: NPCD_LAUNCH_ALS(2); break;
        case 3: NPCD_LAUNCH_ALS(3); break;
        case 4: NPCD_LAUNCH_ALS(4); break;
        case 8: NPCD_LAUNCH_ALS(8); break;
        case 16: NPCD_LAUNCH_ALS(16); break;
        default: return NPCD_ERR_UNSUPPORTED;
    }
#undef NPCD_LAUNCH_ALS
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_split3_bf16(const float* x, const float* bias, void* out, int64_t rows, int K, int gelu, void* stream) {
    if (!x || !out || rows <= 0 || K <= 0) return NPCD_ERR_ARG;
    if (K % 8 != 0 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15)))
        return NPCD_ERR_UNSUPPORTED;
    const int64_t n8 = rows * (K / 8);
    const int grid = (int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (gelu) hipLaunchKernelGGL(split3_kernel<true>, dim3(grid), dim3(256), 0, st, x, bias, static_cast<__bf16*>(out), rows, K);
    else hipLaunchKernelGGL(split3_kernel<false>, dim3(grid), dim3(256), 0, st, x, bias, static_cast<__bf16*>(out), rows, K);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
