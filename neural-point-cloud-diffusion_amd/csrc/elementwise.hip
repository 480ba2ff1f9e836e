// HBM-bound elementwise / normalisation / optimizer kernels of the denoiser training step (gfx950).
//
// They replace chains of eager PyTorch kernels of the reference step
// (npcd/models/diffusion/denoisers/transformer.py:169-172,136-137 under autocast, and
//  npcd/train/diffusion_training.py:169-174 + npcd/utils/ema.py:114-138):
//   residual add + LayerNorm + cast            -> add_ln_fwd      (1 pass instead of 3)
//   LayerNorm backward (dx, dgamma, dbeta) + residual-gradient add + cast + bias-gradient column sum
//                                              -> ln_bwd          (1 pass + a tiny finalize)
//   GELU(erf) forward / backward (+ bias-gradient column sum)     -> gelu_fwd / gelu_bwd
//   column sum of a bf16 matrix (bias gradient of c_qkv)          -> colsum
//   AdamW + EMA + bf16 weight shadow + gradient zeroing           -> adamw_ema (1 pass over 310 M params)
// All are pure streaming kernels: 16-byte accesses per lane, fp32 math, no LDS except the cross-wave
// reduction of column partials.  Roofline: HBM.
#include <math.h>

#include <stdlib.h>

#include "common.h"

namespace npcd {

constexpr int kMaxChunks = 8;  // a wave covers 256 columns per chunk -> width <= 2048

// sum over the 64 lanes, on every lane, entirely on the vector ALU: four DPP row rotations (16-lane rows), v_permlane16_swap,
// v_permlane32_swap -- no ds_bpermute (LDS crossbar) round trips.  Fixed order: bitwise reproducible.
__device__ __forceinline__ float wave_sum(float x) {
#ifdef NPCD_WAVE_SUM_BPERMUTE       // the previous form (A/B builds)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
#else
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124 /* row_ror:4 */, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x122 /* row_ror:2 */, 0xf, 0xf, false));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121 /* row_ror:1 */, 0xf, 0xf, false));
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
#endif
}
// the 16-bit activation type of a training run: __bf16 (bf16 autocast) or _Float16 (the reference's default --dtype, with loss scaling)
template <class E>
struct V {
    typedef E x4 __attribute__((ext_vector_type(4)));
    typedef E x8 __attribute__((ext_vector_type(8)));
};
template <class X4>
__device__ __forceinline__ f32x4 to_f32(X4 v) { return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]}; }
template <class E>
__device__ __forceinline__ typename V<E>::x4 from_f32(f32x4 v) { return typename V<E>::x4{(E)v[0], (E)v[1], (E)v[2], (E)v[3]}; }
__device__ __forceinline__ f32x4 bf16x4_to_f32(bf16x4 v) { return to_f32(v); }
__device__ __forceinline__ bf16x4 f32_to_bf16x4(f32x4 v) { return from_f32<__bf16>(v); }

// ============================================================================================
// x_out = x_in (+ delta);  y = LayerNorm(x_out) * gamma + beta  (bf16);  mean / rstd saved
// one wave per row
// ============================================================================================
// A wave walks rows g, g + NW, g + 2 NW, ... (NW waves in the grid: the grid streams one contiguous band of rows at a time) and
// loads row r + NW before it reduces and stores row r; the affine parameters stay in registers.
template <class E, int NCH>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(const float* __restrict__ x_in, const E* __restrict__ delta,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ x_out, E* __restrict__ y, float* __restrict__ mean,
                                                         float* __restrict__ rstd, int T, int W, float eps) {
    using e4 = typename V<E>::x4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int NW = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    if (row >= T) return;
    f32x4 gm[NCH], bt[NCH], v[NCH], nx[NCH];
    e4 nd[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = ch * 256 + lane * 4;
        gm[ch] = bt[ch] = nx[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
        nd[ch] = e4{0, 0, 0, 0};
        if (c < W) {
            gm[ch] = *reinterpret_cast<const f32x4*>(gamma + c);
            bt[ch] = *reinterpret_cast<const f32x4*>(beta + c);
        }
    }
    auto load_row = [&](int r) {
        const int64_t base = (int64_t)r * W;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = ch * 256 + lane * 4;
            if (c < W) {
                nx[ch] = *reinterpret_cast<const f32x4*>(x_in + base + c);
                if (delta) nd[ch] = *reinterpret_cast<const e4*>(delta + base + c);
            }
        }
    };
    load_row(row);
    for (; row < T; row += NW) {
        const int64_t base = (int64_t)row * W;
        float s = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            v[ch] = nx[ch];
            if (delta) v[ch] += to_f32(nd[ch]);
        }
        if (row + NW < T) load_row(row + NW);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = ch * 256 + lane * 4;
            if (c < W) {
                if (x_out) *reinterpret_cast<f32x4*>(x_out + base + c) = v[ch];
                s += (v[ch][0] + v[ch][1]) + (v[ch][2] + v[ch][3]);
            }
        }
        const float mu = wave_sum(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = ch * 256 + lane * 4;
            if (c < W) {
                const f32x4 d = v[ch] - mu;
                q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)W + eps);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = ch * 256 + lane * 4;
            if (c < W) *reinterpret_cast<e4*>(y + base + c) = from_f32<E>((v[ch] - mu) * rs * gm[ch] + bt[ch]);
        }
        if (lane == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

// ============================================================================================
// LayerNorm backward.  Per row:  xhat = (x - mean) rstd,  g = dy * gamma,
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) (+ dres);  also dx as bf16 for the GEMMs.
// Column partials per workgroup: dgamma += dy * xhat, dbeta += dy, dcol += dx (the bias gradient of the
// Linear whose output was added to the residual stream right before this LayerNorm).
// Workgroup = 4 waves x kRowsPerWave rows; partials [gridDim.x][W] are summed by colsum_finalize.
// ============================================================================================
// Rows per wave are chosen per launch: 2 workgroups are resident per CU (180 VGPRs with the double-buffered row
// loads), i.e. 512 on the chip; the grid is sized to ONE such round when that needs <= 32 rows per wave (cfg-D: 17
// rows, 483 workgroups), otherwise 16 rows per wave.  Small token counts (strong scaling) keep one row per wave.
static inline int ln_rows_per_wave(int T) {
    const int one_round = (T + 4 * 512 - 1) / (4 * 512);
    if (one_round <= 32) return one_round < 1 ? 1 : one_round;
    return 16;
}

template <class E, int NCH>
__global__ __launch_bounds__(256, 2) void ln_bwd_kernel(const E* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                     const float* __restrict__ dres, float* __restrict__ dx, E* __restrict__ dxb,
                                                     float* __restrict__ part_gamma, float* __restrict__ part_beta,
                                                     float* __restrict__ part_col, int T, int W, int rows_per_wave) {
    using e4 = typename V<E>::x4;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    float* red = reinterpret_cast<float*>(dsmem);  // [3][4 waves][W]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 ag[NCH], ab[NCH], ac[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) ag[ch] = ab[ch] = ac[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
    // wave g of NW takes rows g, g + NW, g + 2 NW, ...: the grid reads one contiguous band of NW rows at a time (consecutive
    // rows per wave put every wave on addresses that are equal modulo rows_per_wave * row bytes -- the same HBM channels)
    const int row0 = blockIdx.x * 4 + wave, NW = gridDim.x * 4;
    const int nrows = row0 < T ? min(rows_per_wave, (T - row0 + NW - 1) / NW) : 0;
    // The row loop is software-pipelined: the loads of row r+1 are issued before row r is reduced and stored, so that
    // every wave keeps ~10 KB of reads in flight all the time (the kernel is a pure HBM stream: 536 MB per call at cfg-D).
    e4 dyA[NCH], dyB[NCH];
    f32x4 xA[NCH], xB[NCH], rA[NCH], rB[NCH];
    float muA = 0.f, rsA = 0.f, muB = 0.f, rsB = 0.f;
    auto load_row = [&](int row, e4 (&dyr)[NCH], f32x4 (&xr)[NCH], f32x4 (&rr2)[NCH], float& mu, float& rs) {
        const int64_t base = (int64_t)row * W;
        mu = mean[row];
        rs = rstd[row];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = ch * 256 + lane * 4;
            if (c < W) {
                dyr[ch] = *reinterpret_cast<const e4*>(dy + base + c);
                xr[ch] = *reinterpret_cast<const f32x4*>(x + base + c);
                if (dres) rr2[ch] = *reinterpret_cast<const f32x4*>(dres + base + c);
            }
        }
    };
    if (nrows > 0) load_row(row0, dyA, xA, rA, muA, rsA);
    for (int rr = 0; rr < nrows; ++rr) {
        const int row = row0 + rr * NW;
        const int64_t base = (int64_t)row * W;
        if (rr + 1 < nrows) load_row(row + NW, dyB, xB, rB, muB, rsB);
        const float mu = muA, rs = rsA;
        f32x4 gy[NCH], xh[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = ch * 256 + lane * 4;
            gy[ch] = xh[ch] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < W) {
                const f32x4 d = to_f32(dyA[ch]);
                xh[ch] = (xA[ch] - mu) * rs;
                ag[ch] += d * xh[ch];
                ab[ch] += d;
                gy[ch] = d * *reinterpret_cast<const f32x4*>(gamma + c);
                const f32x4 t = gy[ch] * xh[ch];
                s1 += (gy[ch][0] + gy[ch][1]) + (gy[ch][2] + gy[ch][3]);
                s2 += (t[0] + t[1]) + (t[2] + t[3]);
            }
        }
        const float c1 = wave_sum(s1) / (float)W, c2 = wave_sum(s2) / (float)W;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int c = ch * 256 + lane * 4;
            if (c < W) {
                f32x4 o = (gy[ch] - c1 - xh[ch] * c2) * rs;
                if (dres) o += rA[ch];
                *reinterpret_cast<f32x4*>(dx + base + c) = o;
                if (dxb) *reinterpret_cast<e4*>(dxb + base + c) = from_f32<E>(o);
                ac[ch] += o;
            }
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            dyA[ch] = dyB[ch];
            xA[ch] = xB[ch];
            rA[ch] = rB[ch];
        }
        muA = muB;
        rsA = rsB;
    }
    // cross-wave reduction of the column partials
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int c = ch * 256 + lane * 4;
        if (c < W) {
            *reinterpret_cast<f32x4*>(red + (0 * 4 + wave) * W + c) = ag[ch];
            *reinterpret_cast<f32x4*>(red + (1 * 4 + wave) * W + c) = ab[ch];
            *reinterpret_cast<f32x4*>(red + (2 * 4 + wave) * W + c) = ac[ch];
        }
    }
    __syncthreads();
    float* outs[3] = {part_gamma, part_beta, part_col};
    for (int q = 0; q < 3; ++q) {
        if (!outs[q]) continue;
        for (int c = threadIdx.x; c < W; c += 256) {
            const float* p = red + q * 4 * W + c;
            outs[q][(int64_t)blockIdx.x * W + c] = (p[0] + p[W]) + (p[2 * W] + p[3 * W]);
        }
    }
}

// out[c] (+)= sum_b part[b][c]   (fixed order -> bitwise reproducible).
// Two stages so that the 4 MB of partials are read by many waves: stage 1 = 64 columns x kFinSlices row
// slices per workgroup (coalesced 256-byte rows, 4 row lanes per workgroup reduced through LDS) ->
// stage[kFinSlices][N]; stage 2 sums the slices.
constexpr int kFinSlices = 16;

// A batch of up to NPCD_COLSUM_MAX_JOBS independent column sums shares the two launches (one residual block's
// backward produces eight of them; launched one by one they were 16 launches of ~5 us each per block).
struct FinBatch {
    NpcdColsumJob job[NPCD_COLSUM_MAX_JOBS];
    int first1[NPCD_COLSUM_MAX_JOBS + 1];   // stage-1 blockIdx.x range of each job (64 columns per block)
    int first2[NPCD_COLSUM_MAX_JOBS + 1];   // stage-2 range (256 columns per block)
    int njobs;
};
__device__ __forceinline__ int fin_job(const int* first, int njobs, int bx) {
    int j = 0;
    while (j + 1 < njobs && bx >= first[j + 1]) ++j;
    return j;
}
__device__ __forceinline__ bool fin_two_stage(int nblk) { return nblk > 2 * kFinSlices; }

__global__ __launch_bounds__(256) void colsum_stage1_kernel(FinBatch fb) {
    __shared__ float red[4][64];
    const int j = fin_job(fb.first1, fb.njobs, blockIdx.x);
    const float* __restrict__ part = fb.job[j].part;
    const int nblk = fb.job[j].nblk, N = fb.job[j].N;
    if (!fin_two_stage(nblk)) return;                // few partial rows: stage 2 reads them directly
    float* __restrict__ stage = const_cast<float*>(part) + (int64_t)nblk * N;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = (blockIdx.x - fb.first1[j]) * 64 + tx;
    const int per = (nblk + kFinSlices - 1) / kFinSlices;
    const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
    float s0 = 0.f, s1 = 0.f;
    if (c < N) {
        int b = b0 + ty;
        for (; b + 4 < b1; b += 8) {
            s0 += part[(int64_t)b * N + c];
            s1 += part[(int64_t)(b + 4) * N + c];
        }
        if (b < b1) s0 += part[(int64_t)b * N + c];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && c < N) stage[(int64_t)blockIdx.y * N + c] = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
}

__global__ __launch_bounds__(256) void colsum_stage2_kernel(FinBatch fb) {
    const int j = fin_job(fb.first2, fb.njobs, blockIdx.x);
    const int nblk = fb.job[j].nblk, N = fb.job[j].N;
    const int c = (blockIdx.x - fb.first2[j]) * 256 + threadIdx.x;
    if (c >= N) return;
    const bool two = fin_two_stage(nblk);
    const float* __restrict__ src = two ? fb.job[j].part + (int64_t)nblk * N : fb.job[j].part;
    const int nrows = two ? kFinSlices : nblk;
    float s = 0.f;
    for (int k = 0; k < nrows; ++k) s += src[(int64_t)k * N + c];
    float* __restrict__ out = fb.job[j].out;
    out[c] = fb.job[j].accumulate ? out[c] + s : s;
}

// ============================================================================================
// GELU (exact erf form, nn.GELU() default)
// ============================================================================================
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// branch-free Phi(x) (same A&S 7.1.26 erf as gelu_grad_fast below, |error| <= 7.5e-8): gelu(x) = x Phi(x)
__device__ __forceinline__ float gelu_fast(float x) {
    const float u = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, u, 1.f));
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);
    float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float erf_abs = __builtin_fmaf(-poly * t, e, 1.f);
    return x * __builtin_fmaf(0.5f, copysignf(erf_abs, x), 0.5f);
}

template <class E, int VARIANT>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const typename V<E>::x8* __restrict__ h, typename V<E>::x8* __restrict__ g, int64_t n8) {
    using e8 = typename V<E>::x8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const e8 v = h[i];
        e8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (E)(VARIANT == 0 ? gelu_f((float)v[j]) : gelu_fast((float)v[j]));
        g[i] = o;
    }
}

// column-partial reductions: a thread owns 8 consecutive columns and walks kColRows rows
static inline int col_rows(int T) {
    static const int forced = [] { const char* e = getenv("NPCD_COL_ROWS"); return e ? atoi(e) : 0; }();     // A/B probes only
    if (forced > 0) return forced;
    int r = T / 512;
    return r < 8 ? 8 : (r > 64 ? 64 : r);
}

// Branch-free gelu'(x) = Phi(x) + x phi(x) with erf from Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 absolute, far below the
// bf16 resolution of the result): ONE exponential e = exp(-x^2/2) serves both the erf tail and the density, one reciprocal,
// a degree-5 Horner chain -- ~16 vector instructions per element.  libm's erff is a three-way branch per element (both sides
// executed under the exec mask whenever a wave holds small and large |x|, i.e. always): ~59 instructions per element, which made
// this kernel co-limited by vector-instruction issue (profiles/r2_elementwise_sq_pmc.json).
__device__ __forceinline__ float gelu_grad_fast(float x) {
    const float u = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, u, 1.f));
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170f);          // exp(-x^2 / 2)
    float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    const float erf_abs = __builtin_fmaf(-poly * t, e, 1.f);                         // erf(|x| / sqrt 2)
    const float phi_cdf = __builtin_fmaf(0.5f, copysignf(erf_abs, x), 0.5f);
    return __builtin_fmaf(x * 0.3989422804014327f, e, phi_cdf);
}

// VARIANT (GELU only): 0 = libm erff; 1 = gelu_grad_fast; 2 = gelu_grad_fast, two rows per trip with all four loads issued first;
// 3 = gelu_grad_fast with non-temporal loads of the two once-read inputs
template <class E, bool GELU, int VARIANT = 1>
__global__ __launch_bounds__(256) void colsum_kernel(const E* __restrict__ a, const E* __restrict__ hpre, E* __restrict__ out,
                                                     float* __restrict__ part, int T, int N, int rows) {
    using e8 = typename V<E>::x8;
    const int col = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (col >= N) return;
    // workgroup y takes rows y, y + G, y + 2G, ... (G = gridDim.y): at any moment the grid reads ONE contiguous band of G rows.
    // (Consecutive rows per workgroup made all workgroups touch addresses that are equal modulo rows * N * 2 bytes at the same
    // time -- the same few HBM channels: 4.9 TB/s instead of the 6.0 TB/s of the plain streaming kernels.)
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    auto one = [&](const e8 v, const e8 hp, int64_t off) {
        e8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gp = VARIANT == 0 ? gelu_grad_f((float)hp[j]) : gelu_grad_fast((float)hp[j]);
            o[j] = (E)((float)v[j] * gp);
            acc[j] += (float)o[j];      // the bias gradient sums the SAME rounded values the GEMMs see
        }
        *reinterpret_cast<e8*>(out + off) = o;
    };
    if (GELU && VARIANT == 2) {
        int rr = 0;
        for (; rr + 1 < rows; rr += 2) {
            const int r0 = rr * gridDim.y + blockIdx.y, r1 = r0 + gridDim.y;
            if (r1 >= T) break;
            const int64_t o0 = (int64_t)r0 * N + col, o1 = (int64_t)r1 * N + col;
            const e8 v0 = *reinterpret_cast<const e8*>(a + o0), h0 = *reinterpret_cast<const e8*>(hpre + o0);
            const e8 v1 = *reinterpret_cast<const e8*>(a + o1), h1 = *reinterpret_cast<const e8*>(hpre + o1);
            one(v0, h0, o0);
            one(v1, h1, o1);
        }
        for (; rr < rows; ++rr) {
            const int row = rr * gridDim.y + blockIdx.y;
            if (row >= T) break;
            const int64_t off = (int64_t)row * N + col;
            one(*reinterpret_cast<const e8*>(a + off), *reinterpret_cast<const e8*>(hpre + off), off);
        }
    } else {
        for (int rr = 0; rr < rows; ++rr) {
            const int row = rr * gridDim.y + blockIdx.y;
            if (row >= T) break;
            const int64_t off = (int64_t)row * N + col;
            const e8 v = (GELU && VARIANT == 3) ? __builtin_nontemporal_load(reinterpret_cast<const e8*>(a + off))
                                                    : *reinterpret_cast<const e8*>(a + off);
            if (GELU) {
                one(v, VARIANT == 3 ? __builtin_nontemporal_load(reinterpret_cast<const e8*>(hpre + off))
                                    : *reinterpret_cast<const e8*>(hpre + off), off);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
            }
        }
    }
    float* p = part + (int64_t)blockIdx.y * N + col;
    *reinterpret_cast<f32x4*>(p) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

// Weight gradient of a Linear layer with a HANDFUL of outputs (the heads' last layers of the PointNeRF field: 256 -> 1, 256 -> 3):
// dW[j][k] = sum_p dy[p][j] x[p][k], J <= 4 rows.  As a GEMM this is 1-3 output rows for ~2 x 10^5 reduction steps and the
// library spends 350 us on one workgroup; it is one pass over x: 256 threads = (rows of a band) x (16-byte column chunks), fp32
// sums per thread, the row lanes added through LDS, one partial row of J K sums per workgroup, finished by the column-sum
// finalisation (fixed order).  bf16 operands, fp32 accumulation and output.
template <int J>
__global__ __launch_bounds__(256) void small_wgrad_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ x, float* __restrict__ part,
                                                          int T, int K, int rows_per_block) {
    extern __shared__ float swg_red[];                      // [row lanes][J][K]
    const int chunks = K >> 3, lanes = 256 / chunks;
    const int cc = threadIdx.x % chunks, rl = threadIdx.x / chunks;
    const int row0 = blockIdx.x * rows_per_block, row1 = min(T, row0 + rows_per_block);
    float acc[J][8];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
    for (int row = row0 + rl; row < row1; row += lanes) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (int64_t)row * K + 8 * cc);
        float w[J];
#pragma unroll
        for (int j = 0; j < J; ++j) w[j] = (float)dy[(int64_t)row * J + j];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j][i] = __builtin_fmaf(w[j], (float)v[i], acc[j][i]);
    }
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) swg_red[(rl * J + j) * K + 8 * cc + i] = acc[j][i];
    __syncthreads();
    for (int e = threadIdx.x; e < J * K; e += 256) {
        float sum = 0.f;
        for (int r2 = 0; r2 < lanes; ++r2) sum += swg_red[r2 * J * K + e];       // row lanes in order
        part[(int64_t)blockIdx.x * J * K + e] = sum;
    }
}

// ============================================================================================
// AdamW (torch.optim.AdamW semantics, amsgrad=False, maximize=False) + EMA + bf16 shadow + grad zeroing
//   p *= 1 - lr*wd;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
//   p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps);  ema = ema + (1-decay) (p - ema)   [lerp]
// ============================================================================================
struct AdamArgs {
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, ema_w;
};

__device__ __forceinline__ void adamw_one(f32x4& pp, const f32x4 gg, f32x4& mm, f32x4& vv, const AdamArgs& a) {
    pp *= (1.f - a.lr * a.wd);
    mm = mm * a.beta1 + gg * (1.f - a.beta1);
    vv = vv * a.beta2 + gg * gg * (1.f - a.beta2);
    const float step = a.lr / a.bc1;
#pragma unroll
    for (int j = 0; j < 4; ++j) pp[j] -= step * (mm[j] / (sqrtf(vv[j]) / a.bc2_sqrt + a.eps));
}

// Two float4 per thread and trip: all ten loads of a trip are issued before the first dependent use (the kernel is a pure
// stream of 5 reads + 6 writes per element; more bytes in flight per wave is the only lever).
template <class E>
__global__ __launch_bounds__(256) void adamw_ema_kernel(f32x4* __restrict__ p, f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v,
                                                        f32x4* __restrict__ ema, typename V<E>::x4* __restrict__ shadow, int64_t n4, AdamArgs a, int zero_grad) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
        const int64_t k = i + stride;
        f32x4 p0 = p[i], p1 = p[k];
        const f32x4 g0 = g[i], g1 = g[k];
        f32x4 m0 = m[i], m1 = m[k], v0 = v[i], v1 = v[k];
        f32x4 e0 = {0.f, 0.f, 0.f, 0.f}, e1 = e0;
        if (ema) { e0 = ema[i]; e1 = ema[k]; }
        adamw_one(p0, g0, m0, v0, a);
        adamw_one(p1, g1, m1, v1, a);
        p[i] = p0; p[k] = p1;
        m[i] = m0; m[k] = m1;
        v[i] = v0; v[k] = v1;
        if (ema) { ema[i] = e0 + (p0 - e0) * a.ema_w; ema[k] = e1 + (p1 - e1) * a.ema_w; }
        if (shadow) { shadow[i] = from_f32<E>(p0); shadow[k] = from_f32<E>(p1); }
        if (zero_grad) { g[i] = f32x4{0.f, 0.f, 0.f, 0.f}; g[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    if (i < n4) {
        f32x4 pp = p[i];
        const f32x4 gg = g[i];
        f32x4 mm = m[i], vv = v[i];
        adamw_one(pp, gg, mm, vv, a);
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
        if (ema) {
            const f32x4 e = ema[i];
            ema[i] = e + (pp - e) * a.ema_w;
        }
        if (shadow) shadow[i] = from_f32<E>(pp);
        if (zero_grad) g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

template <class E>
__global__ __launch_bounds__(256) void cast_kernel(const f32x4* __restrict__ src, typename V<E>::x4* __restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) dst[i] = from_f32<E>(src[i]);
}

// out[i] = part[0][i] + part[1][i] + ... + part[S - 1][i] (fp32, added in slice order): the row-split weight-gradient GEMMs of the
// fused backbone produce S partial [N, K] matrices.  All S loads of an element are issued before the first add; one 16-byte piece
// per thread and iteration.  HBM-bound: (S + 1) x 4 bytes per element.
template <int S>
__global__ __launch_bounds__(256) void sum_slices_kernel(const f32x4* __restrict__ part, f32x4* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 v[S];
#pragma unroll
        for (int s2 = 0; s2 < S; ++s2) v[s2] = __builtin_nontemporal_load(part + s2 * n4 + i);
        f32x4 acc = v[0];
#pragma unroll
        for (int s2 = 1; s2 < S; ++s2) acc += v[s2];
        out[i] = acc;
    }
}

// DDPM reverse step (gaussian_diffusion.py:100-146 of the reference): x0 = a x_t - b eps (clamped), mean = c1 x0 + c2 x_t,
// x_{t-1} = mean + [t > 0] exp(logvar / 2) noise; the five per-sample coefficients are looked up from the 1000-entry tables.
struct DdpmTables {
    const float *recip, *recipm1, *c1, *c2, *logvar;
};
template <class EPS>
__global__ __launch_bounds__(256) void ddpm_reverse_kernel(const float* __restrict__ x_t, const EPS* __restrict__ eps, const float* __restrict__ noise,
                                                           float* __restrict__ x_prev, float* __restrict__ x0_out, const int64_t* __restrict__ t,
                                                           int64_t per_sample, DdpmTables tab, float lo, float hi, int has_clip) {
    const int b = blockIdx.y;
    const int64_t ts = t[b];
    const float a = tab.recip[ts], bb = tab.recipm1[ts], c1 = tab.c1[ts], c2 = tab.c2[ts];
    const float sd = ts != 0 ? expf(0.5f * tab.logvar[ts]) : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * 256) {
        const int64_t j = (int64_t)b * per_sample + i;
        const float x = x_t[j];
        float x0 = a * x - bb * (float)eps[j];
        if (has_clip) x0 = fminf(fmaxf(x0, lo), hi);
        if (x0_out) x0_out[j] = x0;
        x_prev[j] = (c1 * x0 + c2 * x) + sd * noise[j];
    }
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace npcd

using namespace npcd;

// rows are interleaved over the waves of a grid of at most 2048 workgroups (8 per CU)
static inline int ln_fwd_blocks(int T) { const int b = (T + 3) / 4; return b < 2048 ? b : 2048; }

// `dtype` (NPCD_BF16 / NPCD_F16) = the 16-bit activation type of the run; the names without _dt are the bf16 forms
static inline bool dt16(int dtype) { return dtype == NPCD_BF16 || dtype == NPCD_F16; }

extern "C" int npcd_add_ln_fwd_dt(const float* x_in, const void* delta, const float* gamma, const float* beta, float* x_out, void* y,
                                  float* mean, float* rstd, int T, int W, float eps, int dtype, void* stream) {
    if (!x_in || !gamma || !beta || !y || !mean || !rstd || T <= 0 || W <= 0) return NPCD_ERR_ARG;
    if (W % 4 != 0 || W > 256 * kMaxChunks || !dt16(dtype)) return NPCD_ERR_UNSUPPORTED;
    if (!al16(x_in) || !al16(gamma) || !al16(beta) || (x_out && !al16(x_out)) || (reinterpret_cast<uintptr_t>(y) & 7) ||
        (delta && (reinterpret_cast<uintptr_t>(delta) & 7)))
        return NPCD_ERR_ARG;
#define NPCD_LAUNCH_LN_FWD(E, NCH)                                                                                            \
    hipLaunchKernelGGL((add_ln_fwd_kernel<E, NCH>), dim3(ln_fwd_blocks(T)), dim3(256), 0, static_cast<hipStream_t>(stream), x_in, \
                       static_cast<const E*>(delta), gamma, beta, x_out, static_cast<E*>(y), mean, rstd, T, W, eps)
#define NPCD_LN_FWD_W(E)                     \
    do {                                     \
        if (W <= 256) NPCD_LAUNCH_LN_FWD(E, 1);      \
        else if (W <= 512) NPCD_LAUNCH_LN_FWD(E, 2); \
        else if (W <= 1024) NPCD_LAUNCH_LN_FWD(E, 4);\
        else NPCD_LAUNCH_LN_FWD(E, 8);               \
    } while (0)
    if (dtype == NPCD_BF16) NPCD_LN_FWD_W(__bf16);
    else NPCD_LN_FWD_W(_Float16);
#undef NPCD_LN_FWD_W
#undef NPCD_LAUNCH_LN_FWD
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
extern "C" int npcd_add_ln_fwd(const float* x_in, const void* delta, const float* gamma, const float* beta, float* x_out, void* y,
                               float* mean, float* rstd, int T, int W, float eps, void* stream) {
    return npcd_add_ln_fwd_dt(x_in, delta, gamma, beta, x_out, y, mean, rstd, T, W, eps, NPCD_BF16, stream);
}

extern "C" int npcd_ln_bwd_blocks(int T) { const int r = ln_rows_per_wave(T); return (T + 4 * r - 1) / (4 * r); }

extern "C" int npcd_ln_bwd_dt(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                              float* dx, void* dxb, float* part_gamma, float* part_beta, float* part_col, int T, int W, int dtype, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || T <= 0 || W <= 0) return NPCD_ERR_ARG;
    if (W % 4 != 0 || W > 256 * kMaxChunks || !dt16(dtype)) return NPCD_ERR_UNSUPPORTED;
    if (!al16(x) || !al16(gamma) || !al16(dx) || (dres && !al16(dres))) return NPCD_ERR_ARG;
    const int nblk = npcd_ln_bwd_blocks(T);
    const size_t lds = (size_t)3 * 4 * W * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    static DynLds lds_attr_b, lds_attr_h;   // W = 2048 needs 96 KiB of dynamic LDS
    if (dtype == NPCD_BF16) NPCD_HIP_CHECK(lds_attr_b.ensure(reinterpret_cast<const void*>(ln_bwd_kernel<__bf16, 8>), 3 * 4 * 2048 * 4));
    else NPCD_HIP_CHECK(lds_attr_h.ensure(reinterpret_cast<const void*>(ln_bwd_kernel<_Float16, 8>), 3 * 4 * 2048 * 4));
#define NPCD_LAUNCH_LN_BWD(E, NCH)                                                                                             \
    hipLaunchKernelGGL((ln_bwd_kernel<E, NCH>), dim3(nblk), dim3(256), lds, st, static_cast<const E*>(dy), x, mean, rstd, gamma, \
                       dres, dx, static_cast<E*>(dxb), part_gamma, part_beta, part_col, T, W, ln_rows_per_wave(T))
#define NPCD_LN_BWD_W(E)                     \
    do {                                     \
        if (W <= 256) NPCD_LAUNCH_LN_BWD(E, 1);      \
        else if (W <= 512) NPCD_LAUNCH_LN_BWD(E, 2); \
        else if (W <= 1024) NPCD_LAUNCH_LN_BWD(E, 4);\
        else NPCD_LAUNCH_LN_BWD(E, 8);               \
    } while (0)
    if (dtype == NPCD_BF16) NPCD_LN_BWD_W(__bf16);
    else NPCD_LN_BWD_W(_Float16);
#undef NPCD_LN_BWD_W
#undef NPCD_LAUNCH_LN_BWD
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
extern "C" int npcd_ln_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                           float* dx, void* dxb, float* part_gamma, float* part_beta, float* part_col, int T, int W, void* stream) {
    return npcd_ln_bwd_dt(dy, x, mean, rstd, gamma, dres, dx, dxb, part_gamma, part_beta, part_col, T, W, NPCD_BF16, stream);
}

// One-launch form (round 6): a workgroup of 1,024 threads = 64 columns x 16 row lanes; lane r adds rows r, r + 16, ... of its column
// (two accumulators), the 16 lane sums are added through LDS in a fixed tree.  One launch instead of two, no stage buffer; the
// summation order differs from the two-stage form (both are fixed: bitwise reproducible either way).  For the token counts of a rank
// of the strong-scaling job, where the finalisation of a block's eight sums sits on the backward's critical path (16 us of ~400).
__global__ __launch_bounds__(1024) void colsum_onepass_kernel(FinBatch fb) {
    __shared__ float red[16][64];
    const int j = fin_job(fb.first1, fb.njobs, blockIdx.x);
    const float* __restrict__ part = fb.job[j].part;
    const int nblk = fb.job[j].nblk, N = fb.job[j].N;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = (blockIdx.x - fb.first1[j]) * 64 + tx;
    float s0 = 0.f, s1 = 0.f;
    if (c < N) {
        int b = ty;
        for (; b + 16 < nblk; b += 32) {
            s0 += part[(int64_t)b * N + c];
            s1 += part[(int64_t)(b + 16) * N + c];
        }
        if (b < nblk) s0 += part[(int64_t)b * N + c];
    }
    red[ty][tx] = s0 + s1;
    __syncthreads();
    if (ty == 0 && c < N) {
        float q[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) q[g] = (red[4 * g][tx] + red[4 * g + 1][tx]) + (red[4 * g + 2][tx] + red[4 * g + 3][tx]);
        const float s = (q[0] + q[1]) + (q[2] + q[3]);
        float* __restrict__ out = fb.job[j].out;
        out[c] = fb.job[j].accumulate ? out[c] + s : s;
    }
}

// `part` must have room for kFinSlices extra rows after its nblk rows (npcd_colsum_scratch_rows()):
// they are used as the stage buffer.
extern "C" int npcd_colsum_scratch_rows(void) { return kFinSlices; }

extern "C" int npcd_colsum_finalize_batch(const NpcdColsumJob* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0 || njobs > NPCD_COLSUM_MAX_JOBS) return NPCD_ERR_ARG;
    FinBatch fb;
    fb.njobs = njobs;
    bool any_two_stage = false;
    int n1 = 0, n2 = 0;
    for (int j = 0; j < njobs; ++j) {
        if (!jobs[j].part || !jobs[j].out || jobs[j].nblk <= 0 || jobs[j].N <= 0) return NPCD_ERR_ARG;
        fb.job[j] = jobs[j];
        fb.first1[j] = n1;
        fb.first2[j] = n2;
        n1 += (jobs[j].N + 63) / 64;
        n2 += (jobs[j].N + 255) / 256;
        any_two_stage |= jobs[j].nblk > 2 * kFinSlices;
    }
    for (int j = njobs; j <= NPCD_COLSUM_MAX_JOBS; ++j) { fb.first1[j] = n1; fb.first2[j] = n2; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    // NPCD_COLSUM_ONEPASS=1: the one-launch form (A/B switch; default off).  Measured in the rank step at per-GPU batch 8, three
    // alternating rounds: 17.13 / 17.24 / 17.19 ms with the two launches, 17.14 / 17.38 / 17.05 with one; batch 64: 82.71 against 82.61 --
    // inside the noise, like deferring the finalisation to the end of the backward in round 5 (docs/experiments.md R5.19, R6.12).
    static const bool onepass = [] { const char* e = getenv("NPCD_COLSUM_ONEPASS"); return e && atoi(e) > 0; }();
    if (onepass) {
        hipLaunchKernelGGL(colsum_onepass_kernel, dim3(n1), dim3(1024), 0, st, fb);
        NPCD_HIP_CHECK(hipGetLastError());
        return NPCD_OK;
    }
    if (any_two_stage) hipLaunchKernelGGL(colsum_stage1_kernel, dim3(n1, kFinSlices), dim3(256), 0, st, fb);
    hipLaunchKernelGGL(colsum_stage2_kernel, dim3(n2), dim3(256), 0, st, fb);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_colsum_finalize(const float* part, int nblk, int N, float* out, int accumulate, void* stream) {
    NpcdColsumJob job;
    job.part = part; job.out = out; job.nblk = nblk; job.N = N; job.accumulate = accumulate; job.reserved = 0;
    return npcd_colsum_finalize_batch(&job, 1, stream);
}

extern "C" int npcd_gelu_fwd_dt(const void* h, void* g, int64_t numel, int dtype, void* stream) {
    if (!h || !g || numel <= 0) return NPCD_ERR_ARG;
    if (numel % 8 != 0 || !al16(h) || !al16(g) || !dt16(dtype)) return NPCD_ERR_UNSUPPORTED;
    const int64_t n8 = numel / 8;
    const int grid = (int)((n8 + 255) / 256 < 16384 ? (n8 + 255) / 256 : 16384);
    static const int variant = [] { const char* e = getenv("NPCD_GELU_FWD_VARIANT"); return e ? atoi(e) : 1; }();   // A/B probes only
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NPCD_LAUNCH_GELU_FWD(E, VV)                                                                                                 \
    hipLaunchKernelGGL((gelu_fwd_kernel<E, VV>), dim3(grid), dim3(256), 0, st, static_cast<const typename V<E>::x8*>(h), \
                       static_cast<typename V<E>::x8*>(g), n8)
    if (dtype == NPCD_BF16) {
        if (variant == 0) NPCD_LAUNCH_GELU_FWD(__bf16, 0);
        else NPCD_LAUNCH_GELU_FWD(__bf16, 1);
    } else {
        NPCD_LAUNCH_GELU_FWD(_Float16, 1);
    }
#undef NPCD_LAUNCH_GELU_FWD
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
extern "C" int npcd_gelu_fwd(const void* h, void* g, int64_t numel, void* stream) { return npcd_gelu_fwd_dt(h, g, numel, NPCD_BF16, stream); }

extern "C" int npcd_colsum_blocks(int T) { const int r = col_rows(T); return (T + r - 1) / r; }

// dh = dg * gelu'(h) (16 bit) and column partials of dh: part [npcd_colsum_blocks(T)][N]
extern "C" int npcd_gelu_bwd_dt(const void* dg, const void* h, void* dh, float* part, int T, int N, int dtype, void* stream) {
    if (!dg || !h || !dh || !part || T <= 0 || N <= 0) return NPCD_ERR_ARG;
    if (N % 8 != 0 || !al16(dg) || !al16(h) || !al16(dh) || !al16(part) || !dt16(dtype)) return NPCD_ERR_UNSUPPORTED;
    dim3 grid((N / 8 + 255) / 256, npcd_colsum_blocks(T));
    static const int variant = [] { const char* e = getenv("NPCD_GELU_BWD_VARIANT"); return e ? atoi(e) : 1; }();   // A/B probes only
#define NPCD_LAUNCH_GELU_BWD(E, VV)                                                                                                    \
    hipLaunchKernelGGL((colsum_kernel<E, true, VV>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const E*>(dg), \
                       static_cast<const E*>(h), static_cast<E*>(dh), part, T, N, col_rows(T))
    if (dtype == NPCD_F16) NPCD_LAUNCH_GELU_BWD(_Float16, 1);
    else if (variant == 0) NPCD_LAUNCH_GELU_BWD(__bf16, 0);
    else if (variant == 2) NPCD_LAUNCH_GELU_BWD(__bf16, 2);
    else if (variant == 3) NPCD_LAUNCH_GELU_BWD(__bf16, 3);
    else NPCD_LAUNCH_GELU_BWD(__bf16, 1);
#undef NPCD_LAUNCH_GELU_BWD
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
extern "C" int npcd_gelu_bwd(const void* dg, const void* h, void* dh, float* part, int T, int N, void* stream) {
    return npcd_gelu_bwd_dt(dg, h, dh, part, T, N, NPCD_BF16, stream);
}

extern "C" int npcd_colsum_dt(const void* a, float* part, int T, int N, int dtype, void* stream) {
    if (!a || !part || T <= 0 || N <= 0) return NPCD_ERR_ARG;
    if (N % 8 != 0 || !al16(a) || !al16(part) || !dt16(dtype)) return NPCD_ERR_UNSUPPORTED;
    dim3 grid((N / 8 + 255) / 256, npcd_colsum_blocks(T));
    if (dtype == NPCD_BF16)
        hipLaunchKernelGGL((colsum_kernel<__bf16, false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const __bf16*>(a),
                           static_cast<const __bf16*>(nullptr), static_cast<__bf16*>(nullptr), part, T, N, col_rows(T));
    else
        hipLaunchKernelGGL((colsum_kernel<_Float16, false>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const _Float16*>(a),
                           static_cast<const _Float16*>(nullptr), static_cast<_Float16*>(nullptr), part, T, N, col_rows(T));
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
extern "C" int npcd_colsum_bf16(const void* a, float* part, int T, int N, void* stream) { return npcd_colsum_dt(a, part, T, N, NPCD_BF16, stream); }

extern "C" int npcd_small_wgrad_blocks(int T) { return T <= 0 ? -1 : (T + 255) / 256 < 1024 ? (T + 255) / 256 : 1024; }

extern "C" int npcd_small_wgrad(const void* dy, const void* x, float* part, int T, int J, int K, void* stream) {
    if (!dy || !x || !part || T <= 0) return NPCD_ERR_ARG;
    if (J < 1 || J > 4 || K < 64 || K > 2048 || (K & (K - 1)) || !al16(x) || !al16(part)) return NPCD_ERR_UNSUPPORTED;
    const int nblk = npcd_small_wgrad_blocks(T), rpb = (T + nblk - 1) / nblk;
    const size_t lds = (size_t)(256 / (K / 8)) * J * K * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const __bf16* d = static_cast<const __bf16*>(dy);
    const __bf16* xx = static_cast<const __bf16*>(x);
    switch (J) {
        case 1: hipLaunchKernelGGL(small_wgrad_kernel<1>, dim3(nblk), dim3(256), lds, st, d, xx, part, T, K, rpb); break;
        case 2: hipLaunchKernelGGL(small_wgrad_kernel<2>, dim3(nblk), dim3(256), lds, st, d, xx, part, T, K, rpb); break;
        case 3: hipLaunchKernelGGL(small_wgrad_kernel<3>, dim3(nblk), dim3(256), lds, st, d, xx, part, T, K, rpb); break;
        default: hipLaunchKernelGGL(small_wgrad_kernel<4>, dim3(nblk), dim3(256), lds, st, d, xx, part, T, K, rpb); break;
    }
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_adamw_ema_dt(float* p, float* g, float* m, float* v, float* ema, void* shadow, int shadow_dtype, int64_t numel, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int step, float ema_decay, int zero_grad, void* stream) {
    if (!p || !g || !m || !v || numel <= 0 || step <= 0) return NPCD_ERR_ARG;
    if (numel % 4 != 0 || !al16(p) || !al16(g) || !al16(m) || !al16(v) || (ema && !al16(ema)) ||
        (shadow && ((reinterpret_cast<uintptr_t>(shadow) & 7) || !dt16(shadow_dtype))))
        return NPCD_ERR_UNSUPPORTED;
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    a.ema_w = 1.f - ema_decay;
    const int64_t n4 = numel / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (shadow && shadow_dtype == NPCD_F16)
        hipLaunchKernelGGL(adamw_ema_kernel<_Float16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<f32x4*>(p), reinterpret_cast<f32x4*>(g),
                           reinterpret_cast<f32x4*>(m), reinterpret_cast<f32x4*>(v), reinterpret_cast<f32x4*>(ema),
                           static_cast<V<_Float16>::x4*>(shadow), n4, a, zero_grad);
    else
        hipLaunchKernelGGL(adamw_ema_kernel<__bf16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<f32x4*>(p), reinterpret_cast<f32x4*>(g),
                           reinterpret_cast<f32x4*>(m), reinterpret_cast<f32x4*>(v), reinterpret_cast<f32x4*>(ema),
                           static_cast<V<__bf16>::x4*>(shadow), n4, a, zero_grad);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
extern "C" int npcd_adamw_ema(float* p, float* g, float* m, float* v, float* ema, void* shadow_bf16, int64_t numel, float lr, float beta1,
                              float beta2, float eps, float weight_decay, int step, float ema_decay, int zero_grad, void* stream) {
    return npcd_adamw_ema_dt(p, g, m, v, ema, shadow_bf16, NPCD_BF16, numel, lr, beta1, beta2, eps, weight_decay, step, ema_decay, zero_grad, stream);
}

extern "C" int npcd_ddpm_reverse_step(const float* x_t, const void* eps, int eps_dtype, const float* noise, float* x_prev, float* x0_out,
                                      const int64_t* t, int B, int64_t per_sample, const float* tab_recip, const float* tab_recipm1,
                                      const float* tab_coef1, const float* tab_coef2, const float* tab_logvar, float clip_lo, float clip_hi,
                                      int has_clip, void* stream) {
    if (!x_t || !eps || !noise || !x_prev || !t || B <= 0 || per_sample <= 0) return NPCD_ERR_ARG;
    if (!tab_recip || !tab_recipm1 || !tab_coef1 || !tab_coef2 || !tab_logvar) return NPCD_ERR_ARG;
    if (eps_dtype != NPCD_F32 && eps_dtype != NPCD_BF16) return NPCD_ERR_UNSUPPORTED;
    const DdpmTables tab{tab_recip, tab_recipm1, tab_coef1, tab_coef2, tab_logvar};
    const int gx = (int)((per_sample + 255) / 256 < 1024 ? (per_sample + 255) / 256 : 1024);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (eps_dtype == NPCD_F32)
        hipLaunchKernelGGL(ddpm_reverse_kernel<float>, dim3(gx, B), dim3(256), 0, st, x_t, static_cast<const float*>(eps), noise, x_prev, x0_out, t,
                           per_sample, tab, clip_lo, clip_hi, has_clip);
    else
        hipLaunchKernelGGL(ddpm_reverse_kernel<__bf16>, dim3(gx, B), dim3(256), 0, st, x_t, static_cast<const __bf16*>(eps), noise, x_prev, x0_out, t,
                           per_sample, tab, clip_lo, clip_hi, has_clip);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

// ============================================================================================
// forward process + training loss of the DDPM (gaussian_diffusion.py:68-76, 199-230)
//   q_sample:  x_t = sqrt(acp[t_b]) x_0 + sqrt(1 - acp[t_b]) noise     (per-sample coefficients looked up on the device)
//   eps-MSE :  loss = mean((noise - eps_hat)^2 / 2);   d loss / d eps_hat = -(noise - eps_hat) / numel * upstream
// The reference runs them as ~12 elementwise / reduction launches per tensor; here one launch per tensor and direction, with a
// fixed-order two-stage sum (bitwise reproducible loss).
// ============================================================================================
namespace npcd {
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const int64_t* __restrict__ t,
                                                       const float* __restrict__ tab_a, const float* __restrict__ tab_s, float* __restrict__ xt,
                                                       int64_t per_sample) {
#pragma clang fp contract(off)      // mul, mul, add, each rounded, like the reference's eager ops: no fused multiply-add here
    const int b = blockIdx.y;
    const float ca = tab_a[t[b]], cs = tab_s[t[b]];
    const int64_t base = (int64_t)b * per_sample;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * 256) {
        const float p0 = ca * x0[base + i], p1 = cs * noise[base + i];
        xt[base + i] = p0 + p1;
    }
}

constexpr int kMseBlocks = 256;
template <class T>
__global__ __launch_bounds__(256) void eps_mse_fwd_kernel(const float* __restrict__ noise, const T* __restrict__ eps, float* __restrict__ pointwise,
                                                          float* __restrict__ part, int64_t numel) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        const float d = noise[i] - (float)eps[i];
        const float pw = d * d * 0.5f;
        if (pointwise) pointwise[i] = pw;
        s += pw;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void eps_mse_finalize_kernel(const float* __restrict__ part, int nblocks, float inv_numel, float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 64) s += part[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) out[0] = s * inv_numel;
}
template <class T>
__global__ __launch_bounds__(256) void eps_mse_bwd_kernel(const float* __restrict__ noise, const T* __restrict__ eps, const float* __restrict__ upstream,
                                                          float inv_numel, T* __restrict__ grad, int64_t numel) {
    const float g = upstream[0] * inv_numel;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256)
        grad[i] = (T)(-(noise[i] - (float)eps[i]) * g);
}
}  // namespace npcd

extern "C" int npcd_q_sample(const float* x0, const float* noise, const int64_t* t, const float* tab_sqrt_acp, const float* tab_sqrt_1macp,
                             float* x_t, int B, int64_t per_sample, void* stream) {
    if (!x0 || !noise || !t || !tab_sqrt_acp || !tab_sqrt_1macp || !x_t || B <= 0 || per_sample <= 0) return NPCD_ERR_ARG;
    const int gx = (int)((per_sample + 255) / 256 < 256 ? (per_sample + 255) / 256 : 256);
    hipLaunchKernelGGL(q_sample_kernel, dim3(gx, B), dim3(256), 0, static_cast<hipStream_t>(stream), x0, noise, t, tab_sqrt_acp, tab_sqrt_1macp, x_t,
                       per_sample);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_eps_mse_blocks(void) { return kMseBlocks; }

extern "C" int npcd_eps_mse_fwd(const float* noise, const void* eps, int eps_dtype, int64_t numel, float* pointwise, float* part, float* loss,
                                void* stream) {
    if (!noise || !eps || !part || !loss || numel <= 0) return NPCD_ERR_ARG;
    if (eps_dtype != NPCD_F32 && eps_dtype != NPCD_BF16) return NPCD_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (eps_dtype == NPCD_F32)
        hipLaunchKernelGGL(eps_mse_fwd_kernel<float>, dim3(kMseBlocks), dim3(256), 0, st, noise, static_cast<const float*>(eps), pointwise, part, numel);
    else
        hipLaunchKernelGGL(eps_mse_fwd_kernel<__bf16>, dim3(kMseBlocks), dim3(256), 0, st, noise, static_cast<const __bf16*>(eps), pointwise, part, numel);
    hipLaunchKernelGGL(eps_mse_finalize_kernel, dim3(1), dim3(64), 0, st, part, kMseBlocks, 1.f / (float)numel, loss);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_eps_mse_bwd(const float* noise, const void* eps, int eps_dtype, int64_t numel, const float* upstream_dev, void* grad,
                                void* stream) {
    if (!noise || !eps || !upstream_dev || !grad || numel <= 0) return NPCD_ERR_ARG;
    if (eps_dtype != NPCD_F32 && eps_dtype != NPCD_BF16) return NPCD_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = (int)((numel + 255) / 256 < 2048 ? (numel + 255) / 256 : 2048);
    if (eps_dtype == NPCD_F32)
        hipLaunchKernelGGL(eps_mse_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, noise, static_cast<const float*>(eps), upstream_dev,
                           1.f / (float)numel, static_cast<float*>(grad), numel);
    else
        hipLaunchKernelGGL(eps_mse_bwd_kernel<__bf16>, dim3(grid), dim3(256), 0, st, noise, static_cast<const __bf16*>(eps), upstream_dev,
                           1.f / (float)numel, static_cast<__bf16*>(grad), numel);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_sum_slices(const float* part, float* out, int S, int64_t numel, void* stream) {
    if (!part || !out || numel <= 0) return NPCD_ERR_ARG;
    if ((S != 2 && S != 4 && S != 8) || numel % 4 != 0 || !al16(part) || !al16(out)) return NPCD_ERR_UNSUPPORTED;
    const int64_t n4 = numel / 4;
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const f32x4* p4 = reinterpret_cast<const f32x4*>(part);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    if (S == 2) hipLaunchKernelGGL(sum_slices_kernel<2>, dim3(grid), dim3(256), 0, st, p4, o4, n4);
    else if (S == 4) hipLaunchKernelGGL(sum_slices_kernel<4>, dim3(grid), dim3(256), 0, st, p4, o4, n4);
    else hipLaunchKernelGGL(sum_slices_kernel<8>, dim3(grid), dim3(256), 0, st, p4, o4, n4);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_cast_f32_dt(const float* src, void* dst, int64_t numel, int dtype, void* stream) {
    if (!src || !dst || numel <= 0) return NPCD_ERR_ARG;
    if (numel % 4 != 0 || !al16(src) || (reinterpret_cast<uintptr_t>(dst) & 7) || !dt16(dtype)) return NPCD_ERR_UNSUPPORTED;
    const int64_t n4 = numel / 4;
    const int grid = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == NPCD_BF16)
        hipLaunchKernelGGL(cast_kernel<__bf16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const f32x4*>(src), static_cast<V<__bf16>::x4*>(dst), n4);
    else
        hipLaunchKernelGGL(cast_kernel<_Float16>, dim3(grid), dim3(256), 0, st, reinterpret_cast<const f32x4*>(src), static_cast<V<_Float16>::x4*>(dst), n4);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
extern "C" int npcd_cast_f32_bf16(const float* src, void* dst, int64_t numel, void* stream) { return npcd_cast_f32_dt(src, dst, numel, NPCD_BF16, stream); }
