// The point-level layers of the PointNeRF field in the reference's fp32 numerics class, fused (round 5).
//
// Reference op chain replaced (eval_pointnerf.py / eval_diffusion.py run the field in plain fp32):
//   last aggregator layer (linear)        npcd/models/pointnerf/fields/aggregators/mlp.py:83-84 (local_field[8])
//   shape_net + softplus(x - 1)           fields/mlp.py:38-51, field.py:30,126-128
//   channel_net + sigmoid                 fields/mlp.py:53-72, field.py:139-140
//
// `PointNeRF.render(mlp_dtype=torch.float32)` ran these six 256 x 256 layers as fp32 library GEMMs on the ~75 k shading points of a
// view: ~0.8 ms of its 1.61 ms.  Here every operand is two bf16 halves (x = hi + lo) and every product three matrix instructions
// hi.hi + hi.lo + lo.hi in fp32 accumulators -- the numerics of csrc/pairs_mlp.hip's precision 1 (~4e-6 relative per layer, fp32's
// exponent range) -- on v_mfma_f32_16x16x32_bf16 (docs/experiments.md R5.13: the instruction under which the power-bound chip
// clocks highest).  Structure of the fp16 point kernel's 16x16x32 form (csrc/shade.hip, points_pass16): weights as the A operand
// straight from L2 in fragment order (4 KiB runs per wave and 32 input channels, hi and lo), activations as the B operand from two
// LDS planes (hi, lo; 64 points x 528 bytes each), biases as the C operand of a layer's first step from an LDS copy, the heads' final
// 256 -> 1 / 256 -> 3 projections from the fp32 accumulators.  Tile = 64 points, 4 waves x 64 output channels, two workgroups per CU.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace npcd {

constexpr int kXHidden = 256;
constexpr int kXRows = 64;                       // points per tile
constexpr int kXPitch = 528;                     // bytes per row of a plane (512 + 16: ds_read_b128 of 16 consecutive rows is conflict-free)
constexpr int kXPlane = kXRows * kXPitch;
constexpr int kXFrag = 1024;
constexpr int kXMat = kXHidden * kXHidden * 2;   // bytes of one half (hi or lo) of a matrix
constexpr float kXLeaky = 0.01f;

struct PointsX2Layout {
    int64_t w[6];        // per matrix: [hi: kXMat bytes][lo: kXMat bytes], fragment order [wave][32-column step][block mb][lane][8]
    int64_t bias;        // 6 x 256 fp32 in the order of the C operands: [layer][wave][g][mb][4]
    int64_t s1, c4;      // fp32: s1 = [256 w | 1 b | pad], c4 = [3 x 256 w | 3 b | pad]
    int64_t total;
};
__host__ __device__ inline PointsX2Layout points_x2_layout() {
    PointsX2Layout L;
    int64_t off = 0;
    for (int i = 0; i < 6; ++i) { L.w[i] = off; off += 2 * kXMat; }
    L.bias = off; off += 6 * kXHidden * 4;
    L.s1 = off; off += 264 * 4;
    L.c4 = off; off += 776 * 4;
    L.total = off;
    return L;
}

typedef float f32x4a __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t xrsrc_t;
__device__ __forceinline__ f32x4a xmfma(bf16x8 a, bf16x8 b, f32x4a c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
// KS = 32-column steps of the layer (8 for a 256-wide input); the lo half of a matrix lies KS * 16 KiB behind its hi half
template <int KS = 8>
__device__ __forceinline__ bf16x8 xfrag(xrsrc_t rs, int w_off, int wave, int lane, int s, int mb) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + mb * kXFrag, w_off + (wave * KS + s) * (4 * kXFrag), 0));
}
struct XRing { bf16x8 h[2][4], l[2][4]; };
template <int KS = 8>
__device__ __forceinline__ void xprefetch(xrsrc_t rs, int w_off, int wave, int lane, XRing& ring) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
        ring.h[0][mb] = xfrag<KS>(rs, w_off, wave, lane, 0, mb);
        ring.l[0][mb] = xfrag<KS>(rs, w_off + KS * 16384, wave, lane, 0, mb);
    }
}
// one layer on a tile: acc[mb][rb] = bias + sum over 256 inputs of (Wh Xh + Wh Xl + Wl Xh); 16 independent accumulators between two
// instructions on the same one
template <int KS = 8, int NB16 = 4>
__device__ __forceinline__ void xlayer(const unsigned char* Hh, const unsigned char* Hl, const unsigned char* bias_l, xrsrc_t rs, int w_off,
                                       int wave, int lane, XRing& ring, f32x4a (&acc)[4][4]) {
    const int off = (lane & 15) * kXPitch + (lane >> 4) * 16;
    f32x4a init[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) init[mb] = *reinterpret_cast<const f32x4a*>(bias_l + wave * 256 + (lane >> 4) * 64 + mb * 16);
    bf16x8 bh[4], bl[4], nh[4], nl[4];
#pragma unroll
    for (int r = 0; r < NB16; ++r) {
        bh[r] = *reinterpret_cast<const bf16x8*>(Hh + off + r * 16 * kXPitch);
        bl[r] = *reinterpret_cast<const bf16x8*>(Hl + off + r * 16 * kXPitch);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                ring.h[(s + 1) & 1][mb] = xfrag<KS>(rs, w_off, wave, lane, s + 1, mb);
                ring.l[(s + 1) & 1][mb] = xfrag<KS>(rs, w_off + KS * 16384, wave, lane, s + 1, mb);
            }
#pragma unroll
            for (int r = 0; r < NB16; ++r) {
                nh[r] = *reinterpret_cast<const bf16x8*>(Hh + off + r * 16 * kXPitch + (s + 1) * 64);
                nl[r] = *reinterpret_cast<const bf16x8*>(Hl + off + r * 16 * kXPitch + (s + 1) * 64);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < NB16; ++r)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb][r] = xmfma(ring.h[s & 1][mb], bh[r], s == 0 ? init[mb] : acc[mb][r]);
#pragma unroll
        for (int r = 0; r < NB16; ++r)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb][r] = xmfma(ring.h[s & 1][mb], bl[r], acc[mb][r]);
#pragma unroll
        for (int r = 0; r < NB16; ++r)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[mb][r] = xmfma(ring.l[s & 1][mb], bh[r], acc[mb][r]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < KS) {
#pragma unroll
            for (int r = 0; r < NB16; ++r) { bh[r] = nh[r]; bl[r] = nl[r]; }
        }
    }
}
__device__ __forceinline__ uint32_t xpack2(__bf16 a, __bf16 b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 v = {a, b};
    return __builtin_bit_cast(uint32_t, v);
}
// write-back: optional LeakyReLU in fp32, then the two halves of every value to the two planes (8 consecutive channels per store:
// blocks 2 p, 2 p + 1 of a lane, the pack's row order)
template <bool ACT, int NB16 = 4>
__device__ __forceinline__ void xstore(unsigned char* Hh, unsigned char* Hl, int wave, int lane, const f32x4a (&acc)[4][4]) {
    const int sb = (lane & 15) * kXPitch + (lane >> 4) * 16 + wave * 128;
#pragma unroll
    for (int rb = 0; rb < NB16; ++rb)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            uint32_t vh[4], vl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x0 = acc[2 * p2 + (q >> 1)][rb][2 * (q & 1)], x1 = acc[2 * p2 + (q >> 1)][rb][2 * (q & 1) + 1];
                if (ACT) {
                    x0 = x0 > 0.f ? x0 : kXLeaky * x0;
                    x1 = x1 > 0.f ? x1 : kXLeaky * x1;
                }
                const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
                vh[q] = xpack2(h0, h1);
                vl[q] = xpack2((__bf16)(x0 - (float)h0), (__bf16)(x1 - (float)h1));
            }
            *reinterpret_cast<u32x4*>(Hh + sb + rb * 16 * kXPitch + p2 * 64) = u32x4{vh[0], vh[1], vh[2], vh[3]};
            *reinterpret_cast<u32x4*>(Hl + sb + rb * 16 * kXPitch + p2 * 64) = u32x4{vl[0], vl[1], vl[2], vl[3]};
        }
}
// the same values in fp32, straight from the accumulators, to a row-major [rows][256] array (training: the backward's activations)
template <bool ACT>
__device__ __forceinline__ void xsave(float* dst, int row0, int P, int wave, int lane, const f32x4a (&acc)[4][4]) {
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const int p = row0 + rb * 16 + n;
        if (p < P) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                f32x4 v;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float x = acc[mb][rb][b];
                    v[b] = (ACT && !(x > 0.f)) ? kXLeaky * x : x;
                }
                *reinterpret_cast<f32x4*>(dst + (int64_t)p * kXHidden + 64 * wave + 32 * (mb >> 1) + 8 * g + 4 * (mb & 1)) = v;
            }
        }
    }
}
__device__ __forceinline__ float xsum_groups(float x) {          // x summed over lanes l, l ^ 16, l ^ 32, l ^ 48
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}
__device__ __forceinline__ float xsoftplus_m1(float x) {
    x -= 1.f;
    return x > 20.f ? x : log1pf(expf(x));  // F.softplus(beta=1, threshold=20)
}

struct PointsX2Args {
    const unsigned char* wpack;
    const float* feat;            // [max_points][256] fp32: the aggregated per-point features (output of the per-pair layers)
    const int32_t* n_points;      // device-side count (may be null: max_points)
    int max_points;
    float *sigma, *rgb;
    const float* dir_bias;        // use_view_dir: [n_rays][256] fp32 rows added to the first colour layer's pre-activation, or null
    const int32_t* point_ray;
    // training forward (SAVE): the six hidden activations [6][max_points][256] fp32 -- feat, s0, c0, c1, c2, c3 (after LeakyReLU where
    // the layer has one) -- and the heads' PRE-activations [max_points][4] = (r, g, b, sigma), biases added; sigma / rgb are not written
    float* save;
    float* pre;
};

template <bool DIR, bool SAVE = false>
__global__ __launch_bounds__(256, 2) void points_x2_kernel(PointsX2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    unsigned char* Hh = dsmem;
    unsigned char* Hl = dsmem + kXPlane;
    float* red = reinterpret_cast<float*>(dsmem + 2 * kXPlane);                   // [4 waves][64 rows][4]
    unsigned char* bias_lds = reinterpret_cast<unsigned char*>(red + 4 * kXRows * 4);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 15, g = lane >> 4;
    const PointsX2Layout L = points_x2_layout();
    const xrsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wpack), 0, (int)L.total, 0x00020000);
    for (int i = tid; i < 6 * kXHidden / 4; i += 256)
        reinterpret_cast<f32x4*>(bias_lds)[i] = *reinterpret_cast<const f32x4*>(a.wpack + L.bias + (int64_t)i * 16);
    const float* s1 = reinterpret_cast<const float*>(a.wpack + L.s1);
    const float* c4 = reinterpret_cast<const float*>(a.wpack + L.c4);
    const int P = a.n_points ? min(*a.n_points, a.max_points) : a.max_points;
    const int ntiles = (P + kXRows - 1) / kXRows;
    const int w0 = (int)L.w[0];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * kXRows;
        XRing ring;
        xprefetch(rs, w0, wave, lane, ring);
        // ---- the tile's 64 x 256 fp32 features -> the two planes ----------------------------
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int cidx = it * 256 + tid, row = cidx >> 5, chunk = cidx & 31;
            const int p = row0 + row;
            f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = x0;
            if (p < P) {
                x0 = *reinterpret_cast<const f32x4*>(a.feat + (int64_t)p * kXHidden + chunk * 8);
                x1 = *reinterpret_cast<const f32x4*>(a.feat + (int64_t)p * kXHidden + chunk * 8 + 4);
            }
            uint32_t vh[4], vl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float y0 = q < 2 ? x0[2 * q] : x1[2 * q - 4], y1 = q < 2 ? x0[2 * q + 1] : x1[2 * q - 3];
                const __bf16 h0 = (__bf16)y0, h1 = (__bf16)y1;
                vh[q] = xpack2(h0, h1);
                vl[q] = xpack2((__bf16)(y0 - (float)h0), (__bf16)(y1 - (float)h1));
            }
            *reinterpret_cast<u32x4*>(Hh + row * kXPitch + chunk * 16) = u32x4{vh[0], vh[1], vh[2], vh[3]};
            *reinterpret_cast<u32x4*>(Hl + row * kXPitch + chunk * 16) = u32x4{vl[0], vl[1], vl[2], vl[3]};
        }
        __syncthreads();
        f32x4a acc[4][4];
        // ---- last aggregator layer (linear): feat -------------------------------------------
        xlayer(Hh, Hl, bias_lds, rs, w0, wave, lane, ring, acc);
        xprefetch(rs, w0 + 2 * kXMat, wave, lane, ring);
        const int64_t plane = (int64_t)a.max_points * kXHidden;        // (SAVE) one saved activation
        if (SAVE) xsave<false>(a.save, row0, P, wave, lane, acc);
        __syncthreads();
        xstore<false>(Hh, Hl, wave, lane, acc);
        __syncthreads();
        // ---- density head ---------------------------------------------------------------------
        xlayer(Hh, Hl, bias_lds + kXHidden * 4, rs, w0 + 2 * kXMat, wave, lane, ring, acc);
        xprefetch(rs, w0 + 4 * kXMat, wave, lane, ring);
        if (SAVE) xsave<true>(a.save + plane, row0, P, wave, lane, acc);
        {
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(s1 + 64 * wave + 32 * (mb >> 1) + 8 * g + 4 * (mb & 1));
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        float x = acc[mb][rb][b];
                        x = x > 0.f ? x : kXLeaky * x;
                        part[rb] = __builtin_fmaf(x, wv[b], part[rb]);
                    }
            }
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const float t = xsum_groups(part[rb]);
                if (g == 0) red[(wave * kXRows + rb * 16 + n) * 4 + 3] = t;
            }
        }
        // (the planes still hold feat: the density layer did not write activations)
        // ---- colour head ------------------------------------------------------------------------
#pragma unroll 1
        for (int l = 0; l < 4; ++l) {
            const int w_off = w0 + (2 + l) * (2 * kXMat);
            xlayer(Hh, Hl, bias_lds + (2 + l) * (kXHidden * 4), rs, w_off, wave, lane, ring, acc);
            if (l < 3) xprefetch(rs, w_off + 2 * kXMat, wave, lane, ring);
            if (DIR && l == 0) {
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    const int p = row0 + rb * 16 + n;
                    const float* db = a.dir_bias + (int64_t)(p < P ? a.point_ray[p] : 0) * kXHidden;
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(db + 64 * wave + 32 * (mb >> 1) + 8 * g + 4 * (mb & 1));
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[mb][rb][b] += v[b];
                    }
                }
            }
            if (SAVE) xsave<true>(a.save + (2 + l) * plane, row0, P, wave, lane, acc);
            if (l < 3) {
                __syncthreads();
                xstore<true>(Hh, Hl, wave, lane, acc);
                __syncthreads();
            }
        }
        {
            float pc[4][3];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) pc[rb][0] = pc[rb][1] = pc[rb][2] = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const float* cp = c4 + 64 * wave + 32 * (mb >> 1) + 8 * g + 4 * (mb & 1);
                const f32x4 wr = *reinterpret_cast<const f32x4*>(cp), wg = *reinterpret_cast<const f32x4*>(cp + kXHidden),
                            wb = *reinterpret_cast<const f32x4*>(cp + 2 * kXHidden);
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) {
                        float x = acc[mb][rb][b];
                        x = x > 0.f ? x : kXLeaky * x;
                        pc[rb][0] = __builtin_fmaf(x, wr[b], pc[rb][0]);
                        pc[rb][1] = __builtin_fmaf(x, wg[b], pc[rb][1]);
                        pc[rb][2] = __builtin_fmaf(x, wb[b], pc[rb][2]);
                    }
            }
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const float pr = xsum_groups(pc[rb][0]), pg = xsum_groups(pc[rb][1]), pb = xsum_groups(pc[rb][2]);
                if (g == 0) {
                    float* q = red + (wave * kXRows + rb * 16 + n) * 4;
                    q[0] = pr; q[1] = pg; q[2] = pb;
                }
            }
        }
        __syncthreads();
        if (tid < kXRows) {
            const int p = row0 + tid;
            if (p < P) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(red + (w * kXRows + tid) * 4);
                if (SAVE) {
                    *reinterpret_cast<f32x4*>(a.pre + (int64_t)p * 4) =
                        f32x4{t[0] + c4[3 * kXHidden], t[1] + c4[3 * kXHidden + 1], t[2] + c4[3 * kXHidden + 2], t[3] + s1[kXHidden]};
                } else {
                    a.sigma[p] = xsoftplus_m1(t[3] + s1[kXHidden]);
#pragma unroll
                    for (int c = 0; c < 3; ++c) a.rgb[(int64_t)p * 3 + c] = 1.f / (1.f + expf(-(t[c] + c4[3 * kXHidden + c])));
                }
            }
        }
        __syncthreads();
    }
}

// ============================================================================================
// The four non-linear per-pair layers + the inverse-distance mean in the same numerics, forward only (rendering)
// ============================================================================================
// Reference: gather + x_rel + weights + positional encoding (aggregators/mlp.py:62-88, aggregator.py:122-156, positional_encoder.py:16-20),
// local_field[0..7] (aggregators/mlp.py:83-84, utils/model.py:22-36), weighted aggregation (aggregators/mlp.py:102-125).  The training
// kernel of this numerics class (csrc/pairs_mlp.hip, precision 1) also renders, but it is built around saving activations: 128-row
// tiles on 32x32x16, one workgroup per CU -- 75 % of a fp32-class view.  This one is the fp16 pair kernel's 16x16x32 form
// (csrc/shade.hip) on two planes: tile = 8 points x 8 neighbour slots = 64 candidate rows packed to the front, row blocks of 16,
// layers through xlayer / xstore, the weighted mean in fp32 from hi + lo.  Takes the compact query's int32 lists as they are (-1 anywhere)
// and its device-side count: no host preprocessing.
constexpr int kXEnc = 64;            // 3 + 60 positional-encoding columns + 1 zero pad
struct PairsX2Layout {
    int64_t w[4];        // per matrix [hi][lo]; A0 has K0 = feat + 64 input columns (K0 / 32 steps), A1..A3 256
    int64_t bias;        // 4 x 256 fp32, [layer][wave][g][mb][4]
    int64_t total;
};
__host__ __device__ inline PairsX2Layout pairs_x2_layout(int feat) {
    PairsX2Layout L;
    int64_t off = 0;
    L.w[0] = off; off += (int64_t)2 * (feat + kXEnc) * kXHidden * 2;
    for (int i = 1; i < 4; ++i) { L.w[i] = off; off += 2 * kXMat; }
    L.bias = off; off += 4 * kXHidden * 4;
    L.total = off;
    return L;
}
__device__ __forceinline__ float xenc_value(int q, const float (&rel)[3]) {
    if (q < 3) return rel[q];
    if (q >= 63) return 0.f;
    const int c = (q - 3) / 20, rem = (q - 3) % 20, i = rem % 10;
    const float u = rel[c] * (0.5f * (float)(1 << i));           // sin(x 2^i pi) = sin(2 pi u): v_sin / v_cos take revolutions
    const float f = __builtin_amdgcn_fractf(u);
    return rem < 10 ? __builtin_amdgcn_sinf(f) : __builtin_amdgcn_cosf(f);
}
__device__ __forceinline__ void xput8(unsigned char* Hh, unsigned char* Hl, int off, const float (&x)[8]) {
    uint32_t vh[4], vl[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const __bf16 h0 = (__bf16)x[2 * q], h1 = (__bf16)x[2 * q + 1];
        vh[q] = xpack2(h0, h1);
        vl[q] = xpack2((__bf16)(x[2 * q] - (float)h0), (__bf16)(x[2 * q + 1] - (float)h1));
    }
    *reinterpret_cast<u32x4*>(Hh + off) = u32x4{vh[0], vh[1], vh[2], vh[3]};
    *reinterpret_cast<u32x4*>(Hl + off) = u32x4{vl[0], vl[1], vl[2], vl[3]};
}
template <int FEAT, int PART>
__device__ __forceinline__ void xenc_put16(unsigned char* Hh, unsigned char* Hl, int prow, const float (&rel)[3]) {
#pragma unroll
    for (int c8 = 0; c8 < 2; ++c8) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = xenc_value(PART * 16 + c8 * 8 + j, rel);
        xput8(Hh, Hl, prow * kXPitch + (FEAT / 8 + PART * 2 + c8) * 16, e);
    }
}
template <int FEAT, int NB16>
__device__ __forceinline__ void xpair_layers(unsigned char* Hh, unsigned char* Hl, const unsigned char* bias_lds, xrsrc_t rs, int w0, int w1,
                                             int wave, int lane) {
    constexpr int KS0 = (FEAT + kXEnc) / 32;
    f32x4a acc[4][4];
    XRing ring;
    xprefetch<KS0>(rs, w0, wave, lane, ring);
    xlayer<KS0, NB16>(Hh, Hl, bias_lds, rs, w0, wave, lane, ring, acc);
    xprefetch<8>(rs, w1, wave, lane, ring);
    __syncthreads();
    xstore<true, NB16>(Hh, Hl, wave, lane, acc);
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l < 4; ++l) {
        const int w_off = w1 + (l - 1) * (2 * kXMat);
        xlayer<8, NB16>(Hh, Hl, bias_lds + l * (kXHidden * 4), rs, w_off, wave, lane, ring, acc);
        if (l < 3) xprefetch<8>(rs, w_off + 2 * kXMat, wave, lane, ring);
        __syncthreads();
        xstore<true, NB16>(Hh, Hl, wave, lane, acc);
        __syncthreads();
    }
}
struct PairsX2Args {
    const unsigned char* wpack;
    const int32_t* nb_idx;        // [max_points][k] global neighbour indices, -1 = none (anywhere in a row)
    const float *pts, *kp_pos, *kp_feat;
    const int32_t* n_points;      // device-side count (may be null: max_points)
    int max_points, k;
    float* G;                     // [max_points][256] fp32: the aggregated per-point features
};
template <int FEAT>
__global__ __launch_bounds__(256, 2) void pairs_x2_kernel(PairsX2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    unsigned char* Hh = dsmem;
    unsigned char* Hl = dsmem + kXPlane;
    float* wrow = reinterpret_cast<float*>(dsmem + 2 * kXPlane);                 // [64] inverse distances of the packed rows
    int* pstart = reinterpret_cast<int*>(wrow + kXRows);                          // [8] first packed row of each point
    int* pcount = pstart + 8;                                                     // [8] its number of valid neighbours
    unsigned char* bias_lds = reinterpret_cast<unsigned char*>(pcount + 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const PairsX2Layout L = pairs_x2_layout(FEAT);
    const xrsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(a.wpack), 0, (int)L.total, 0x00020000);
    for (int i = tid; i < 4 * kXHidden / 4; i += 256)
        reinterpret_cast<f32x4*>(bias_lds)[i] = *reinterpret_cast<const f32x4*>(a.wpack + L.bias + (int64_t)i * 16);
    const int P = a.n_points ? min(*a.n_points, a.max_points) : a.max_points;
    const int ntiles = (P + 7) / 8;
    const int w0 = (int)L.w[0], w1 = (int)L.w[1];
    constexpr int FQ = FEAT / 4;          // input features per wave (a wave = the tile's 64 candidates x one quarter of the input columns)
    // what a lane needs to build its row of a tile, requested one tile AHEAD (the neighbour index while the previous tile's layers
    // run, the gathered point / position / features while its aggregation runs): the prologue starts from registers
    const int slot = lane & 7, part = wave;
    auto load_index = [&](int t) {
        const int p = t * 8 + (lane >> 3);
        return (p < P && slot < a.k) ? a.nb_idx[(int64_t)p * a.k + slot] : -1;
    };
    float in_pt[3], in_kp[3], in_fx[FQ];
    auto load_data = [&](int t, int gi) {       // (unconditional, clamped indices: no value of the previous tile has to survive the layers)
        const int pc = min(t * 8 + (lane >> 3), a.max_points - 1), gc = max(gi, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) { in_pt[c] = a.pts[(int64_t)pc * 3 + c]; in_kp[c] = a.kp_pos[(int64_t)gc * 3 + c]; }
#pragma unroll
        for (int c4 = 0; c4 < FQ / 4; ++c4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(a.kp_feat + (int64_t)gc * FEAT + part * FQ + c4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) in_fx[c4 * 4 + j] = v[j];
        }
    };
    int tile = blockIdx.x, gi = -1;
    if (tile < ntiles) {
        gi = load_index(tile);
        load_data(tile, gi);
    }
    while (tile < ntiles) {
        int nblk;
        {
            const int row = lane;
            float rel[3] = {0.f, 0.f, 0.f};
            if (gi >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rel[c] = in_pt[c] - in_kp[c];
            }
            const unsigned long long mine = __ballot(gi >= 0);
            const int V = __popcll(mine), prow = __popcll(mine & ((1ull << lane) - 1ull));
            nblk = (V + 15) >> 4;
            if (part == 0 && slot == 0) {
                pstart[row >> 3] = prow;
                pcount[row >> 3] = __popcll((mine >> (lane & ~7)) & 0xffull);
            }
            if (part == 0 && gi >= 0) wrow[prow] = 1.f / (sqrtf(__builtin_fmaf(rel[0], rel[0], __builtin_fmaf(rel[1], rel[1], rel[2] * rel[2]))) + 1e-5f);
            if (gi >= 0) {
#pragma unroll
                for (int c8 = 0; c8 < FQ / 8; ++c8) {
                    float e[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) e[j] = in_fx[c8 * 8 + j];
                    xput8(Hh, Hl, prow * kXPitch + (part * (FQ / 8) + c8) * 16, e);
                }
                switch (part) {        // (wave-uniform)
                    case 0: xenc_put16<FEAT, 0>(Hh, Hl, prow, rel); break;
                    case 1: xenc_put16<FEAT, 1>(Hh, Hl, prow, rel); break;
                    case 2: xenc_put16<FEAT, 2>(Hh, Hl, prow, rel); break;
                    default: xenc_put16<FEAT, 3>(Hh, Hl, prow, rel); break;
                }
            }
            if (row >= V && row < 16 * nblk) {          // computed (whole row blocks) but never aggregated: defined inputs
                const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int c8 = 0; c8 < FQ / 8; ++c8) {
                    *reinterpret_cast<u32x4*>(Hh + row * kXPitch + (part * (FQ / 8) + c8) * 16) = z;
                    *reinterpret_cast<u32x4*>(Hl + row * kXPitch + (part * (FQ / 8) + c8) * 16) = z;
                }
#pragma unroll
                for (int c8 = 0; c8 < 2; ++c8) {
                    *reinterpret_cast<u32x4*>(Hh + row * kXPitch + (FEAT / 8 + part * 2 + c8) * 16) = z;
                    *reinterpret_cast<u32x4*>(Hl + row * kXPitch + (FEAT / 8 + part * 2 + c8) * 16) = z;
                }
            }
        }
        const int ntile = tile + (int)gridDim.x;
        gi = ntile < ntiles ? load_index(ntile) : -1;          // arrives while the layers run
        __syncthreads();
        switch (nblk) {                    // (workgroup-uniform)
            case 4: xpair_layers<FEAT, 4>(Hh, Hl, bias_lds, rs, w0, w1, wave, lane); break;
            case 3: xpair_layers<FEAT, 3>(Hh, Hl, bias_lds, rs, w0, w1, wave, lane); break;
            case 2: xpair_layers<FEAT, 2>(Hh, Hl, bias_lds, rs, w0, w1, wave, lane); break;
            case 1: xpair_layers<FEAT, 1>(Hh, Hl, bias_lds, rs, w0, w1, wave, lane); break;
            default: break;                // no valid pair in the tile
        }
        load_data(min(ntile, ntiles - 1), gi);                  // ... and the gathered inputs while the aggregation runs
        __builtin_amdgcn_sched_barrier(0);
        {
            // point pl, channels 8 cc .. + 7: all eight slots at once (slot >= count re-reads the point's last row with weight 0)
            const int pl = tid >> 5, cc = tid & 31;
            const int p = tile * 8 + pl;
            const int r0 = pstart[pl], cnt = pcount[pl], last = max(cnt - 1, 0);
            float w8[8], wsum = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                w8[s2] = wrow[r0 + min(s2, last)];
                w8[s2] = s2 < cnt ? w8[s2] : 0.f;
                wsum += w8[s2];
            }
            const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
            float out[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                const int o = (r0 + min(s2, last)) * kXPitch + cc * 16;
                const bf16x8 vh = *reinterpret_cast<const bf16x8*>(Hh + o), vl = *reinterpret_cast<const bf16x8*>(Hl + o);
                const float ws = w8[s2] * inv;
#pragma unroll
                for (int j = 0; j < 8; ++j) out[j] = __builtin_fmaf(ws, (float)vh[j] + (float)vl[j], out[j]);
            }
            if (p < P) {
                float* gp = a.G + (int64_t)p * kXHidden + cc * 8;
                const bool any = cnt > 0;
                *reinterpret_cast<f32x4*>(gp) = any ? f32x4{out[0], out[1], out[2], out[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(gp + 4) = any ? f32x4{out[4], out[5], out[6], out[7]} : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();
        tile = ntile;
    }
}

// device-side packing of the point-level pack (training: the weights change every step; the host packer costs ~5 ms per call)
struct PointsX2PackArgs {
    const float* w[8];      // local_field.8, shape_net.0, shape_net.2, channel_net.0, .2, .4, .6, .8 (row-major [out][in])
    const float* b[8];
    unsigned char* out;
    int c0_in_dim;
};
__global__ __launch_bounds__(256) void points_x2_pack_kernel(PointsX2PackArgs a) {
    const PointsX2Layout L = points_x2_layout();
    const int wide[6] = {0, 1, 3, 4, 5, 6};              // the six 256 x 256 matrices among the eight
    const int i = blockIdx.y;                             // matrix
    const int e = blockIdx.x * 256 + threadIdx.x;         // element of its fragment stream: [wave][s][mb][lane][8]
    if (i < 6) {
        const int j = e & 7, lane = (e >> 3) & 63, mb = (e >> 9) & 3, s2 = (e >> 11) & 7, w = e >> 14;
        const int m = lane & 15, o = 64 * w + 32 * (mb >> 1) + 8 * (m >> 2) + 4 * (mb & 1) + (m & 3), c = 32 * s2 + 8 * (lane >> 4) + j;
        const int in_dim = wide[i] == 3 ? a.c0_in_dim : kXHidden;
        const float v = a.w[wide[i]][(int64_t)o * in_dim + c];
        const __bf16 h = (__bf16)v;
        reinterpret_cast<__bf16*>(a.out + L.w[i])[e] = h;
        reinterpret_cast<__bf16*>(a.out + L.w[i] + kXMat)[e] = (__bf16)(v - (float)h);
        if (e < kXHidden) {                                // bias: [w][g][mb][b]
            const int b = e & 3, mb2 = (e >> 2) & 3, g = (e >> 4) & 3, w2 = e >> 6;
            reinterpret_cast<float*>(a.out + L.bias)[i * kXHidden + e] = a.b[wide[i]][64 * w2 + 32 * (mb2 >> 1) + 8 * g + 4 * (mb2 & 1) + b];
        }
    } else if (e < kXHidden) {                             // the heads' last layers, fp32
        float* s1 = reinterpret_cast<float*>(a.out + L.s1);
        float* c4 = reinterpret_cast<float*>(a.out + L.c4);
        s1[e] = a.w[2][e];
        c4[e] = a.w[7][e]; c4[kXHidden + e] = a.w[7][kXHidden + e]; c4[2 * kXHidden + e] = a.w[7][2 * kXHidden + e];
        if (e == 0) s1[kXHidden] = a.b[2][0];
        if (e < 3) c4[3 * kXHidden + e] = a.b[7][e];
    }
}

}  // namespace npcd

using namespace npcd;

extern "C" int64_t npcd_points_x2_wpack_bytes(void) { return points_x2_layout().total; }

// weights_host / biases_host: the twelve pointers of npcd_shade_pack_weights (aggregator.local_field.{0,2,4,6,8}, shape_net.{0,2},
// channel_net.{0,2,4,6,8}); entries 4..11 are read.  c0_in_dim = input columns of channel_net.0 (256, or 256 + the direction
// encoding with use_view_dir: its first 256 columns are packed, the rest enters as dir_bias).
extern "C" int npcd_points_x2_pack(const float* const* weights_host, const float* const* biases_host, int c0_in_dim, void* wpack_host) {
    if (!weights_host || !biases_host || !wpack_host || c0_in_dim < kXHidden) return NPCD_ERR_ARG;
    for (int i = 4; i < 12; ++i)
        if (!weights_host[i] || !biases_host[i]) return NPCD_ERR_ARG;
    const PointsX2Layout L = points_x2_layout();
    unsigned char* out = static_cast<unsigned char*>(wpack_host);
    memset(out, 0, L.total);
    const int src[6] = {4, 5, 7, 8, 9, 10};
    for (int i = 0; i < 6; ++i) {
        const float* W = weights_host[src[i]];
        const int in_dim = src[i] == 7 ? c0_in_dim : kXHidden;
        __bf16* dh = reinterpret_cast<__bf16*>(out + L.w[i]);
        __bf16* dl = reinterpret_cast<__bf16*>(out + L.w[i] + kXMat);
        for (int w = 0; w < 4; ++w)
            for (int s = 0; s < 8; ++s)
                for (int mb = 0; mb < 4; ++mb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int m = lane & 15, o = 64 * w + 32 * (mb >> 1) + 8 * (m >> 2) + 4 * (mb & 1) + (m & 3), c = 32 * s + 8 * (lane >> 4) + j;
                            const float v = W[(int64_t)o * in_dim + c];
                            const __bf16 h = (__bf16)v;
                            const int64_t at = ((((int64_t)w * 8 + s) * 4 + mb) * 64 + lane) * 8 + j;
                            dh[at] = h;
                            dl[at] = (__bf16)(v - (float)h);
                        }
        float* db = reinterpret_cast<float*>(out + L.bias) + i * kXHidden;
        for (int w = 0; w < 4; ++w)
            for (int g = 0; g < 4; ++g)
                for (int mb = 0; mb < 4; ++mb)
                    for (int b = 0; b < 4; ++b) db[((w * 4 + g) * 4 + mb) * 4 + b] = biases_host[src[i]][64 * w + 32 * (mb >> 1) + 8 * g + 4 * (mb & 1) + b];
    }
    float* s1 = reinterpret_cast<float*>(out + L.s1);
    memcpy(s1, weights_host[6], kXHidden * 4);
    s1[kXHidden] = biases_host[6][0];
    float* c4 = reinterpret_cast<float*>(out + L.c4);
    memcpy(c4, weights_host[11], 3 * kXHidden * 4);
    memcpy(c4 + 3 * kXHidden, biases_host[11], 3 * 4);
    return NPCD_OK;
}

extern "C" int npcd_points_x2(const void* wpack, const float* feat, const int32_t* n_points_dev, int max_points, float* sigma, float* rgb,
                              const float* dir_bias, const int32_t* point_ray, void* stream) {
    if (!wpack || !feat || !sigma || !rgb || max_points < 0) return NPCD_ERR_ARG;
    if ((dir_bias != nullptr) != (point_ray != nullptr)) return NPCD_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(dir_bias) | reinterpret_cast<uintptr_t>(wpack)) & 15) return NPCD_ERR_ARG;
    if (max_points == 0) return NPCD_OK;
    PointsX2Args a{static_cast<const unsigned char*>(wpack), feat, n_points_dev, max_points, sigma, rgb, dir_bias, point_ray, nullptr, nullptr};
    const int lds = 2 * kXPlane + 4 * kXRows * 4 * 4 + 6 * kXHidden * 4;
    static DynLds lds_p, lds_d;
    NPCD_HIP_CHECK(lds_p.ensure(reinterpret_cast<const void*>(points_x2_kernel<false>), lds));
    NPCD_HIP_CHECK(lds_d.ensure(reinterpret_cast<const void*>(points_x2_kernel<true>), lds));
    const int tiles = (max_points + kXRows - 1) / kXRows, grid = tiles < 512 ? tiles : 512;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dir_bias) hipLaunchKernelGGL(points_x2_kernel<true>, dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(points_x2_kernel<false>, dim3(grid), dim3(256), lds, st, a);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int64_t npcd_pairs_x2_wpack_bytes(int feat_dim) {
    if (feat_dim != 32 && feat_dim != 128) return -1;
    return pairs_x2_layout(feat_dim).total;
}

// weights_host / biases_host: the twelve pointers of npcd_shade_pack_weights; entries 0..3 (aggregator.local_field.{0,2,4,6}) are read.
extern "C" int npcd_pairs_x2_pack(const float* const* weights_host, const float* const* biases_host, int feat_dim, void* wpack_host) {
    if (!weights_host || !biases_host || !wpack_host) return NPCD_ERR_ARG;
    if (feat_dim != 32 && feat_dim != 128) return NPCD_ERR_UNSUPPORTED;
    for (int i = 0; i < 4; ++i)
        if (!weights_host[i] || !biases_host[i]) return NPCD_ERR_ARG;
    const PairsX2Layout L = pairs_x2_layout(feat_dim);
    unsigned char* out = static_cast<unsigned char*>(wpack_host);
    memset(out, 0, L.total);
    const int in0 = feat_dim + 63, k0 = feat_dim + kXEnc;
    for (int i = 0; i < 4; ++i) {
        const float* W = weights_host[i];
        const int in_dim = i == 0 ? in0 : kXHidden, ks = (i == 0 ? k0 : kXHidden) / 32;
        __bf16* dh = reinterpret_cast<__bf16*>(out + L.w[i]);
        __bf16* dl = reinterpret_cast<__bf16*>(out + L.w[i] + (int64_t)ks * 16384);
        for (int w = 0; w < 4; ++w)
            for (int s = 0; s < ks; ++s)
                for (int mb = 0; mb < 4; ++mb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int m = lane & 15, o = 64 * w + 32 * (mb >> 1) + 8 * (m >> 2) + 4 * (mb & 1) + (m & 3), c = 32 * s + 8 * (lane >> 4) + j;
                            const float v = c < in_dim ? W[(int64_t)o * in_dim + c] : 0.f;
                            const __bf16 h = (__bf16)v;
                            const int64_t at = ((((int64_t)w * ks + s) * 4 + mb) * 64 + lane) * 8 + j;
                            dh[at] = h;
                            dl[at] = (__bf16)(v - (float)h);
                        }
        float* db = reinterpret_cast<float*>(out + L.bias) + i * kXHidden;
        for (int w = 0; w < 4; ++w)
            for (int g = 0; g < 4; ++g)
                for (int mb = 0; mb < 4; ++mb)
                    for (int b = 0; b < 4; ++b) db[((w * 4 + g) * 4 + mb) * 4 + b] = biases_host[i][64 * w + 32 * (mb >> 1) + 8 * g + 4 * (mb & 1) + b];
    }
    return NPCD_OK;
}

extern "C" int npcd_pairs_x2(const void* wpack, int feat_dim, const int32_t* nb_idx, const float* pts, const float* kp_pos, const float* kp_feat,
                             const int32_t* n_points_dev, int max_points, int k, float* G, void* stream) {
    if (!wpack || !nb_idx || !pts || !kp_pos || !kp_feat || !G || max_points < 0) return NPCD_ERR_ARG;
    if (feat_dim != 32 && feat_dim != 128) return NPCD_ERR_UNSUPPORTED;
    if (k <= 0 || k > 8) return NPCD_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(kp_feat) | reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(wpack)) & 15) return NPCD_ERR_ARG;
    if (max_points == 0) return NPCD_OK;
    PairsX2Args a{static_cast<const unsigned char*>(wpack), nb_idx, pts, kp_pos, kp_feat, n_points_dev, max_points, k, G};
    const int lds = 2 * kXPlane + kXRows * 4 + 16 * 4 + 4 * kXHidden * 4;
    static DynLds lds32, lds128;
    NPCD_HIP_CHECK(lds32.ensure(reinterpret_cast<const void*>(pairs_x2_kernel<32>), lds));
    NPCD_HIP_CHECK(lds128.ensure(reinterpret_cast<const void*>(pairs_x2_kernel<128>), lds));
    const int tiles = (max_points + 7) / 8, grid = tiles < 2048 ? tiles : 2048;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (feat_dim == 32) hipLaunchKernelGGL(pairs_x2_kernel<32>, dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(pairs_x2_kernel<128>, dim3(grid), dim3(256), lds, st, a);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

// Training forward of the same layers: `save` [6][max_points][256] fp32 = feat, s0, c0, c1, c2, c3 (the activations the backward needs, after
// LeakyReLU where the layer has one), `pre` [max_points][4] = the heads' pre-activations (r, g, b, sigma) with their biases.
extern "C" int npcd_points_x2_train(const void* wpack, const float* feat, int max_points, float* save, float* pre, void* stream) {
    if (!wpack || !feat || !save || !pre || max_points < 0) return NPCD_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(save) | reinterpret_cast<uintptr_t>(pre) | reinterpret_cast<uintptr_t>(wpack)) & 15)
        return NPCD_ERR_ARG;
    if (max_points == 0) return NPCD_OK;
    PointsX2Args a{static_cast<const unsigned char*>(wpack), feat, nullptr, max_points, nullptr, nullptr, nullptr, nullptr, save, pre};
    const int lds = 2 * kXPlane + 4 * kXRows * 4 * 4 + 6 * kXHidden * 4;
    static DynLds lds_s;
    NPCD_HIP_CHECK(lds_s.ensure(reinterpret_cast<const void*>(points_x2_kernel<false, true>), lds));
    const int tiles = (max_points + kXRows - 1) / kXRows, grid = tiles < 512 ? tiles : 512;
    hipLaunchKernelGGL((points_x2_kernel<false, true>), dim3(grid), dim3(256), lds, static_cast<hipStream_t>(stream), a);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

// The same pack from DEVICE tensors (training: once per optimizer step): weights_dev / biases_dev = the eight device pointers of
// local_field.8, shape_net.{0,2}, channel_net.{0,2,4,6,8} in that order (a host array of pointers).
extern "C" int npcd_points_x2_pack_dev(const float* const* weights_dev, const float* const* biases_dev, int c0_in_dim, void* wpack_dev, void* stream) {
    if (!weights_dev || !biases_dev || !wpack_dev || c0_in_dim < kXHidden) return NPCD_ERR_ARG;
    PointsX2PackArgs a;
    for (int i = 0; i < 8; ++i) {
        if (!weights_dev[i] || !biases_dev[i]) return NPCD_ERR_ARG;
        a.w[i] = weights_dev[i];
        a.b[i] = biases_dev[i];
    }
    a.out = static_cast<unsigned char*>(wpack_dev);
    a.c0_in_dim = c0_in_dim;
    hipLaunchKernelGGL(points_x2_pack_kernel, dim3(kXHidden * kXHidden / 256, 7), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
