// Stage-1 training path, the per-pair MLP itself on the matrix cores: forward AND backward of the four
// non-linear layers of the aggregator network  Linear(F+63 -> 256), 3 x Linear(256 -> 256), each followed by LeakyReLU(0.01)
// (aggregators/mlp.py:36-100, utils/model.py:22-36, pointnerf.py:174-179), fused with what surrounds them:
//
//   forward  (one launch)   gather feat[nb] | rel = pt - pos[nb] | positional encoding  ->  4 layers, activations resident in LDS
//                           ->  inverse-distance weighted mean over each point's pairs (aggregators/mlp.py:102-125).
//                           The network's fifth layer is linear and commutes with that mean: it is applied by the caller on
//                           POINTS (6x fewer rows), like the evaluation path (csrc/shade.hip) does.
//                           Written for the backward: the layer inputs X0 [Q, F+64] and A_0..A_3 [Q, 256] (bf16), the
//                           normalised weights wn [Q].
//   backward (one launch per layer, last to first)   dZ_l = dA_l * leaky'(A_l);  db_l = colsum dZ_l;  dW_l = dZ_l^T A_{l-1};
//                           dA_{l-1} = dZ_l W_l.  A workgroup keeps its 256 x K slab of dW_l in registers over all its row tiles
//                           (a 10^6-long reduction), writes it once, and a second launch sums the slabs in a fixed order.
//                           dA_3 = wn * dG[owner] is formed on the fly from the gradient of the aggregated features.
//
// Numerics, two modes (`precision` of the entry points):
//   0  bf16 operands, fp32 accumulation, fp32 weight / bias gradients (what `PointNeRFTrainer(mlp_dtype=torch.bfloat16)` computed
//      with library GEMMs before): NARROWER than the reference, which trains this stage in fp32.
//   1  "x2", the fp32-class mode (template parameter X2): every matrix operand -- weights, activations, gradients -- is carried as TWO
//      bf16 halves  x = hi + lo,  hi = bf16(x), lo = bf16(x - hi)  (16 mantissa bits, fp32's exponent range: nothing can overflow
//      that fp32 holds), every product as THREE matrix instructions  hi*hi + hi*lo + lo*hi  accumulated in fp32 (the lo*lo term,
//      2^-18 relative, is dropped): ~1e-5 relative per product against fp32's 6e-8 and bf16's 4e-3, at a third of the bf16 matrix
//      rate = 5 x the fp32 matrix instruction's.  Activations saved for the backward as the two planes (4 B per element, like fp32).
//      The row tile of the backward is 64 instead of 128 rows (four LDS planes), the forward runs eight waves on one 128-row tile.
// Pairs are compact and ordered by point (row q of every [Q, .] array; off[p] = first pair of point p), see csrc/pairs.hip.
//
// Bounds.  Forward: matrix pipe (0.41 MFLOP per pair, 2.2 kB written per pair).  Backward layer: HBM -- per 128-row tile
// 33.5 MFLOP against 256 kB moved (dA in, A_l, A_{l-1}, dA out), i.e. 131 FLOP/B, under the 312 FLOP/B ridge of the chip.
#include <stdlib.h>

#include "common.h"

namespace npcd {

constexpr int kPH = 256;                 // hidden width
constexpr int kPEnc = 64;                // 3 + 60 positional-encoding columns + 1 zero pad
constexpr int kPTile = 128;              // rows per tile
constexpr int kPitchR = kPH * 2 + 16;    // 528 B: ds_read_b128 row reads of 16 consecutive rows cover all 64 banks
constexpr int kPFrag = 1024;             // one 32 x 16 bf16 fragment (64 lanes x 16 B)
constexpr float kPSlope = 0.01f;
constexpr int kPFreqs = 10;

// ---- packed weights: bf16 MFMA A-operand fragments for the forward (W_l) and the data gradient (W_l^T), fp32 biases ----
// byte offsets inside the packed buffer (plain arithmetic: a struct of arrays indexed by a runtime layer number lands in scratch)
//   W_l   (forward)       : [8 out blocks][K_l / 16 steps][64 lanes][8],  K_0 = F + 64, K_l = 256
//   W_l^T (data gradient) : [in blocks][16 steps over the 256 outputs][64][8]   (layer 0: the F / 32 feature blocks only)
//   bias                  : fp32 [256]
__host__ __device__ inline int64_t pl_fw(int feat, int l) {
    const int64_t l0 = (int64_t)8 * ((feat + kPEnc) / 16) * kPFrag, ll = (int64_t)8 * (kPH / 16) * kPFrag;
    return l == 0 ? 0 : l0 + (l - 1) * ll;
}
__host__ __device__ inline int64_t pl_bw(int feat, int l) {
    const int64_t base = pl_fw(feat, 4), l0 = (int64_t)(feat / 32) * 16 * kPFrag, ll = (int64_t)(kPH / 32) * 16 * kPFrag;
    return l == 0 ? base : base + l0 + (l - 1) * ll;
}
// x2: a second set of matrices (the lo halves) in the same order behind the first, the biases behind both
__host__ __device__ inline int64_t pl_mats(int feat) { return pl_bw(feat, 4); }
__host__ __device__ inline int64_t pl_bias(int feat, int l, bool x2 = false) { return (x2 ? 2 : 1) * pl_mats(feat) + (int64_t)l * kPH * 4; }
__host__ __device__ inline int64_t pl_total(int feat, bool x2 = false) { return pl_bias(feat, 4, x2); }

struct PackArgs {
    const float* W[4];
    const float* b[4];
    unsigned char* out;
    int feat_dim;
    int x2;
};
// the two bf16 halves of a float: x ~ hi + lo
__device__ __forceinline__ void pm_split(float x, __bf16& hi, __bf16& lo) {
    hi = (__bf16)x;
    lo = (__bf16)(x - (float)hi);
}
// one thread per packed 16-byte fragment piece (8 elements)
__global__ __launch_bounds__(256) void pair_pack_kernel(PackArgs a) {
    const int F = a.feat_dim;
    const int in0 = a.feat_dim + 3 + 6 * kPFreqs;
    const int64_t piece = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t npieces = pl_mats(F) / 16;
    const int plane = piece >= npieces ? 1 : 0;                  // x2: the lo halves are a second pass over the same pieces
    if (piece < npieces * (a.x2 ? 2 : 1)) {
        const int64_t byte = (piece - plane * npieces) * 16;
        const bool bwd = byte >= pl_bw(F, 0);
        int l = 0;
        for (int i = 1; i < 4; ++i)
            if (byte >= (bwd ? pl_bw(F, i) : pl_fw(F, i))) l = i;
        const int in_dim = l == 0 ? in0 : kPH;
        const int64_t rel = (byte - (bwd ? pl_bw(F, l) : pl_fw(F, l))) / 16;      // piece index inside the matrix: ((blk * steps) + s) * 64 + lane
        const int lane = (int)(rel & 63);
        const int steps = bwd ? 16 : (l == 0 ? F + kPEnc : kPH) / 16;
        const int s = (int)((rel >> 6) % steps), blk = (int)((rel >> 6) / steps);
        bf16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int m = blk * 32 + (lane & 31), k = 16 * s + 8 * (lane >> 5) + j;
            float x;
            if (!bwd) x = k < in_dim ? a.W[l][(int64_t)m * in_dim + k] : 0.f;      // A = W[out m][in k]
            else x = a.W[l][(int64_t)k * in_dim + m];                              // A = W^T[in m][out k]   (m < F <= in_dim for layer 0)
            __bf16 hi, lo;
            pm_split(x, hi, lo);
            v[j] = plane ? lo : hi;
        }
        *reinterpret_cast<bf16x8*>(a.out + plane * pl_mats(F) + byte) = v;
    }
    if (piece < 4 * kPH) {
        const int l = (int)(piece / kPH), c = (int)(piece % kPH);
        reinterpret_cast<float*>(a.out + pl_bias(F, l, a.x2 != 0))[c] = a.b[l][c];
    }
}

__device__ __forceinline__ int row_off(int row, int chunk) { return row * kPitchR + (chunk << 4); }

// acc[oi][cb] = bias + W[(NOB*wave+oi)*32.., :] . H^T[:, cb*32..]   (activations = B operand, weights = A operand streamed from L2)
// X2: the lo plane of the activations lies `hplane` bytes behind the hi plane in LDS, the lo weights `wlo` bytes behind the hi ones;
// three matrix instructions per (fragment, fragment) pair: hi*hi + hi*lo + lo*hi.
template <int KSTEPS, int NOB, int NCB, bool X2 = false>
__device__ __forceinline__ void pm_layer_mfma(const unsigned char* H, int pitch, const unsigned char* wfrag, const float* bias, int wave, int lane,
                                              f32x16 (&acc)[NOB][NCB], int nblk, int hplane = 0, int64_t wlo = 0) {
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int oi = 0; oi < NOB; ++oi) {
        f32x16 init;
        if (bias) {
            const float* bp = bias + (NOB * wave + oi) * 32 + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + 8 * g);
#pragma unroll
                for (int b = 0; b < 4; ++b) init[4 * g + b] = b4[b];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) init[i] = 0.f;
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[oi][cb] = init;
    }
    const bf16x8* w[NOB];
#pragma unroll
    for (int oi = 0; oi < NOB; ++oi) w[oi] = reinterpret_cast<const bf16x8*>(wfrag + (int64_t)(NOB * wave + oi) * KSTEPS * kPFrag) + lane;
    const unsigned char* hb = H + r * pitch + hh * 16;
    bf16x8 a[NOB], nx[NOB], al[NOB], nxl[NOB];
#pragma unroll
    for (int oi = 0; oi < NOB; ++oi) {
        a[oi] = w[oi][0];
        if (X2) al[oi] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(w[oi]) + wlo);
    }
#pragma unroll 4
    for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
        for (int oi = 0; oi < NOB; ++oi) {
            nx[oi] = s + 1 < KSTEPS ? w[oi][(s + 1) * 64] : a[oi];
            if (X2) nxl[oi] = s + 1 < KSTEPS ? *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(w[oi] + (s + 1) * 64) + wlo) : al[oi];
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            if (cb < nblk) {                         // wave-uniform: whole 32-row blocks past the tile's rows are skipped
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(hb + cb * 32 * pitch + s * 32);
                if (X2) {
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(hb + hplane + cb * 32 * pitch + s * 32);
#pragma unroll
                    for (int oi = 0; oi < NOB; ++oi) {          // small terms first, the leading product last
                        acc[oi][cb] = BF16::mfma32(al[oi], b, acc[oi][cb]);
                        acc[oi][cb] = BF16::mfma32(a[oi], bl, acc[oi][cb]);
                    }
                }
#pragma unroll
                for (int oi = 0; oi < NOB; ++oi) acc[oi][cb] = BF16::mfma32(a[oi], b, acc[oi][cb]);
            }
        }
#pragma unroll
        for (int oi = 0; oi < NOB; ++oi) {
            a[oi] = nx[oi];
            if (X2) al[oi] = nxl[oi];
        }
    }
}

__device__ __forceinline__ uint32_t pm_pack2(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
}

// epilogue: optional LeakyReLU, convert to bf16, write in place: lane owns row (cb*32 + r), channels (NOB wave + oi)*32 + 8 g + 4 hh ..
// X2: the remainder x - hi goes to the lo plane, `hplane` bytes further
template <bool ACT, int NOB, int NCB, bool X2 = false>
__device__ __forceinline__ void pm_layer_store(unsigned char* H, int pitch, int wave, int lane, const f32x16 (&acc)[NOB][NCB], int nblk, int hplane = 0) {
    const int r = lane & 31, hh = lane >> 5;
    unsigned char* sb = H + r * pitch + hh * 8 + wave * (NOB * 64);
#pragma unroll
    for (int oi = 0; oi < NOB; ++oi)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (cb >= nblk) continue;
                u32x2 v, vl;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float x0 = acc[oi][cb][4 * g + 2 * b], x1 = acc[oi][cb][4 * g + 2 * b + 1];
                    if (ACT) {
                        x0 = fmaxf(x0, kPSlope * x0);
                        x1 = fmaxf(x1, kPSlope * x1);
                    }
                    v[b] = pm_pack2(x0, x1);
                    if (X2) vl[b] = pm_pack2(x0 - BF16::lo(v[b]), x1 - BF16::hi(v[b]));
                }
                *reinterpret_cast<u32x2*>(sb + cb * 32 * pitch + (oi * 4 + g) * 16) = v;
                if (X2) *reinterpret_cast<u32x2*>(sb + hplane + cb * 32 * pitch + (oi * 4 + g) * 16) = vl;
            }
}

// eight fp32 values -> one 16-byte chunk of the hi plane (and of the lo plane)
template <bool X2>
__device__ __forceinline__ void pm_put8(unsigned char* p, int hplane, const float (&x)[8]) {
    bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 hi, lo;
        pm_split(x[j], hi, lo);
        h[j] = hi;
        l[j] = lo;
    }
    *reinterpret_cast<bf16x8*>(p) = h;
    if (X2) *reinterpret_cast<bf16x8*>(p + hplane) = l;
}

__device__ __forceinline__ float pm_enc_value(int q, const float rel[3]) {
    if (q < 3) return rel[q];
    if (q >= 63) return 0.f;
    const int c = (q - 3) / 20, rem = (q - 3) % 20, i = rem % 10;
    const float u = rel[c] * (0.5f * (float)(1 << i));           // sin(x 2^i pi) = sin(2 pi u): v_sin / v_cos take revolutions
    const float f = __builtin_amdgcn_fractf(u);
    return rem < 10 ? __builtin_amdgcn_sinf(f) : __builtin_amdgcn_cosf(f);
}

struct PairFwdArgs {
    const unsigned char* wpack;
    const int64_t* nb_idx;     // [P, k] global neighbour indices, -1 pad (valid entries first)
    const float *pts, *kp_pos, *kp_feat;
    const int64_t* off;        // [P] first pair row of each point
    int64_t P, Q;
    int k;
    __bf16* x0;                // [Q, F + 64]            (x2: [2][Q][F + 64], hi plane then lo plane); nullptr = not saved (inference)
    __bf16* acts;              // [4][Q][256]            (x2: [4][2][Q][256]); nullptr = not saved
    float* wn;                 // [Q]; nullptr = not saved
    float* G;                  // [P, 256]
};

// copy the first `rows` rows of the LDS tile (width bytes each) to a row-major global array starting at row `row0`
template <int NT>
__device__ __forceinline__ void pm_copy_out(const unsigned char* H, int rows, int width_bytes, unsigned char* dst, int64_t row0, int tid) {
    const int cpr = width_bytes >> 4;
    for (int c = tid; c < rows * cpr; c += NT) {
        const int row = c / cpr, chunk = c - row * cpr;
        *reinterpret_cast<u32x4*>(dst + (row0 + row) * width_bytes + (chunk << 4)) = *reinterpret_cast<const u32x4*>(H + row_off(row, chunk));
    }
}
// eight positional-encoding columns BASE .. BASE + 7 of a row
// (`base` is a compile-time constant at every call site once the caller's loop is unrolled: the column tests of pm_enc_value fold)
template <bool X2>
__device__ __forceinline__ void pm_enc8(unsigned char* p, int hplane, const float rel[3], int base) {
    float e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = pm_enc_value(base + j, rel);
    pm_put8<X2>(p, hplane, e);
}

// tile = 16 points x 8 neighbour slots; the tile's valid pairs are packed to the front (a point's rows stay consecutive).
// X2 = false: 256 threads (wave = two 32-channel output blocks), two workgroups per CU.  X2 = true: the two LDS planes of a tile are
// 135 KB, one workgroup per CU -- of 512 threads (wave = one output block), so that a SIMD still holds two waves.
template <int FEAT, bool X2>
__global__ __launch_bounds__(X2 ? 512 : 256, 2) void pair_mlp_fwd_kernel(PairFwdArgs a) {
    constexpr int K0 = FEAT + kPEnc, NT = X2 ? 512 : 256, NPART = NT / 128, NOB = X2 ? 1 : 2;
    constexpr int HP = X2 ? kPTile * kPitchR : 0;                 // byte distance hi plane -> lo plane
    constexpr int FQ = FEAT / NPART, EC = kPEnc / NPART;          // feature / encoding columns one thread fills
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    unsigned char* H = dsmem;
    float* wrow = reinterpret_cast<float*>(dsmem + (X2 ? 2 : 1) * kPTile * kPitchR);
    int* pstart = reinterpret_cast<int*>(wrow + kPTile);
    int* pcount = pstart + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ntiles = (a.P + 15) / 16;
    const int64_t wlo = X2 ? pl_mats(FEAT) : 0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int nblk, V;
        const int64_t row0 = a.off[tile * 16];                     // first pair row of the tile
        {
            const int row = tid & 127, part = __builtin_amdgcn_readfirstlane(tid >> 7);      // part is wave-uniform
            const int64_t p = tile * 16 + (row >> 3);
            const int slot = row & 7;
            int64_t gi = -1, gi_other = -1;
            if (p < a.P && slot < a.k) gi = a.nb_idx[p * a.k + slot];
            const int64_t po = tile * 16 + ((row ^ 64) >> 3);
            if (po < a.P && slot < a.k) gi_other = a.nb_idx[po * a.k + slot];
            const unsigned long long mine = __ballot(gi >= 0), other = __ballot(gi_other >= 0);
            const int n_mine = __popcll(mine), n_other = __popcll(other);
            const int prow = __popcll(mine & ((1ull << lane) - 1ull)) + ((row & 64) ? n_other : 0);
            V = n_mine + n_other;
            nblk = (V + 31) >> 5;
            if (part == 0 && slot == 0) {
                pstart[row >> 3] = prow;
                pcount[row >> 3] = __popcll((mine >> (lane & ~7)) & 0xffull);
            }
            float rel[3] = {0.f, 0.f, 0.f};
            if (gi >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rel[c] = a.pts[p * 3 + c] - a.kp_pos[gi * 3 + c];
            }
            if (part == 0 && gi >= 0) wrow[prow] = 1.f / (sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]) + 1e-5f);
            if (gi >= 0) {
                const float* fp = a.kp_feat + gi * FEAT + part * FQ;
#pragma unroll
                for (int c8 = 0; c8 < FQ / 8; ++c8) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(fp + c8 * 8);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(fp + c8 * 8 + 4);
                    const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    pm_put8<X2>(H + row_off(prow, part * (FQ / 8) + c8), HP, x);
                }
                // positional encoding: part fills EC of the 64 columns (the column numbers are compile-time inside each branch)
                unsigned char* ep = H + row_off(prow, FEAT / 8 + part * (EC / 8));
#pragma unroll
                for (int c8 = 0; c8 < EC / 8; ++c8) {
                    if (part == 0) pm_enc8<X2>(ep + c8 * 16, HP, rel, c8 * 8);
                    else if (part == 1) pm_enc8<X2>(ep + c8 * 16, HP, rel, EC + c8 * 8);
                    else if (part == 2) pm_enc8<X2>(ep + c8 * 16, HP, rel, (2 * EC + c8 * 8) % kPEnc);
                    else pm_enc8<X2>(ep + c8 * 16, HP, rel, (3 * EC + c8 * 8) % kPEnc);
                }
            }
            if (row >= V && row < 32 * nblk) {       // rows of a partly filled 32-row block: defined (zero) inputs
                const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c8 = 0; c8 < FQ / 8; ++c8) pm_put8<X2>(H + row_off(row, part * (FQ / 8) + c8), HP, z);
#pragma unroll
                for (int c8 = 0; c8 < EC / 8; ++c8) pm_put8<X2>(H + row_off(row, FEAT / 8 + part * (EC / 8) + c8), HP, z);
            }
        }
        __syncthreads();
        if (a.x0) {
            pm_copy_out<NT>(H, V, K0 * 2, reinterpret_cast<unsigned char*>(a.x0), row0, tid);
            if (X2) pm_copy_out<NT>(H + HP, V, K0 * 2, reinterpret_cast<unsigned char*>(a.x0 + a.Q * K0), row0, tid);
        }
        f32x16 acc[NOB][4];
        pm_layer_mfma<K0 / 16, NOB, 4, X2>(H, kPitchR, a.wpack + pl_fw(FEAT, 0), reinterpret_cast<const float*>(a.wpack + pl_bias(FEAT, 0, X2)), wave, lane,
                                           acc, nblk, HP, wlo);
        __syncthreads();
        pm_layer_store<true, NOB, 4, X2>(H, kPitchR, wave, lane, acc, nblk, HP);
        __syncthreads();
        if (a.acts) {
            pm_copy_out<NT>(H, V, kPH * 2, reinterpret_cast<unsigned char*>(a.acts), row0, tid);
            if (X2) pm_copy_out<NT>(H + HP, V, kPH * 2, reinterpret_cast<unsigned char*>(a.acts + a.Q * kPH), row0, tid);
        }
#pragma unroll 1
        for (int l = 1; l < 4; ++l) {
            pm_layer_mfma<kPH / 16, NOB, 4, X2>(H, kPitchR, a.wpack + pl_fw(FEAT, l), reinterpret_cast<const float*>(a.wpack + pl_bias(FEAT, l, X2)), wave,
                                                lane, acc, nblk, HP, wlo);
            __syncthreads();
            pm_layer_store<true, NOB, 4, X2>(H, kPitchR, wave, lane, acc, nblk, HP);
            __syncthreads();
            if (a.acts) {
                __bf16* dst = a.acts + (int64_t)l * (X2 ? 2 : 1) * a.Q * kPH;
                pm_copy_out<NT>(H, V, kPH * 2, reinterpret_cast<unsigned char*>(dst), row0, tid);
                if (X2) pm_copy_out<NT>(H + HP, V, kPH * 2, reinterpret_cast<unsigned char*>(dst + a.Q * kPH), row0, tid);
            }
        }
        if (tid < 256) {   // inverse-distance weighted mean over each point's pairs
            const int pl = tid >> 4, cc = tid & 15;
            const int64_t p = tile * 16 + pl;
            const int r0 = pstart[pl], cnt = pcount[pl];
            float wsum = 0.f;
            for (int s2 = 0; s2 < cnt; ++s2) wsum += wrow[r0 + s2];
            const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
            float out[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) out[j] = 0.f;
            for (int s2 = 0; s2 < cnt; ++s2) {
                const int row = r0 + s2;
                const float ws = wrow[row] * inv;
                if (cc == 0 && a.wn) a.wn[row0 + row] = ws;
                const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(H + row_off(row, 2 * cc));
                const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(H + row_off(row, 2 * cc + 1));
                if (X2) {
                    const bf16x8 l0 = *reinterpret_cast<const bf16x8*>(H + HP + row_off(row, 2 * cc));
                    const bf16x8 l1 = *reinterpret_cast<const bf16x8*>(H + HP + row_off(row, 2 * cc + 1));
#pragma unroll
                    for (int j = 0; j < 8; ++j) { out[j] += ws * ((float)v0[j] + (float)l0[j]); out[8 + j] += ws * ((float)v1[j] + (float)l1[j]); }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { out[j] += ws * (float)v0[j]; out[8 + j] += ws * (float)v1[j]; }
                }
            }
            if (p < a.P) {
                f32x4* gp = reinterpret_cast<f32x4*>(a.G + p * kPH + cc * 16);
#pragma unroll
                for (int g = 0; g < 4; ++g) gp[g] = f32x4{out[4 * g], out[4 * g + 1], out[4 * g + 2], out[4 * g + 3]};
            }
        }
        __syncthreads();
    }
}

// ============================================================================================
// backward, one layer
// ============================================================================================
struct PairBwdArgs {
    const unsigned char* wT;   // W_l^T fragments (data gradient); unused when there is no data gradient to produce
    int64_t wlo;               // x2: byte distance to the lo halves of the same fragments
    const float* dG;           // layer 3: [P, 256] gradient of the aggregated features
    const int64_t* owner;      // layer 3: [Q]
    const float* wn;           // layer 3: [Q]
    const __bf16* dA;          // layers 0..2: [Q, 256] gradient w.r.t. this layer's output      (x2: [2][Q][256], hi then lo)
    const __bf16* A;           // [Q, 256] this layer's output (sign of the LeakyReLU; x2: its hi plane)
    const __bf16* Aprev;       // [Q, KP] this layer's input                                      (x2: [2][Q][KP])
    int64_t Q;
    __bf16* dAprev;            // layers 1..3: [Q, 256]                                           (x2: [2][Q][256])
    float* dfeat;              // layer 0: [Q, FB*32] fp32 gradient w.r.t. the gathered features
    float* part;               // [gridDim.x][256 * KP + 256] fp32 slabs: dW_l then db_l
};

// transposed 32 x 16 fragment from a row-major LDS tile: k = rows 16 s + {4h..4h+3, 8+4h..8+4h+3} (same order for both operands),
// m / n = channels cb*32 + (lane & 31)
__device__ __forceinline__ bf16x8 pm_tr_frag(uint32_t base0, uint32_t base1, int imm) {
    u32x2 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(base0 + imm) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(base1 + imm) : "memory");
    const u32x4 x = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8, x);
}
__device__ __forceinline__ void pm_lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ uint32_t pm_lds_addr(const unsigned char* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
}

// LAYER3: dA is formed from dG / owner / wn.  KP = input width of the layer (256, or F + 64 for layer 0); FB = number of 32-wide
// input blocks whose data gradient is produced (8 for layers 1..3, F / 32 for layer 0).  X2: row tiles of 64 (two planes of dZ and
// two of A_{l-1} take the LDS the 128-row tile of the bf16 mode takes), three matrix instructions per fragment pair.
template <bool LAYER3, int KP, int FB, bool FEAT_OUT, bool X2>
__global__ __launch_bounds__(512, 2) void pair_mlp_bwd_kernel(PairBwdArgs a) {
    constexpr int TILE = X2 ? 64 : kPTile;   // rows per tile
    constexpr int NIB = KP / 32;             // dW column blocks per wave (one 32-row output block per wave)
    constexpr int PITCH_P = (KP * 2 + 255) / 256 * 256 + 64;     // previous activations (transposed reads only): pitch = 64 mod 256 bytes
    constexpr int ZP = TILE * kPitchR, PP = TILE * PITCH_P;      // bytes of one plane
    constexpr int NPL = X2 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    unsigned char* Z = dsmem;                               // [NPL][TILE][528]  dZ (bf16): row reads + transposed reads
    unsigned char* Pt = dsmem + NPL * ZP;                   // [NPL][TILE][PITCH_P] A_{l-1}; later the staging area of dA_{l-1}
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 dw[NIB];
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
        for (int i = 0; i < 16; ++i) dw[ib][i] = 0.f;
    float db[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) db[j] = 0.f;
    // per-lane bases of the transposed reads (see pm_tr_frag): 16-lane group g = lane >> 4 -> lane half h = g >> 1, column half g & 1
    const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3, th = grp >> 1;
    const uint32_t zt0 = pm_lds_addr(Z) + (4 * th + tq) * kPitchR + (16 * (grp & 1) + 4 * tp) * 2 + wave * 64;   // + 16 s rows, + ob*64 B
    const uint32_t zt1 = zt0 + 8 * kPitchR;
    const uint32_t pt0 = pm_lds_addr(Pt) + (4 * th + tq) * PITCH_P + (16 * (grp & 1) + 4 * tp) * 2;
    const uint32_t pt1 = pt0 + 8 * PITCH_P;
    const int64_t ntiles = (a.Q + TILE - 1) / TILE;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * TILE;
        // ---- load: dZ = dA * leaky'(A) -> Z, bias-gradient partials; A_{l-1} -> Pt -----------------------------------------
        // (the 128 dW accumulators stay live through this phase: a few rows in flight per thread, not all eight)
#pragma unroll LAYER3 ? 2 : 4
        for (int it = 0; it < TILE / 16; ++it) {
            const int cidx = it * 512 + tid, row = cidx >> 5, chunk = cidx & 31;       // chunk = tid & 31 for every it
            const int64_t q = row0 + row;
            float dz[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) dz[j] = 0.f;
            if (q < a.Q) {
                const bf16x8 act = *reinterpret_cast<const bf16x8*>(a.A + q * kPH + chunk * 8);
                float g[8];
                if (LAYER3) {
                    const int64_t p = a.owner[q];
                    const float w = a.wn[q];
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.dG + p * kPH + chunk * 8);
                    const f32x4 g1 = *reinterpret_cast<const f32x4*>(a.dG + p * kPH + chunk * 8 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { g[j] = w * g0[j]; g[4 + j] = w * g1[j]; }
                } else {
                    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(a.dA + q * kPH + chunk * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] = (float)gv[j];
                    if (X2) {
                        const bf16x8 gl = *reinterpret_cast<const bf16x8*>(a.dA + (a.Q + q) * kPH + chunk * 8);
#pragma unroll
                        for (int j = 0; j < 8; ++j) g[j] += (float)gl[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dz[j] = g[j] * ((float)act[j] > 0.f ? 1.f : kPSlope);
                    if (!X2) dz[j] = (float)(__bf16)dz[j];  // bf16 mode: the bias gradient sums the rounded values the matrix products see
                    db[j] += dz[j];
                }
            }
            pm_put8<X2>(Z + row_off(row, chunk), ZP, dz);
        }
        constexpr int CPR = KP / 8;                        // 16-byte chunks per row of A_{l-1}
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
            for (int c = tid; c < TILE * CPR; c += 512) {
                const int row = c / CPR, chunk = c - row * CPR;
                const int64_t q = row0 + row;
                u32x4 v = {0, 0, 0, 0};
                if (q < a.Q) v = *reinterpret_cast<const u32x4*>(a.Aprev + (pl * a.Q + q) * KP + chunk * 8);
                *reinterpret_cast<u32x4*>(Pt + pl * PP + row * PITCH_P + (chunk << 4)) = v;
            }
        __syncthreads();
        // ---- dW[o-block = wave][all input blocks] += dZ^T A_{l-1}  (K = the tile's rows) -------------------------------------
        {
            constexpr int KS = TILE / 16;
            bf16x8 zf[KS], zl[X2 ? KS : 1];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                zf[s] = pm_tr_frag(zt0, zt1, s * 16 * kPitchR);
                if (X2) zl[s] = pm_tr_frag(zt0, zt1, ZP + s * 16 * kPitchR);
            }
            pm_lds_wait();
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) {
#pragma unroll
                for (int s4 = 0; s4 < KS; s4 += 4) {         // four k-steps at a time: 16 transient registers instead of 32
                    bf16x8 pf[4], pfl[X2 ? 4 : 1];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        pf[s] = pm_tr_frag(pt0, pt1, (s4 + s) * 16 * PITCH_P + ib * 64);
                        if (X2) pfl[s] = pm_tr_frag(pt0, pt1, PP + (s4 + s) * 16 * PITCH_P + ib * 64);
                    }
                    pm_lds_wait();
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        if (X2) {
                            dw[ib] = BF16::mfma32(zl[s4 + s], pf[s], dw[ib]);
                            dw[ib] = BF16::mfma32(zf[s4 + s], pfl[s], dw[ib]);
                        }
                        dw[ib] = BF16::mfma32(zf[s4 + s], pf[s], dw[ib]);
                    }
                }
            }
        }
        // ---- dA_{l-1}^T[i-block][rows] = W^T dZ^T : wave w owns input block w (FB of them), all 32-row blocks ---------------------
        {
            const bool mine = wave < FB;                   // wave-uniform
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int half = 0; half < TILE / 64; ++half) { // 64-row halves: 32 accumulator registers at a time
                f32x16 acc[1][2];
                if (mine) pm_layer_mfma<kPH / 16, 1, 2, X2>(Z + half * 64 * kPitchR, kPitchR, a.wT, nullptr, wave, lane, acc, 2, ZP, a.wlo);
                if (half == 0) __syncthreads();            // every wave is done with Pt (the dW products' operand)
                if (mine) {
                    if (FEAT_OUT) {                        // layer 0: fp32 feature gradient rows, 4 consecutive columns per lane
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int64_t q = row0 + half * 64 + cb * 32 + r;
                                if (q < a.Q)
                                    *reinterpret_cast<f32x4*>(a.dfeat + q * (FB * 32) + wave * 32 + 8 * g + 4 * hh) =
                                        f32x4{acc[0][cb][4 * g], acc[0][cb][4 * g + 1], acc[0][cb][4 * g + 2], acc[0][cb][4 * g + 3]};
                            }
                    } else {
                        pm_layer_store<false, 1, 2, X2>(Pt + half * 64 * PITCH_P, PITCH_P, wave, lane, acc, 2, PP);
                    }
                }
            }
            if (!FEAT_OUT) {
                __syncthreads();
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl)
                    for (int c = tid; c < TILE * 32; c += 512) {
                        const int row = c >> 5, chunk = c & 31;
                        const int64_t q = row0 + row;
                        if (q < a.Q)
                            *reinterpret_cast<u32x4*>(a.dAprev + (pl * a.Q + q) * kPH + chunk * 8) =
                                *reinterpret_cast<const u32x4*>(Pt + pl * PP + row * PITCH_P + (chunk << 4));
                    }
            }
        }
        __syncthreads();
    }
    // ---- this workgroup's slab: dW rows [wave*32, +32) x KP, then the bias-gradient partial -----------------------------------
    float* slab = a.part + (int64_t)blockIdx.x * (kPH * KP + kPH);
#pragma unroll
    for (int ib = 0; ib < NIB; ++ib)
#pragma unroll
        for (int i = 0; i < 16; ++i) slab[(int64_t)(wave * 32 + acc_row(i, hh)) * KP + ib * 32 + r] = dw[ib][i];
    // bias: thread owns channels (tid & 31) * 8 .. + 8 over the rows it loaded; 16 threads share a channel group
    float* red = reinterpret_cast<float*>(dsmem);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(tid >> 5) * kPH + (tid & 31) * 8 + j] = db[j];
    __syncthreads();
    if (tid < kPH) {
        float s = 0.f;
        for (int g = 0; g < 16; ++g) s += red[g * kPH + tid];
        slab[(int64_t)kPH * KP + tid] = s;
    }
}

// out[e] = sum over the slabs in a FIXED order (bitwise reproducible): a workgroup owns 64 consecutive elements, its four waves
// each sum every fourth slab (independent chains: the loads of a chain are coalesced 256-byte rows), then one wave adds the four
// partial sums.
__global__ __launch_bounds__(256) void pair_slab_sum_kernel(const float* __restrict__ part, int nslabs, int64_t slab_elems, int64_t n_w,
                                                            float* __restrict__ dW, float* __restrict__ db) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (e < slab_elems) {
        int g = w;
        for (; g + 4 < nslabs; g += 8) {
            s0 += part[(int64_t)g * slab_elems + e];
            s1 += part[(int64_t)(g + 4) * slab_elems + e];
        }
        if (g < nslabs) s0 += part[(int64_t)g * slab_elems + e];
    }
    red[w][lane] = s0 + s1;
    __syncthreads();
    if (w == 0 && e < slab_elems) {
        const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        if (e < n_w) dW[e] = s;
        else db[e - n_w] = s;
    }
}

}  // namespace npcd

using namespace npcd;

static int pm_check(int feat_dim, int precision = 0) {
    if (precision != 0 && precision != 1) return NPCD_ERR_UNSUPPORTED;
    return (feat_dim == 32 || feat_dim == 128) ? NPCD_OK : NPCD_ERR_UNSUPPORTED;
}

extern "C" int64_t npcd_pair_mlp_wpack_bytes(int feat_dim, int precision) {
    if (pm_check(feat_dim, precision) != NPCD_OK) return -1;
    return pl_total(feat_dim, precision == 1);
}

extern "C" int npcd_pair_mlp_pack(const float* const* weights_dev, const float* const* biases_dev, int feat_dim, int precision, void* wpack_dev,
                                  void* stream) {
    int rc = pm_check(feat_dim, precision);
    if (rc != NPCD_OK) return rc;
    if (!weights_dev || !biases_dev || !wpack_dev) return NPCD_ERR_ARG;
    PackArgs a;
    for (int l = 0; l < 4; ++l) {
        if (!weights_dev[l] || !biases_dev[l]) return NPCD_ERR_ARG;
        a.W[l] = weights_dev[l];
        a.b[l] = biases_dev[l];
    }
    a.out = static_cast<unsigned char*>(wpack_dev);
    a.feat_dim = feat_dim;
    a.x2 = precision == 1;
    const int64_t pieces = pl_mats(feat_dim) / 16 * (a.x2 ? 2 : 1);
    hipLaunchKernelGGL(pair_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

template <int FEAT, bool X2>
static hipError_t pm_launch_fwd(const PairFwdArgs& a, int grid, hipStream_t st) {
    constexpr int lds = (X2 ? 2 : 1) * kPTile * kPitchR + kPTile * 4 + 32 * 4;
    static DynLds attr;
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(pair_mlp_fwd_kernel<FEAT, X2>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pair_mlp_fwd_kernel<FEAT, X2>), dim3(grid), dim3(X2 ? 512 : 256), lds, st, a);
    return hipGetLastError();
}

extern "C" int npcd_pair_mlp_fwd(const void* wpack, int feat_dim, int precision, const int64_t* nb_idx, const float* pts, const float* kp_pos,
                                 const float* kp_feat, const int64_t* off, int64_t n_points, int k, int64_t n_pairs, void* x0, void* acts,
                                 float* wn, float* G, void* stream) {
    int rc = pm_check(feat_dim, precision);
    if (rc != NPCD_OK) return rc;
    if (!wpack || !nb_idx || !pts || !kp_pos || !kp_feat || !off || !G) return NPCD_ERR_ARG;
    if ((x0 == nullptr) != (acts == nullptr) || (x0 == nullptr) != (wn == nullptr)) return NPCD_ERR_ARG;     // saved for the backward: all or none
    if (k <= 0 || k > 8 || n_points < 0 || n_pairs < 0) return NPCD_ERR_ARG;
    if (n_points == 0) return NPCD_OK;
    PairFwdArgs a;
    a.wpack = static_cast<const unsigned char*>(wpack);
    a.nb_idx = nb_idx; a.pts = pts; a.kp_pos = kp_pos; a.kp_feat = kp_feat; a.off = off;
    a.P = n_points; a.Q = n_pairs; a.k = k;
    a.x0 = static_cast<__bf16*>(x0); a.acts = static_cast<__bf16*>(acts); a.wn = wn; a.G = G;
    const int64_t tiles = (n_points + 15) / 16;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e;
    if (precision == 1) {
        const int grid = (int)(tiles < 256 ? tiles : 256);          // one 512-thread workgroup per CU
        e = feat_dim == 32 ? pm_launch_fwd<32, true>(a, grid, st) : pm_launch_fwd<128, true>(a, grid, st);
    } else {
        const int grid = (int)(tiles < 512 ? tiles : 512);
        e = feat_dim == 32 ? pm_launch_fwd<32, false>(a, grid, st) : pm_launch_fwd<128, false>(a, grid, st);
    }
    NPCD_HIP_CHECK(e);
    return NPCD_OK;
}

extern "C" int npcd_pair_mlp_bwd_slabs(int64_t n_pairs, int precision) {
    const int rows = precision == 1 ? 64 : kPTile;
    const int64_t tiles = (n_pairs + rows - 1) / rows;
    return (int)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
}

extern "C" int64_t npcd_pair_mlp_bwd_workspace_floats(int feat_dim, int64_t n_pairs, int precision) {
    if (pm_check(feat_dim, precision) != NPCD_OK) return -1;
    const int kmax = feat_dim + kPEnc > kPH ? feat_dim + kPEnc : kPH;
    return (int64_t)npcd_pair_mlp_bwd_slabs(n_pairs, precision) * ((int64_t)kPH * kmax + kPH) + (int64_t)kPH * kmax + kPH;   // slabs + one summed slab
}

template <bool L3, int KP, int FB, bool FO, bool X2>
static hipError_t pm_launch_bwd(const PairBwdArgs& a, int grid, hipStream_t st) {
    constexpr int TILE = X2 ? 64 : kPTile;
    constexpr int lds = (X2 ? 2 : 1) * (TILE * kPitchR + TILE * ((KP * 2 + 255) / 256 * 256 + 64));
    static DynLds attr;
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(pair_mlp_bwd_kernel<L3, KP, FB, FO, X2>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pair_mlp_bwd_kernel<L3, KP, FB, FO, X2>), dim3(grid), dim3(512), lds, st, a);
    return hipGetLastError();
}
template <bool X2>
static hipError_t pm_launch_bwd_layer(const PairBwdArgs& a, int l, int feat_dim, int grid, hipStream_t st) {
    if (l == 3) return pm_launch_bwd<true, 256, 8, false, X2>(a, grid, st);
    if (l > 0) return pm_launch_bwd<false, 256, 8, false, X2>(a, grid, st);
    if (feat_dim == 32) return pm_launch_bwd<false, 96, 1, true, X2>(a, grid, st);
    return pm_launch_bwd<false, 192, 4, true, X2>(a, grid, st);
}

// dG [P,256] fp32; x0 [Q,F+64], acts [4][Q][256] bf16 (from the forward; precision 1: two planes each, see PairFwdArgs); dact: workspace
// 2 x [Q][256] bf16 (precision 1: 2 x 2 x [Q][256]); dfeat [Q,F] fp32 out; part: npcd_pair_mlp_bwd_workspace_floats(); dW[l] fp32
// [256, in_l] (in_0 = F + 63), db[l] fp32 [256]: overwritten.
extern "C" int npcd_pair_mlp_bwd(const void* wpack, int feat_dim, int precision, const float* dG, const int64_t* owner, const float* wn,
                                 const void* x0, const void* acts, int64_t n_pairs, void* dact, float* dfeat, float* part, float* const* dW,
                                 float* const* db, void* stream) {
    int rc = pm_check(feat_dim, precision);
    if (rc != NPCD_OK) return rc;
    if (!wpack || !dG || !owner || !wn || !x0 || !acts || !dact || !dfeat || !part || !dW || !db) return NPCD_ERR_ARG;
    for (int l = 0; l < 4; ++l)
        if (!dW[l] || !db[l]) return NPCD_ERR_ARG;
    if (n_pairs <= 0) return NPCD_ERR_ARG;
    const bool x2 = precision == 1;
    const int npl = x2 ? 2 : 1;
    const unsigned char* wp = static_cast<const unsigned char*>(wpack);
    const __bf16* A = static_cast<const __bf16*>(acts);
    __bf16* d0 = static_cast<__bf16*>(dact);
    __bf16* d1 = d0 + (int64_t)npl * n_pairs * kPH;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = npcd_pair_mlp_bwd_slabs(n_pairs, precision);
    const int in0 = feat_dim + 3 + 6 * kPFreqs;
    for (int l = 3; l >= 0; --l) {
        PairBwdArgs a{};
        a.wT = wp + pl_bw(feat_dim, l);
        a.wlo = x2 ? pl_mats(feat_dim) : 0;
        a.dG = dG; a.owner = owner; a.wn = wn;
        a.A = A + (int64_t)l * npl * n_pairs * kPH;
        a.Aprev = l > 0 ? A + (int64_t)(l - 1) * npl * n_pairs * kPH : static_cast<const __bf16*>(x0);
        a.Q = n_pairs;
        a.dA = (l & 1) ? d1 : d0;          // written by layer l + 1
        a.dAprev = (l & 1) ? d0 : d1;
        a.dfeat = dfeat;
        a.part = part;
        NPCD_HIP_CHECK(x2 ? pm_launch_bwd_layer<true>(a, l, feat_dim, grid, st) : pm_launch_bwd_layer<false>(a, l, feat_dim, grid, st));
        const int KP = l > 0 ? kPH : feat_dim + kPEnc;
        const int64_t slab = (int64_t)kPH * KP + kPH;
        if (l > 0) {
            hipLaunchKernelGGL(pair_slab_sum_kernel, dim3((unsigned)((slab + 63) / 64)), dim3(256), 0, st, part, grid, slab, (int64_t)kPH * KP, dW[l], db[l]);
        } else {
            // layer 0: the slab rows are K0 = F + 64 wide, the weight gradient F + 63: sum into the head of the workspace, then copy rows
            float* tmp = part + (int64_t)grid * slab;
            hipLaunchKernelGGL(pair_slab_sum_kernel, dim3((unsigned)((slab + 63) / 64)), dim3(256), 0, st, part, grid, slab, (int64_t)kPH * KP, tmp, db[0]);
            NPCD_HIP_CHECK(hipMemcpy2DAsync(dW[0], (size_t)in0 * 4, tmp, (size_t)KP * 4, (size_t)in0 * 4, kPH, hipMemcpyDeviceToDevice, st));
        }
        NPCD_HIP_CHECK(hipGetLastError());
    }
    return NPCD_OK;
}
