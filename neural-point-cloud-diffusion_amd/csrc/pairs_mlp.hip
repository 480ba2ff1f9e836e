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
// Numerics: bf16 operands, fp32 accumulation, fp32 weight / bias gradients (what `PointNeRFTrainer(mlp_dtype=torch.bfloat16)`
// computed with library GEMMs before; the reference trains this stage in fp32, which stays the trainer's default).
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
__host__ __device__ inline int64_t pl_bias(int feat, int l) { return pl_bw(feat, 4) + (int64_t)l * kPH * 4; }
__host__ __device__ inline int64_t pl_total(int feat) { return pl_bias(feat, 4); }

struct PackArgs {
    const float* W[4];
    const float* b[4];
    unsigned char* out;
    int feat_dim;
};
// one thread per packed 16-byte fragment piece (8 elements)
__global__ __launch_bounds__(256) void pair_pack_kernel(PackArgs a) {
    const int F = a.feat_dim;
    const int in0 = a.feat_dim + 3 + 6 * kPFreqs;
    const int64_t piece = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t npieces = pl_bias(F, 0) / 16;
    if (piece < npieces) {
        const int64_t byte = piece * 16;
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
            v[j] = (__bf16)x;
        }
        *reinterpret_cast<bf16x8*>(a.out + byte) = v;
    }
    if (piece < 4 * kPH) {
        const int l = (int)(piece / kPH), c = (int)(piece % kPH);
        reinterpret_cast<float*>(a.out + pl_bias(F, l))[c] = a.b[l][c];
    }
}

__device__ __forceinline__ int row_off(int row, int chunk) { return row * kPitchR + (chunk << 4); }

// acc[oi][cb] = bias + W[(NOB*wave+oi)*32.., :] . H^T[:, cb*32..]   (activations = B operand, weights = A operand streamed from L2)
template <int KSTEPS, int NOB, int NCB>
__device__ __forceinline__ void pm_layer_mfma(const unsigned char* H, int pitch, const unsigned char* wfrag, const float* bias, int wave, int lane,
                                              f32x16 (&acc)[NOB][NCB], int nblk) {
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
    bf16x8 a[NOB], nx[NOB];
#pragma unroll
    for (int oi = 0; oi < NOB; ++oi) a[oi] = w[oi][0];
#pragma unroll 4
    for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
        for (int oi = 0; oi < NOB; ++oi) nx[oi] = s + 1 < KSTEPS ? w[oi][(s + 1) * 64] : a[oi];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
            if (cb < nblk) {                         // wave-uniform: whole 32-row blocks past the tile's rows are skipped
                const bf16x8 b = *reinterpret_cast<const bf16x8*>(hb + cb * 32 * pitch + s * 32);
#pragma unroll
                for (int oi = 0; oi < NOB; ++oi) acc[oi][cb] = BF16::mfma32(a[oi], b, acc[oi][cb]);
            }
        }
#pragma unroll
        for (int oi = 0; oi < NOB; ++oi) a[oi] = nx[oi];
    }
}

__device__ __forceinline__ uint32_t pm_pack2(float a, float b) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, b2));
}

// epilogue: optional LeakyReLU, convert to bf16, write in place: lane owns row (cb*32 + r), channels (NOB wave + oi)*32 + 8 g + 4 hh ..
template <bool ACT, int NOB, int NCB>
__device__ __forceinline__ void pm_layer_store(unsigned char* H, int pitch, int wave, int lane, const f32x16 (&acc)[NOB][NCB], int nblk) {
    const int r = lane & 31, hh = lane >> 5;
    unsigned char* sb = H + r * pitch + hh * 8 + wave * (NOB * 64);
#pragma unroll
    for (int oi = 0; oi < NOB; ++oi)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (cb >= nblk) continue;
                u32x2 v;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float x0 = acc[oi][cb][4 * g + 2 * b], x1 = acc[oi][cb][4 * g + 2 * b + 1];
                    if (ACT) {
                        x0 = fmaxf(x0, kPSlope * x0);
                        x1 = fmaxf(x1, kPSlope * x1);
                    }
                    v[b] = pm_pack2(x0, x1);
                }
                *reinterpret_cast<u32x2*>(sb + cb * 32 * pitch + (oi * 4 + g) * 16) = v;
            }
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
    __bf16* x0;                // [Q, F + 64]
    __bf16* acts;              // [4][Q][256]
    float* wn;                 // [Q]
    float* G;                  // [P, 256]
};

// copy the first `rows` rows of the LDS tile (width bytes each) to a row-major global array starting at row `row0`
__device__ __forceinline__ void pm_copy_out(const unsigned char* H, int rows, int width_bytes, unsigned char* dst, int64_t row0, int tid) {
    const int cpr = width_bytes >> 4;
    for (int c = tid; c < rows * cpr; c += 256) {
        const int row = c / cpr, chunk = c - row * cpr;
        *reinterpret_cast<u32x4*>(dst + (row0 + row) * width_bytes + (chunk << 4)) = *reinterpret_cast<const u32x4*>(H + row_off(row, chunk));
    }
}

// tile = 16 points x 8 neighbour slots; the tile's valid pairs are packed to the front (a point's rows stay consecutive)
template <int FEAT>
__global__ __launch_bounds__(256, 2) void pair_mlp_fwd_kernel(PairFwdArgs a) {
    constexpr int K0 = FEAT + kPEnc;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    unsigned char* H = dsmem;
    float* wrow = reinterpret_cast<float*>(dsmem + kPTile * kPitchR);
    int* pstart = reinterpret_cast<int*>(wrow + kPTile);
    int* pcount = pstart + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ntiles = (a.P + 15) / 16;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int nblk, V;
        const int64_t row0 = a.off[tile * 16];                     // first pair row of the tile
        {
            const int row = tid & 127, half = tid >> 7;
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
            if (half == 0 && slot == 0) {
                pstart[row >> 3] = prow;
                pcount[row >> 3] = __popcll((mine >> (lane & ~7)) & 0xffull);
            }
            float rel[3] = {0.f, 0.f, 0.f};
            if (gi >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rel[c] = a.pts[p * 3 + c] - a.kp_pos[gi * 3 + c];
            }
            if (half == 0 && gi >= 0) wrow[prow] = 1.f / (sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]) + 1e-5f);
            constexpr int FH = FEAT / 2;
            if (gi >= 0) {
                const float* fp = a.kp_feat + gi * FEAT + half * FH;
#pragma unroll
                for (int c8 = 0; c8 < FH / 8; ++c8) {
                    bf16x8 v;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(fp + c8 * 8);
                    const f32x4 x1 = *reinterpret_cast<const f32x4*>(fp + c8 * 8 + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = (__bf16)x0[j]; v[4 + j] = (__bf16)x1[j]; }
                    *reinterpret_cast<bf16x8*>(H + row_off(prow, half * (FH / 8) + c8)) = v;
                }
#pragma unroll
                for (int c8 = 0; c8 < 4; ++c8) {
                    bf16x8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(half == 0 ? pm_enc_value(c8 * 8 + j, rel) : pm_enc_value(32 + c8 * 8 + j, rel));
                    *reinterpret_cast<bf16x8*>(H + row_off(prow, FEAT / 8 + half * 4 + c8)) = v;
                }
            }
            if (row >= V && row < 32 * nblk) {       // rows of a partly filled 32-row block: defined (zero) inputs
                bf16x8 z;
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
#pragma unroll
                for (int c8 = 0; c8 < FH / 8; ++c8) *reinterpret_cast<bf16x8*>(H + row_off(row, half * (FH / 8) + c8)) = z;
#pragma unroll
                for (int c8 = 0; c8 < 4; ++c8) *reinterpret_cast<bf16x8*>(H + row_off(row, FEAT / 8 + half * 4 + c8)) = z;
            }
        }
        __syncthreads();
        pm_copy_out(H, V, K0 * 2, reinterpret_cast<unsigned char*>(a.x0), row0, tid);
        f32x16 acc[2][4];
        pm_layer_mfma<K0 / 16, 2, 4>(H, kPitchR, a.wpack + pl_fw(FEAT, 0), reinterpret_cast<const float*>(a.wpack + pl_bias(FEAT, 0)), wave, lane, acc, nblk);
        __syncthreads();
        pm_layer_store<true, 2, 4>(H, kPitchR, wave, lane, acc, nblk);
        __syncthreads();
        pm_copy_out(H, V, kPH * 2, reinterpret_cast<unsigned char*>(a.acts), row0, tid);
#pragma unroll 1
        for (int l = 1; l < 4; ++l) {
            pm_layer_mfma<kPH / 16, 2, 4>(H, kPitchR, a.wpack + pl_fw(FEAT, l), reinterpret_cast<const float*>(a.wpack + pl_bias(FEAT, l)), wave, lane, acc, nblk);
            __syncthreads();
            pm_layer_store<true, 2, 4>(H, kPitchR, wave, lane, acc, nblk);
            __syncthreads();
            pm_copy_out(H, V, kPH * 2, reinterpret_cast<unsigned char*>(a.acts + (int64_t)l * a.Q * kPH), row0, tid);
        }
        {   // inverse-distance weighted mean over each point's pairs
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
                if (cc == 0) a.wn[row0 + row] = ws;
                const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(H + row_off(row, 2 * cc));
                const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(H + row_off(row, 2 * cc + 1));
#pragma unroll
                for (int j = 0; j < 8; ++j) { out[j] += ws * (float)v0[j]; out[8 + j] += ws * (float)v1[j]; }
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
    const float* dG;           // layer 3: [P, 256] gradient of the aggregated features
    const int64_t* owner;      // layer 3: [Q]
    const float* wn;           // layer 3: [Q]
    const __bf16* dA;          // layers 0..2: [Q, 256] gradient w.r.t. this layer's output
    const __bf16* A;           // [Q, 256] this layer's output (sign of the LeakyReLU)
    const __bf16* Aprev;       // [Q, KP] this layer's input
    int64_t Q;
    __bf16* dAprev;            // layers 1..3: [Q, 256]
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
// input blocks whose data gradient is produced (8 for layers 1..3, F / 32 for layer 0).
template <bool LAYER3, int KP, int FB, bool FEAT_OUT>
__global__ __launch_bounds__(512, 2) void pair_mlp_bwd_kernel(PairBwdArgs a) {
    constexpr int NIB = KP / 32;             // dW column blocks per wave (one 32-row output block per wave)
    constexpr int PITCH_P = (KP * 2 + 255) / 256 * 256 + 64;     // previous activations (transposed reads only): pitch = 64 mod 256 bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    unsigned char* Z = dsmem;                               // [128][528]  dZ (bf16): row reads + transposed reads
    unsigned char* Pt = dsmem + kPTile * kPitchR;           // [128][PITCH_P] A_{l-1}; later the staging area of dA_{l-1}
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
    const int64_t ntiles = (a.Q + kPTile - 1) / kPTile;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * kPTile;
        // ---- load: dZ = dA * leaky'(A) -> Z, bias-gradient partials; A_{l-1} -> Pt -----------------------------------------
        // (the 128 dW accumulators stay live through this phase: a few rows in flight per thread, not all eight)
#pragma unroll LAYER3 ? 2 : 4
        for (int it = 0; it < 8; ++it) {
            const int cidx = it * 512 + tid, row = cidx >> 5, chunk = cidx & 31;       // chunk = tid & 31 for every it
            const int64_t q = row0 + row;
            bf16x8 z;
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
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
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    z[j] = (__bf16)(g[j] * ((float)act[j] > 0.f ? 1.f : kPSlope));
                    db[j] += (float)z[j];                  // the bias gradient sums the rounded values the matrix products see
                }
            }
            *reinterpret_cast<bf16x8*>(Z + row_off(row, chunk)) = z;
        }
        constexpr int CPR = KP / 8;                        // 16-byte chunks per row of A_{l-1}
        for (int c = tid; c < kPTile * CPR; c += 512) {
            const int row = c / CPR, chunk = c - row * CPR;
            const int64_t q = row0 + row;
            u32x4 v = {0, 0, 0, 0};
            if (q < a.Q) v = *reinterpret_cast<const u32x4*>(a.Aprev + q * KP + chunk * 8);
            *reinterpret_cast<u32x4*>(Pt + row * PITCH_P + (chunk << 4)) = v;
        }
        __syncthreads();
        // ---- dW[o-block = wave][all input blocks] += dZ^T A_{l-1}  (K = the tile's 128 rows) ---------------------------------
        {
            bf16x8 zf[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) zf[s] = pm_tr_frag(zt0, zt1, s * 16 * kPitchR);
            pm_lds_wait();
#pragma unroll
            for (int ib = 0; ib < NIB; ++ib) {
#pragma unroll
                for (int s4 = 0; s4 < 8; s4 += 4) {          // four k-steps at a time: 16 transient registers instead of 32
                    bf16x8 pf[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) pf[s] = pm_tr_frag(pt0, pt1, (s4 + s) * 16 * PITCH_P + ib * 64);
                    pm_lds_wait();
#pragma unroll
                    for (int s = 0; s < 4; ++s) dw[ib] = BF16::mfma32(zf[s4 + s], pf[s], dw[ib]);
                }
            }
        }
        // ---- dA_{l-1}^T[i-block][rows] = W^T dZ^T : wave w owns input block w (FB of them), all four 32-row blocks -----------
        {
            const bool mine = wave < FB;                   // wave-uniform
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int half = 0; half < 2; ++half) {         // two 64-row halves: 32 accumulator registers at a time
                f32x16 acc[1][2];
                if (mine) pm_layer_mfma<kPH / 16, 1, 2>(Z + half * 64 * kPitchR, kPitchR, a.wT, nullptr, wave, lane, acc, 2);
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
                        pm_layer_store<false, 1, 2>(Pt + half * 64 * PITCH_P, PITCH_P, wave, lane, acc, 2);
                    }
                }
            }
            if (!FEAT_OUT) {
                __syncthreads();
                for (int c = tid; c < kPTile * 32; c += 512) {
                    const int row = c >> 5, chunk = c & 31;
                    const int64_t q = row0 + row;
                    if (q < a.Q) *reinterpret_cast<u32x4*>(a.dAprev + q * kPH + chunk * 8) = *reinterpret_cast<const u32x4*>(Pt + row * PITCH_P + (chunk << 4));
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

static int pm_check(int feat_dim) { return (feat_dim == 32 || feat_dim == 128) ? NPCD_OK : NPCD_ERR_UNSUPPORTED; }

extern "C" int64_t npcd_pair_mlp_wpack_bytes(int feat_dim) {
    if (pm_check(feat_dim) != NPCD_OK) return -1;
    return pl_total(feat_dim);
}

extern "C" int npcd_pair_mlp_pack(const float* const* weights_dev, const float* const* biases_dev, int feat_dim, void* wpack_dev, void* stream) {
    int rc = pm_check(feat_dim);
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
    const int64_t pieces = pl_bias(feat_dim, 0) / 16;
    hipLaunchKernelGGL(pair_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_pair_mlp_fwd(const void* wpack, int feat_dim, const int64_t* nb_idx, const float* pts, const float* kp_pos,
                                 const float* kp_feat, const int64_t* off, int64_t n_points, int k, int64_t n_pairs, void* x0, void* acts,
                                 float* wn, float* G, void* stream) {
    int rc = pm_check(feat_dim);
    if (rc != NPCD_OK) return rc;
    if (!wpack || !nb_idx || !pts || !kp_pos || !kp_feat || !off || !x0 || !acts || !wn || !G) return NPCD_ERR_ARG;
    if (k <= 0 || k > 8 || n_points < 0 || n_pairs < 0) return NPCD_ERR_ARG;
    if (n_points == 0) return NPCD_OK;
    PairFwdArgs a;
    a.wpack = static_cast<const unsigned char*>(wpack);
    a.nb_idx = nb_idx; a.pts = pts; a.kp_pos = kp_pos; a.kp_feat = kp_feat; a.off = off;
    a.P = n_points; a.Q = n_pairs; a.k = k;
    a.x0 = static_cast<__bf16*>(x0); a.acts = static_cast<__bf16*>(acts); a.wn = wn; a.G = G;
    const int lds = kPTile * kPitchR + kPTile * 4 + 32 * 4;
    static DynLds l32, l128;
    NPCD_HIP_CHECK(l32.ensure(reinterpret_cast<const void*>(pair_mlp_fwd_kernel<32>), lds));
    NPCD_HIP_CHECK(l128.ensure(reinterpret_cast<const void*>(pair_mlp_fwd_kernel<128>), lds));
    const int64_t tiles = (n_points + 15) / 16;
    const int grid = (int)(tiles < 512 ? tiles : 512);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (feat_dim == 32) hipLaunchKernelGGL(pair_mlp_fwd_kernel<32>, dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(pair_mlp_fwd_kernel<128>, dim3(grid), dim3(256), lds, st, a);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_pair_mlp_bwd_slabs(int64_t n_pairs) {
    const int64_t tiles = (n_pairs + kPTile - 1) / kPTile;
    return (int)(tiles < 256 ? (tiles < 1 ? 1 : tiles) : 256);
}

extern "C" int64_t npcd_pair_mlp_bwd_workspace_floats(int feat_dim, int64_t n_pairs) {
    if (pm_check(feat_dim) != NPCD_OK) return -1;
    const int kmax = feat_dim + kPEnc > kPH ? feat_dim + kPEnc : kPH;
    return (int64_t)npcd_pair_mlp_bwd_slabs(n_pairs) * ((int64_t)kPH * kmax + kPH) + (int64_t)kPH * kmax + kPH;   // slabs + one summed slab
}

template <bool L3, int KP, int FB, bool FO>
static hipError_t pm_launch_bwd(const PairBwdArgs& a, int grid, hipStream_t st) {
    constexpr int lds = kPTile * kPitchR + kPTile * ((KP * 2 + 255) / 256 * 256 + 64);
    static DynLds attr;
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(pair_mlp_bwd_kernel<L3, KP, FB, FO>), lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((pair_mlp_bwd_kernel<L3, KP, FB, FO>), dim3(grid), dim3(512), lds, st, a);
    return hipGetLastError();
}

// dG [P,256] fp32; x0 [Q,F+64], acts [4][Q][256] bf16 (from the forward); dact: workspace 2 x [Q][256] bf16; dfeat [Q,F] fp32 out;
// part: npcd_pair_mlp_bwd_workspace_floats(); dW[l] fp32 [256, in_l] (in_0 = F + 63), db[l] fp32 [256]: overwritten.
extern "C" int npcd_pair_mlp_bwd(const void* wpack, int feat_dim, const float* dG, const int64_t* owner, const float* wn, const void* x0,
                                 const void* acts, int64_t n_pairs, void* dact, float* dfeat, float* part, float* const* dW, float* const* db,
                                 void* stream) {
    int rc = pm_check(feat_dim);
    if (rc != NPCD_OK) return rc;
    if (!wpack || !dG || !owner || !wn || !x0 || !acts || !dact || !dfeat || !part || !dW || !db) return NPCD_ERR_ARG;
    for (int l = 0; l < 4; ++l)
        if (!dW[l] || !db[l]) return NPCD_ERR_ARG;
    if (n_pairs <= 0) return NPCD_ERR_ARG;
    const unsigned char* wp = static_cast<const unsigned char*>(wpack);
    const __bf16* A = static_cast<const __bf16*>(acts);
    __bf16* d0 = static_cast<__bf16*>(dact);
    __bf16* d1 = d0 + n_pairs * kPH;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = npcd_pair_mlp_bwd_slabs(n_pairs);
    const int in0 = feat_dim + 3 + 6 * kPFreqs;
    for (int l = 3; l >= 0; --l) {
        PairBwdArgs a{};
        a.wT = wp + pl_bw(feat_dim, l);
        a.dG = dG; a.owner = owner; a.wn = wn;
        a.A = A + (int64_t)l * n_pairs * kPH;
        a.Aprev = l > 0 ? A + (int64_t)(l - 1) * n_pairs * kPH : static_cast<const __bf16*>(x0);
        a.Q = n_pairs;
        a.dA = (l & 1) ? d1 : d0;          // written by layer l + 1
        a.dAprev = (l & 1) ? d0 : d1;
        a.dfeat = dfeat;
        a.part = part;
        hipError_t e;
        if (l == 3) e = pm_launch_bwd<true, 256, 8, false>(a, grid, st);
        else if (l > 0) e = pm_launch_bwd<false, 256, 8, false>(a, grid, st);
        else if (feat_dim == 32) e = pm_launch_bwd<false, 96, 1, true>(a, grid, st);
        else e = pm_launch_bwd<false, 192, 4, true>(a, grid, st);
        NPCD_HIP_CHECK(e);
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
