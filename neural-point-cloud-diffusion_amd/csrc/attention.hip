// Attention over points for gfx950: out = softmax(q k^T * scale) v, non-causal, head_dim 64.
//
// Replaces flash_attn_func at npcd/models/diffusion/denoisers/transformer.py:75 (forward) and the
// flash-attn autograd backward.  q, k, v are strided [B, n, H, 64] views of the interleaved
// c_qkv output (transformer.py:71-72): no split / contiguous copies are made.
//
// Layout choice (all three kernels): every 32x32x16 MFMA is oriented so that the quantity that
// needs a per-row softmax statistic sits on the LANE (accumulator column) and the reduction
// index of the NEXT product sits in the accumulator ROWS (registers).  The accumulator of
// S^T = K Q^T can then be exponentiated lane-locally and fed, converted to 16 bit, straight back
// as the B operand of O^T = V^T P^T -- no LDS round trip for P (cdna_hip_programming.md §3,
// "An accumulator tile as the next MFMA's operand").
//
//   fwd  : one wave = 32 query rows, workgroup = 4 waves = 128 rows; 64-key K / V^T tiles are
//          double-buffered in LDS (32 KiB) and shared by the 4 waves.
//   dq   : same decomposition; per key tile S^T, dP^T = V dO^T, dQ^T += K^T dS^T; also emits
//          delta = rowsum(dO * O).
//   dkdv : one wave = 32 keys (on the lanes), workgroup = 128 keys; loops over 64-row query
//          tiles; S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.  No atomics: dq and
//          dk/dv come from two passes that each own their outputs (deterministic).
#include <math.h>

#include "common.h"

namespace npcd {

struct AttnParams {
    const void *q, *k, *v, *out, *dout;
    void *o_w, *dq, *dk, *dv;
    float* lse;
    float* delta;
    int B, n, H;
    int64_t sb, sn, sh;     // q/k/v strides (elements)
    int64_t osb, osn, osh;  // out / dout strides
    int64_t gsb, gsn, gsh;  // dq/dk/dv strides
    float scale, scale_log2;
};

// ---- tile staging (256 threads, 64 rows x 64 sixteen-bit elements) --------------------------
// row-major image: thread t -> row t>>2, 16-byte chunks 2(t&3), 2(t&3)+1
template <class E>
__device__ __forceinline__ void rm_load(u32x4 (&reg)[2], const E* base, int64_t row_stride, int row0, int nrows, int tid) {
    const int grow = row0 + (tid >> 2);
    if (grow < nrows) {
        const u32x4* p = reinterpret_cast<const u32x4*>(base + grow * row_stride + (tid & 3) * 16);
        reg[0] = p[0];
        reg[1] = p[1];
    } else {
        reg[0] = u32x4{0, 0, 0, 0};
        reg[1] = u32x4{0, 0, 0, 0};
    }
}
__device__ __forceinline__ void rm_store(unsigned char* lds, const u32x4 (&reg)[2], int tid) {
    const int row = tid >> 2, c0 = (tid & 3) * 2;
    *reinterpret_cast<u32x4*>(lds + tile_off(row, c0)) = reg[0];
    *reinterpret_cast<u32x4*>(lds + tile_off(row, c0 + 1)) = reg[1];
}
// transposed image T[d][pos(row)]: thread t -> rows 2(t>>3), 2(t>>3)+1, element chunk t&7.
// Within each group of 16 rows, row 8a+4h+b is stored at position 8h+4a+b, so that the 8 rows a
// lane-half h needs for one MFMA k-step (the accumulator-row order, acc_row()) are 16 contiguous
// bytes.
template <class E>
__device__ __forceinline__ void tr_load(u32x4 (&reg)[2], const E* base, int64_t row_stride, int row0, int nrows, int tid) {
    const int grow = row0 + 2 * (tid >> 3);
    const E* p = base + grow * row_stride + (tid & 7) * 8;
    reg[0] = (grow < nrows) ? *reinterpret_cast<const u32x4*>(p) : u32x4{0, 0, 0, 0};
    reg[1] = (grow + 1 < nrows) ? *reinterpret_cast<const u32x4*>(p + row_stride) : u32x4{0, 0, 0, 0};
}
__device__ __forceinline__ void tr_store(unsigned char* lds, const u32x4 (&reg)[2], int tid) {
    const int row = 2 * (tid >> 3), dc = tid & 7;
    const int g = row >> 4, kl = row & 15;
    const int a = kl >> 3, h = (kl >> 2) & 1, b = kl & 3;
    const int chunk = 2 * g + h, e = 4 * a + b;  // e is even
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t lo = (reg[0][i >> 1] >> (16 * (i & 1))) & 0xffffu;
        const uint32_t hi = (reg[1][i >> 1] >> (16 * (i & 1))) & 0xffffu;
        *reinterpret_cast<uint32_t*>(lds + tile_off(8 * dc + i, chunk) + e * 2) = lo | (hi << 16);
    }
}

template <class TR>
__device__ __forceinline__ typename TR::vec8 lds_frag(const unsigned char* tile, int row, int chunk) {
    return *reinterpret_cast<const typename TR::vec8*>(tile + tile_off(row, chunk));
}

// store one wave's 64(d) x 32(rows on lanes) transposed accumulator pair as rows of [.., 64]
template <class TR>
__device__ __forceinline__ void store_rows(typename TR::elem* row_ptr, const f32x16& a0, const f32x16& a1, float mul, int hh) {
    using V4 = typename TR::vec4;
    using E = typename TR::elem;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        V4 x, y;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            x[b] = (E)(a0[4 * g + b] * mul);
            y[b] = (E)(a1[4 * g + b] * mul);
        }
        *reinterpret_cast<V4*>(row_ptr + 8 * g + 4 * hh) = x;
        *reinterpret_cast<V4*>(row_ptr + 32 + 8 * g + 4 * hh) = y;
    }
}

// ============================================================================================
// forward
// ============================================================================================
template <class TR>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 16384];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int nqt = (p.n + 127) >> 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = bid % nqt, bh = bid / nqt, h = bh % p.H, b = bh / p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const int q0 = qt * 128 + wave * 32;
    const bool wave_active = q0 < p.n;
    const int qrow = q0 + r;

    V8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 raw = (qrow < p.n) ? *reinterpret_cast<const u32x4*>(qb + qrow * p.sn + 16 * s + 8 * hh) : u32x4{0, 0, 0, 0};
        qf[s] = __builtin_bit_cast(V8, raw);
    }
    f32x16 o0 = {0}, o1 = {0};
    float m = -INFINITY, l = 0.f;
    const int nt = (p.n + 63) >> 6;
    const float c = p.scale_log2;

    u32x4 kreg[2], vreg[2];
    rm_load(kreg, kb, p.sn, 0, p.n, tid);
    tr_load(vreg, vb, p.sn, 0, p.n, tid);
    rm_store(smem, kreg, tid);
    tr_store(smem + 8192, vreg, tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const unsigned char* Kc = smem + (t & 1) * 16384;
        const unsigned char* Vc = Kc + 8192;
        const bool more = t + 1 < nt;
        if (more) {
            rm_load(kreg, kb, p.sn, (t + 1) * 64, p.n, tid);
            tr_load(vreg, vb, p.sn, (t + 1) * 64, p.n, tid);
        }
        if (wave_active) {
            f32x16 s0 = {0}, s1 = {0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                s0 = TR::mfma32(lds_frag<TR>(Kc, r, 2 * s + hh), qf[s], s0);
                s1 = TR::mfma32(lds_frag<TR>(Kc, 32 + r, 2 * s + hh), qf[s], s1);
            }
            const bool partial = t * 64 + 64 > p.n;
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float x0 = s0[i] * c, x1 = s1[i] * c;
                if (partial) {
                    const int k0 = t * 64 + acc_row(i, hh);
                    if (k0 >= p.n) x0 = -INFINITY;
                    if (k0 + 32 >= p.n) x1 = -INFINITY;
                }
                s0[i] = x0;
                s1[i] = x1;
                mx = fmaxf(mx, fmaxf(x0, x1));
            }
            mx = fmaxf(mx, swap_half(mx));
            const float mn = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            float rs = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s0[i] = __builtin_amdgcn_exp2f(s0[i] - mn);
                s1[i] = __builtin_amdgcn_exp2f(s1[i] - mn);
                rs += s0[i] + s1[i];
            }
            l = l * alpha + rs;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o0[i] *= alpha;
                o1[i] *= alpha;
            }
            V8 pf[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                pf[0][j] = (E)s0[j];
                pf[1][j] = (E)s0[8 + j];
                pf[2][j] = (E)s1[j];
                pf[3][j] = (E)s1[8 + j];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                o0 = TR::mfma32(lds_frag<TR>(Vc, r, 2 * g + hh), pf[g], o0);
                o1 = TR::mfma32(lds_frag<TR>(Vc, 32 + r, 2 * g + hh), pf[g], o1);
            }
        }
        if (more) {
            unsigned char* Kn = smem + ((t + 1) & 1) * 16384;
            rm_store(Kn, kreg, tid);
            tr_store(Kn + 8192, vreg, tid);
        }
        __syncthreads();
    }
    l += swap_half(l);
    if (qrow < p.n) {
        E* orow = static_cast<E*>(p.o_w) + b * p.osb + qrow * p.osn + h * p.osh;
        store_rows<TR>(orow, o0, o1, 1.f / l, hh);
        if (hh == 0) p.lse[(int64_t)(b * p.H + h) * p.n + qrow] = m * kLn2 + logf(l);
    }
}

// ============================================================================================
// backward, pass 1: dQ (+ delta)
// ============================================================================================
template <class TR>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 24576];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int nqt = (p.n + 127) >> 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = bid % nqt, bh = bid / nqt, h = bh % p.H, b = bh / p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const E* ob = static_cast<const E*>(p.out) + b * p.osb + h * p.osh;
    const E* dob = static_cast<const E*>(p.dout) + b * p.osb + h * p.osh;
    const int q0 = qt * 128 + wave * 32;
    const bool wave_active = q0 < p.n;
    const int qrow = q0 + r;
    const bool row_ok = qrow < p.n;

    V8 qf[4], dof[4];
    float delta = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const u32x4 z = {0, 0, 0, 0};
        qf[s] = __builtin_bit_cast(V8, row_ok ? *reinterpret_cast<const u32x4*>(qb + qrow * p.sn + 16 * s + 8 * hh) : z);
        dof[s] = __builtin_bit_cast(V8, row_ok ? *reinterpret_cast<const u32x4*>(dob + qrow * p.osn + 16 * s + 8 * hh) : z);
        V8 of = __builtin_bit_cast(V8, row_ok ? *reinterpret_cast<const u32x4*>(ob + qrow * p.osn + 16 * s + 8 * hh) : z);
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += (float)dof[s][j] * (float)of[j];
    }
    delta += swap_half(delta);
    const int64_t stat = (int64_t)(b * p.H + h) * p.n + qrow;
    if (row_ok && hh == 0) p.delta[stat] = delta;
    const float lse2 = row_ok ? p.lse[stat] * kLog2e : INFINITY;

    f32x16 dq0 = {0}, dq1 = {0};
    const int nt = (p.n + 63) >> 6;
    const float c = p.scale_log2;

    u32x4 kreg[2], vreg[2], ktreg[2];
    rm_load(kreg, kb, p.sn, 0, p.n, tid);
    rm_load(vreg, vb, p.sn, 0, p.n, tid);
    tr_load(ktreg, kb, p.sn, 0, p.n, tid);
    rm_store(smem, kreg, tid);
    rm_store(smem + 8192, vreg, tid);
    tr_store(smem + 16384, ktreg, tid);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const unsigned char* Kc = smem + (t & 1) * 24576;
        const unsigned char* Vc = Kc + 8192;
        const unsigned char* KTc = Kc + 16384;
        const bool more = t + 1 < nt;
        if (more) {
            rm_load(kreg, kb, p.sn, (t + 1) * 64, p.n, tid);
            rm_load(vreg, vb, p.sn, (t + 1) * 64, p.n, tid);
            tr_load(ktreg, kb, p.sn, (t + 1) * 64, p.n, tid);
        }
        if (wave_active) {
            f32x16 s0 = {0}, s1 = {0}, d0 = {0}, d1 = {0};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                s0 = TR::mfma32(lds_frag<TR>(Kc, r, 2 * s + hh), qf[s], s0);
                s1 = TR::mfma32(lds_frag<TR>(Kc, 32 + r, 2 * s + hh), qf[s], s1);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                d0 = TR::mfma32(lds_frag<TR>(Vc, r, 2 * s + hh), dof[s], d0);
                d1 = TR::mfma32(lds_frag<TR>(Vc, 32 + r, 2 * s + hh), dof[s], d1);
            }
            const bool partial = t * 64 + 64 > p.n;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float p0 = __builtin_amdgcn_exp2f(s0[i] * c - lse2);
                float p1 = __builtin_amdgcn_exp2f(s1[i] * c - lse2);
                if (partial) {
                    const int k0 = t * 64 + acc_row(i, hh);
                    if (k0 >= p.n) p0 = 0.f;
                    if (k0 + 32 >= p.n) p1 = 0.f;
                }
                s0[i] = p0 * (d0[i] - delta);
                s1[i] = p1 * (d1[i] - delta);
            }
            V8 df[4];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                df[0][j] = (E)s0[j];
                df[1][j] = (E)s0[8 + j];
                df[2][j] = (E)s1[j];
                df[3][j] = (E)s1[8 + j];
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                dq0 = TR::mfma32(lds_frag<TR>(KTc, r, 2 * g + hh), df[g], dq0);
                dq1 = TR::mfma32(lds_frag<TR>(KTc, 32 + r, 2 * g + hh), df[g], dq1);
            }
        }
        if (more) {
            unsigned char* Kn = smem + ((t + 1) & 1) * 24576;
            rm_store(Kn, kreg, tid);
            rm_store(Kn + 8192, vreg, tid);
            tr_store(Kn + 16384, ktreg, tid);
        }
        __syncthreads();
    }
    if (row_ok) {
        E* grow = static_cast<E*>(p.dq) + b * p.gsb + qrow * p.gsn + h * p.gsh;
        store_rows<TR>(grow, dq0, dq1, p.scale, hh);
    }
}

// ============================================================================================
// backward, pass 2: dK, dV
// ============================================================================================
constexpr int kDkdvBuf = 4 * 8192 + 512;  // Q, dO row-major; Q^T, dO^T; lse2[64], delta[64]

template <class TR>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int nkt = (p.n + 127) >> 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int kt = bid % nkt, bh = bid / nkt, h = bh % p.H, b = bh / p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const E* dob = static_cast<const E*>(p.dout) + b * p.osb + h * p.osh;
    const float* lse = p.lse + (int64_t)(b * p.H + h) * p.n;
    const float* dlt = p.delta + (int64_t)(b * p.H + h) * p.n;
    const int key0 = kt * 128 + wave * 32;
    const bool wave_active = key0 < p.n;
    const int key = key0 + r;
    const bool key_ok = key < p.n;

    V8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const u32x4 z = {0, 0, 0, 0};
        kf[s] = __builtin_bit_cast(V8, key_ok ? *reinterpret_cast<const u32x4*>(kb + key * p.sn + 16 * s + 8 * hh) : z);
        vf[s] = __builtin_bit_cast(V8, key_ok ? *reinterpret_cast<const u32x4*>(vb + key * p.sn + 16 * s + 8 * hh) : z);
    }
    f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
    const int nt = (p.n + 63) >> 6;
    const float c = p.scale_log2;

    u32x4 qreg[2], doreg[2], qtreg[2], dotreg[2];
    float stat = 0.f;  // threads 0..63 stage lse2, 64..127 stage delta
    auto load_stats = [&](int row0) {
        if (tid < 128) {
            const int row = row0 + (tid & 63);
            if (tid < 64) stat = (row < p.n) ? lse[row] * kLog2e : INFINITY;
            else stat = (row < p.n) ? dlt[row] : 0.f;
        }
    };
    auto stage_load = [&](int row0) {
        rm_load(qreg, qb, p.sn, row0, p.n, tid);
        rm_load(doreg, dob, p.osn, row0, p.n, tid);
        tr_load(qtreg, qb, p.sn, row0, p.n, tid);
        tr_load(dotreg, dob, p.osn, row0, p.n, tid);
        load_stats(row0);
    };
    auto stage_store = [&](unsigned char* buf) {
        rm_store(buf, qreg, tid);
        rm_store(buf + 8192, doreg, tid);
        tr_store(buf + 16384, qtreg, tid);
        tr_store(buf + 24576, dotreg, tid);
        if (tid < 128) reinterpret_cast<float*>(buf + 32768)[tid] = stat;
    };
    stage_load(0);
    stage_store(dsmem);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const unsigned char* Qc = dsmem + (t & 1) * kDkdvBuf;
        const unsigned char* DOc = Qc + 8192;
        const unsigned char* QTc = Qc + 16384;
        const unsigned char* DOTc = Qc + 24576;
        const float* st = reinterpret_cast<const float*>(Qc + 32768);
        const bool more = t + 1 < nt;
        if (more) stage_load((t + 1) * 64);
        if (wave_active) {
            V8 pf[4], df[4];
#pragma unroll
            for (int qb2 = 0; qb2 < 2; ++qb2) {
                f32x16 s = {0}, d = {0};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) s = TR::mfma32(lds_frag<TR>(Qc, qb2 * 32 + r, 2 * ks + hh), kf[ks], s);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) d = TR::mfma32(lds_frag<TR>(DOc, qb2 * 32 + r, 2 * ks + hh), vf[ks], d);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(st + qb2 * 32 + 8 * g + 4 * hh);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(st + 64 + qb2 * 32 + 8 * g + 4 * hh);
#pragma unroll
                    for (int bq = 0; bq < 4; ++bq) {
                        const int i = 4 * g + bq;
                        const float pr = __builtin_amdgcn_exp2f(s[i] * c - l4[bq]);
                        s[i] = pr;
                        d[i] = pr * (d[i] - d4[bq]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    pf[2 * qb2][j] = (E)s[j];
                    pf[2 * qb2 + 1][j] = (E)s[8 + j];
                    df[2 * qb2][j] = (E)d[j];
                    df[2 * qb2 + 1][j] = (E)d[8 + j];
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                dv0 = TR::mfma32(lds_frag<TR>(DOTc, r, 2 * g + hh), pf[g], dv0);
                dv1 = TR::mfma32(lds_frag<TR>(DOTc, 32 + r, 2 * g + hh), pf[g], dv1);
                dk0 = TR::mfma32(lds_frag<TR>(QTc, r, 2 * g + hh), df[g], dk0);
                dk1 = TR::mfma32(lds_frag<TR>(QTc, 32 + r, 2 * g + hh), df[g], dk1);
            }
        }
        if (more) stage_store(dsmem + ((t + 1) & 1) * kDkdvBuf);
        __syncthreads();
    }
    if (key_ok) {
        E* gk = static_cast<E*>(p.dk) + b * p.gsb + key * p.gsn + h * p.gsh;
        E* gv = static_cast<E*>(p.dv) + b * p.gsb + key * p.gsn + h * p.gsh;
        store_rows<TR>(gk, dk0, dk1, p.scale, hh);
        store_rows<TR>(gv, dv0, dv1, 1.f, hh);
    }
}

// ============================================================================================
// host entry points
// ============================================================================================
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

static int check_common(int B, int n, int H, int d, int dtype) {
    if (B <= 0 || n <= 0 || H <= 0) return NPCD_ERR_ARG;
    if (d != 64) return NPCD_ERR_UNSUPPORTED;
    if (dtype != NPCD_BF16 && dtype != NPCD_F16) return NPCD_ERR_UNSUPPORTED;
    return NPCD_OK;
}
static bool strides_ok(int64_t sb, int64_t sn, int64_t sh) { return (sb % 8 == 0) && (sn % 8 == 0) && (sh % 8 == 0); }

}  // namespace npcd

using namespace npcd;

extern "C" int npcd_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int n, int H, int d,
                             int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                             float scale, int dtype, void* stream) {
    int rc = check_common(B, n, H, d, dtype);
    if (rc != NPCD_OK) return rc;
    if (!q || !k || !v || !out || !lse) return NPCD_ERR_ARG;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out)) return NPCD_ERR_ARG;
    if (!strides_ok(qkv_sb, qkv_sn, qkv_sh) || !strides_ok(out_sb, out_sn, out_sh)) return NPCD_ERR_ARG;
    AttnParams p{};
    p.q = q; p.k = k; p.v = v; p.o_w = out; p.lse = lse;
    p.B = B; p.n = n; p.H = H;
    p.sb = qkv_sb; p.sn = qkv_sn; p.sh = qkv_sh;
    p.osb = out_sb; p.osn = out_sn; p.osh = out_sh;
    p.scale = scale; p.scale_log2 = scale * kLog2e;
    const int grid = B * H * ceil_div(n, 128);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == NPCD_BF16) hipLaunchKernelGGL(attn_fwd_kernel<BF16>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<F16>, dim3(grid), dim3(256), 0, st, p);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

static int attn_bwd_launch(int passes, const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                           void* dq, void* dk, void* dv, float* delta, int B, int n, int H, int d,
                           int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                           int64_t g_sb, int64_t g_sn, int64_t g_sh, float scale, int dtype, void* stream) {
    int rc = check_common(B, n, H, d, dtype);
    if (rc != NPCD_OK) return rc;
    if (!q || !k || !v || !out || !dout || !lse || !delta) return NPCD_ERR_ARG;
    if ((passes & 1) && !dq) return NPCD_ERR_ARG;
    if ((passes & 2) && (!dk || !dv)) return NPCD_ERR_ARG;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || !aligned16(dout) || !aligned16(dq) ||
        !aligned16(dk) || !aligned16(dv))
        return NPCD_ERR_ARG;
    if (!strides_ok(qkv_sb, qkv_sn, qkv_sh) || !strides_ok(out_sb, out_sn, out_sh) || !strides_ok(g_sb, g_sn, g_sh)) return NPCD_ERR_ARG;
    AttnParams p{};
    p.q = q; p.k = k; p.v = v; p.out = out; p.dout = dout; p.lse = const_cast<float*>(lse); p.delta = delta;
    p.dq = dq; p.dk = dk; p.dv = dv;
    p.B = B; p.n = n; p.H = H;
    p.sb = qkv_sb; p.sn = qkv_sn; p.sh = qkv_sh;
    p.osb = out_sb; p.osn = out_sn; p.osh = out_sh;
    p.gsb = g_sb; p.gsn = g_sn; p.gsh = g_sh;
    p.scale = scale; p.scale_log2 = scale * kLog2e;
    const int grid = B * H * ceil_div(n, 128);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int dyn = 2 * kDkdvBuf;
    static bool attr_set = false;
    if (!attr_set) {
        NPCD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<BF16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
        NPCD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<F16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
        attr_set = true;
    }
    if (dtype == NPCD_BF16) {
        if (passes & 1) hipLaunchKernelGGL(attn_bwd_dq_kernel<BF16>, dim3(grid), dim3(256), 0, st, p);
        if (passes & 2) hipLaunchKernelGGL(attn_bwd_dkdv_kernel<BF16>, dim3(grid), dim3(256), dyn, st, p);
    } else {
        if (passes & 1) hipLaunchKernelGGL(attn_bwd_dq_kernel<F16>, dim3(grid), dim3(256), 0, st, p);
        if (passes & 2) hipLaunchKernelGGL(attn_bwd_dkdv_kernel<F16>, dim3(grid), dim3(256), dyn, st, p);
    }
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                             void* dq, void* dk, void* dv, float* delta, int B, int n, int H, int d,
                             int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                             int64_t g_sb, int64_t g_sn, int64_t g_sh, float scale, int dtype, void* stream) {
    return attn_bwd_launch(3, q, k, v, out, dout, lse, dq, dk, dv, delta, B, n, H, d, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh,
                           g_sb, g_sn, g_sh, scale, dtype, stream);
}

extern "C" int npcd_attn_bwd_pass(int pass, const void* q, const void* k, const void* v, const void* out, const void* dout,
                                  const float* lse, void* dq, void* dk, void* dv, float* delta, int B, int n, int H, int d,
                                  int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                                  int64_t g_sb, int64_t g_sn, int64_t g_sh, float scale, int dtype, void* stream) {
    if (pass != 1 && pass != 2) return NPCD_ERR_ARG;
    return attn_bwd_launch(pass, q, k, v, out, dout, lse, dq, dk, dv, delta, B, n, H, d, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh,
                           g_sb, g_sn, g_sh, scale, dtype, stream);
}
