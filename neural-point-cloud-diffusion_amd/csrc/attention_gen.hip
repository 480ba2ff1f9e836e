// Attention over points for head dims other than 64 (gfx950): out = softmax(q k^T * scale) v, non-causal, D in {32, 128} (the
// template also instantiates 64, which the tests run against the specialised kernels of attention.hip).
//
// Replaces flash_attn_func at npcd/models/diffusion/denoisers/transformer.py:75 for `width / heads` != 64 (the reference's
// QKVMultiheadAttention works for any head width, transformer.py:68-84) and the flash-attn autograd backward.  q, k, v are strided
// [B, n, H, D] views (the interleaved c_qkv output is consumed in place, like in attention.hip).
//
// Same orientation as attention.hip -- the softmax row on the LANE, the next product's reduction index in the accumulator rows, so that
// an exponentiated accumulator (packed to 16 bit) is directly the B operand of the following matrix instruction and P never touches
// LDS -- in a plain structure: 64-row tiles of the streamed operand go global -> registers -> LDS (the registers of tile t + 1 are
// requested before tile t is computed), one LDS buffer, two barriers per tile; row-major tiles with the XOR swizzle of common.h per
// 64-column sub-tile, the transposed operand read with ds_read_b64_tr_b16.  No LDS-DMA ring, no seeds, no edge kernel: this is the
// correct-and-reasonable path for the head widths outside the published configuration, not the tuned one.
//   fwd  : wave = 32 query rows, workgroup = 128 rows; K / V tiles streamed.
//   dq   : same decomposition; S^T, dP^T = V dO^T, dS^T = P (dP - delta), dQ^T += K^T dS^T; writes delta = rowsum(dO * O).
//   dkdv : wave = 32 keys (on the lanes); Q / dO tiles streamed; dV^T += dO^T P, dK^T += Q^T dS.  No atomics (deterministic).
#include <math.h>

#include "attention_gen.h"
#include "common.h"

namespace npcd {

namespace {

struct GenParams {
    const void *q, *k, *v, *out, *dout;
    void *o_w, *dq, *dk, *dv;
    float* lse;
    float* delta;
    int B, n, H;
    int64_t sb, sn, sh;     // q / k / v strides (elements)
    int64_t osb, osn, osh;  // out / dout strides
    int64_t gsb, gsn, gsh;  // dq / dk / dv strides
    float scale, scale_log2;
};

constexpr int kSubBytes = 64 * 128;      // one 64-row x 64-column sub-tile (128-byte rows; D = 32 uses the first 64 bytes of a row)

template <int D>
struct Geo {
    static constexpr int NSUB = D >= 64 ? D / 64 : 1;       // 64-column sub-tiles per tile
    static constexpr int TILE = NSUB * kSubBytes;          // bytes of one 64-row tile
    static constexpr int CPR = D / 8;                      // 16-byte chunks per row
    static constexpr int NCH = 64 * CPR / 256;             // chunks per thread and tile (256 threads)
    static constexpr int KS = D / 16;                      // k-steps of a product contracted over D
    static constexpr int DB = D / 32;                      // 32-wide blocks of D
};

typedef __attribute__((address_space(3))) unsigned char lds_byte;

// ---- tile staging: global -> registers -> LDS -----------------------------------------------------------------------------------
template <class E, int D>
__device__ __forceinline__ void tile_fetch(u32x4 (&reg)[Geo<D>::NCH], const E* base, int64_t row_stride, int row0, int n, int tid) {
#pragma unroll
    for (int i = 0; i < Geo<D>::NCH; ++i) {
        const int idx = tid + 256 * i, row = idx / Geo<D>::CPR, chunk = idx % Geo<D>::CPR;
        const int grow = min(row0 + row, n - 1);           // rows past the end are clamped; their contribution is masked downstream
        reg[i] = *reinterpret_cast<const u32x4*>(base + (int64_t)grow * row_stride + chunk * 8);
    }
}
template <int D>
__device__ __forceinline__ void tile_put(unsigned char* buf, const u32x4 (&reg)[Geo<D>::NCH], int tid) {
#pragma unroll
    for (int i = 0; i < Geo<D>::NCH; ++i) {
        const int idx = tid + 256 * i, row = idx / Geo<D>::CPR, chunk = idx % Geo<D>::CPR;
        *reinterpret_cast<u32x4*>(buf + (chunk >> 3) * kSubBytes + tile_off(row, chunk & 7)) = reg[i];
    }
}

// row fragment: the 8 elements d = 16 s + 8 hh .. of tile row `row` (an MFMA operand whose k index is the head dimension)
template <class TR>
__device__ __forceinline__ typename TR::vec8 row_frag(const unsigned char* tile, int row, int s, int hh) {
    const int chunk = 2 * s + hh;
    return *reinterpret_cast<const typename TR::vec8*>(tile + (chunk >> 3) * kSubBytes + tile_off(row, chunk & 7));
}
// transposed fragment: A operand T^T[d = 32 db + (lane & 31)][k] whose 16 k indices are tile rows row0 .. row0 + 15 in the
// accumulator-row order (element j of lane half h <-> row row0 + 8 (j >> 2) + 4 h + (j & 3)): two ds_read_b64_tr_b16
template <class TR>
__device__ __forceinline__ typename TR::vec8 tr_frag(const unsigned char* tile, int row0, int db, int lane) {
    const int grp = lane >> 4, i = lane & 15, qq = i >> 2, pp = i & 3, h = grp >> 1;
    const int col = (db & 1) * 32 + 16 * (grp & 1) + 4 * pp;
    const unsigned char* sub = tile + (db >> 1) * kSubBytes;
    const typename TR::vec4 lo = TR::tr_read(sub + tile_off(row0 + 4 * h + qq, col >> 3) + (col & 7) * 2);
    const typename TR::vec4 hi = TR::tr_read(sub + tile_off(row0 + 4 * h + qq + 8, col >> 3) + (col & 7) * 2);
    typename TR::vec8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = lo[j];
        v[4 + j] = hi[j];
    }
    return v;
}

template <class TR>
__device__ __forceinline__ typename TR::vec8 pack8(const f32x16& a, int g16) {
    typename TR::vec8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (typename TR::elem)a[8 * g16 + j];
    return v;
}

__device__ __forceinline__ float xhalf_max(float x) { return fmaxf(x, __shfl_xor(x, 32, 64)); }
__device__ __forceinline__ float xhalf_sum(float x) { return x + __shfl_xor(x, 32, 64); }

// the lane's row of a transposed accumulator set (d = 32 db + 8 g + 4 hh + j) -> 16-bit, 8-byte stores
template <class TR, int D>
__device__ __forceinline__ void store_row(typename TR::elem* row_ptr, const f32x16 (&acc)[Geo<D>::DB], float mul, int hh) {
    using E = typename TR::elem;
#pragma unroll
    for (int db = 0; db < Geo<D>::DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typename TR::vec4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (E)(acc[db][4 * g + j] * mul);
            *reinterpret_cast<typename TR::vec4*>(row_ptr + 32 * db + 8 * g + 4 * hh) = v;
        }
}

// ============================================================================================================================ forward
template <class TR, int D>
__global__ __launch_bounds__(256) void attn_gen_fwd_kernel(GenParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    using G = Geo<D>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G::TILE];      // K tile, V tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int rb = (p.n + 127) / 128;
    const int bh = blockIdx.x / rb, blk = blockIdx.x % rb, b = bh / p.H, h = bh % p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const int qrow = blk * 128 + wave * 32 + r, qld = min(qrow, p.n - 1);
    V8 qf[G::KS];
#pragma unroll
    for (int s = 0; s < G::KS; ++s) qf[s] = *reinterpret_cast<const V8*>(qb + (int64_t)qld * p.sn + 16 * s + 8 * hh);
    f32x16 o[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) o[db] = f32x16{0};
    float m = -INFINITY, l = 0.f;
    const int nt = (p.n + 63) / 64;
    u32x4 kreg[G::NCH], vreg[G::NCH];
    tile_fetch<E, D>(kreg, kb, p.sn, 0, p.n, tid);
    tile_fetch<E, D>(vreg, vb, p.sn, 0, p.n, tid);
    for (int t = 0; t < nt; ++t) {
        tile_put<D>(smem, kreg, tid);
        tile_put<D>(smem + G::TILE, vreg, tid);
        __syncthreads();
        if (t + 1 < nt) {
            tile_fetch<E, D>(kreg, kb, p.sn, (t + 1) * 64, p.n, tid);
            tile_fetch<E, D>(vreg, vb, p.sn, (t + 1) * 64, p.n, tid);
        }
        const int key0 = t * 64;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 s = {0};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) s = TR::mfma32(row_frag<TR>(smem, 32 * half + r, ks, hh), qf[ks], s);
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = key0 + 32 * half + acc_row(i, hh);
                s[i] = key < p.n ? s[i] * p.scale_log2 : -INFINITY;
                mx = fmaxf(mx, s[i]);
            }
            mx = xhalf_max(mx);
            const float m_new = fmaxf(m, mx);                       // (tile 0 holds key 0: finite from the first half on ... unless this
            const float m_use = m_new == -INFINITY ? 0.f : m_new;   //  half is entirely past the end, which leaves the state untouched)
            const float alpha = __builtin_amdgcn_exp2f(m - m_use);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[i] = __builtin_amdgcn_exp2f(s[i] - m_use);
                sum += s[i];
            }
            l = l * alpha + xhalf_sum(sum);
            m = m_new;
#pragma unroll
            for (int db = 0; db < G::DB; ++db) o[db] *= alpha;
#pragma unroll
            for (int g16 = 0; g16 < 2; ++g16) {
                const V8 pk = pack8<TR>(s, g16);
#pragma unroll
                for (int db = 0; db < G::DB; ++db) o[db] = TR::mfma32(tr_frag<TR>(smem + G::TILE, 32 * half + 16 * g16, db, lane), pk, o[db]);
            }
        }
        __syncthreads();
    }
    if (qrow < p.n) {
        store_row<TR, D>(static_cast<E*>(p.o_w) + b * p.osb + (int64_t)qrow * p.osn + h * p.osh, o, 1.f / l, hh);
        if (hh == 0) p.lse[(int64_t)bh * p.n + qrow] = m * kLn2 + logf(l);
    }
}

// ============================================================================================================================ dq pass
template <class TR, int D>
__global__ __launch_bounds__(256) void attn_gen_dq_kernel(GenParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    using G = Geo<D>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G::TILE];      // K tile, V tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int rb = (p.n + 127) / 128;
    const int bh = blockIdx.x / rb, blk = blockIdx.x % rb, b = bh / p.H, h = bh % p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const E* ob = static_cast<const E*>(p.out) + b * p.osb + h * p.osh;
    const E* dob = static_cast<const E*>(p.dout) + b * p.osb + h * p.osh;
    const int qrow = blk * 128 + wave * 32 + r, qld = min(qrow, p.n - 1);
    V8 qf[G::KS], dof[G::KS];
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
        qf[s] = *reinterpret_cast<const V8*>(qb + (int64_t)qld * p.sn + 16 * s + 8 * hh);
        dof[s] = *reinterpret_cast<const V8*>(dob + (int64_t)qld * p.osn + 16 * s + 8 * hh);
        const V8 of = *reinterpret_cast<const V8*>(ob + (int64_t)qld * p.osn + 16 * s + 8 * hh);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += (float)dof[s][j] * (float)of[j];
    }
    const float delta = xhalf_sum(dl);
    const float lse2 = p.lse[(int64_t)bh * p.n + qld] * kLog2e;
    if (qrow < p.n && hh == 0) p.delta[(int64_t)bh * p.n + qrow] = delta;
    f32x16 dq[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) dq[db] = f32x16{0};
    const int nt = (p.n + 63) / 64;
    u32x4 kreg[G::NCH], vreg[G::NCH];
    tile_fetch<E, D>(kreg, kb, p.sn, 0, p.n, tid);
    tile_fetch<E, D>(vreg, vb, p.sn, 0, p.n, tid);
    for (int t = 0; t < nt; ++t) {
        tile_put<D>(smem, kreg, tid);
        tile_put<D>(smem + G::TILE, vreg, tid);
        __syncthreads();
        if (t + 1 < nt) {
            tile_fetch<E, D>(kreg, kb, p.sn, (t + 1) * 64, p.n, tid);
            tile_fetch<E, D>(vreg, vb, p.sn, (t + 1) * 64, p.n, tid);
        }
        const int key0 = t * 64;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 s = {0}, dp = {0};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                s = TR::mfma32(row_frag<TR>(smem, 32 * half + r, ks, hh), qf[ks], s);
                dp = TR::mfma32(row_frag<TR>(smem + G::TILE, 32 * half + r, ks, hh), dof[ks], dp);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = key0 + 32 * half + acc_row(i, hh);
                const float pr = key < p.n ? __builtin_amdgcn_exp2f(s[i] * p.scale_log2 - lse2) : 0.f;
                s[i] = pr * (dp[i] - delta);                       // dS (the factor `scale` is applied to the finished row)
            }
#pragma unroll
            for (int g16 = 0; g16 < 2; ++g16) {
                const V8 ds = pack8<TR>(s, g16);
#pragma unroll
                for (int db = 0; db < G::DB; ++db) dq[db] = TR::mfma32(tr_frag<TR>(smem, 32 * half + 16 * g16, db, lane), ds, dq[db]);
            }
        }
        __syncthreads();
    }
    if (qrow < p.n) store_row<TR, D>(static_cast<E*>(p.dq) + b * p.gsb + (int64_t)qrow * p.gsn + h * p.gsh, dq, p.scale, hh);
}

// ============================================================================================================================ dk / dv pass
template <class TR, int D>
__global__ __launch_bounds__(256) void attn_gen_dkdv_kernel(GenParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    using G = Geo<D>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * G::TILE + 512];      // Q tile, dO tile, lse' [64], delta [64]
    float* rowc = reinterpret_cast<float*>(smem + 2 * G::TILE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int rb = (p.n + 127) / 128;
    const int bh = blockIdx.x / rb, blk = blockIdx.x % rb, b = bh / p.H, h = bh % p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const E* dob = static_cast<const E*>(p.dout) + b * p.osb + h * p.osh;
    const int krow = blk * 128 + wave * 32 + r, kld = min(krow, p.n - 1);
    V8 kf[G::KS], vf[G::KS];
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
        kf[s] = *reinterpret_cast<const V8*>(kb + (int64_t)kld * p.sn + 16 * s + 8 * hh);
        vf[s] = *reinterpret_cast<const V8*>(vb + (int64_t)kld * p.sn + 16 * s + 8 * hh);
    }
    f32x16 dk[G::DB], dv[G::DB];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) dk[db] = dv[db] = f32x16{0};
    const int nt = (p.n + 63) / 64;
    u32x4 qreg[G::NCH], dreg[G::NCH];
    float creg = 0.f;
    auto fetch_consts = [&](int t) {                     // threads 0..63: -lse log2(e) of a query row, 64..127: its delta
        if (tid < 128) {
            const int row = min(t * 64 + (tid & 63), p.n - 1);
            creg = tid < 64 ? p.lse[(int64_t)bh * p.n + row] * kLog2e : p.delta[(int64_t)bh * p.n + row];
        }
    };
    tile_fetch<E, D>(qreg, qb, p.sn, 0, p.n, tid);
    tile_fetch<E, D>(dreg, dob, p.osn, 0, p.n, tid);
    fetch_consts(0);
    for (int t = 0; t < nt; ++t) {
        tile_put<D>(smem, qreg, tid);
        tile_put<D>(smem + G::TILE, dreg, tid);
        if (tid < 128) rowc[tid] = creg;
        __syncthreads();
        if (t + 1 < nt) {
            tile_fetch<E, D>(qreg, qb, p.sn, (t + 1) * 64, p.n, tid);
            tile_fetch<E, D>(dreg, dob, p.osn, (t + 1) * 64, p.n, tid);
            fetch_consts(t + 1);
        }
        const int q0 = t * 64;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 s = {0}, dp = {0};
#pragma unroll
            for (int ks = 0; ks < G::KS; ++ks) {
                s = TR::mfma32(row_frag<TR>(smem, 32 * half + r, ks, hh), kf[ks], s);                 // S [query, key]
                dp = TR::mfma32(row_frag<TR>(smem + G::TILE, 32 * half + r, ks, hh), vf[ks], dp);      // dP [query, key]
            }
            f32x16 pr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(rowc + 32 * half + 8 * g + 4 * hh);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(rowc + 64 + 32 * half + 8 * g + 4 * hh);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * g + j, qrow = q0 + 32 * half + 8 * g + 4 * hh + j;
                    const float pv = qrow < p.n ? __builtin_amdgcn_exp2f(s[i] * p.scale_log2 - l4[j]) : 0.f;
                    pr[i] = pv;
                    s[i] = pv * (dp[i] - d4[j]);
                }
            }
#pragma unroll
            for (int g16 = 0; g16 < 2; ++g16) {
                const V8 pk = pack8<TR>(pr, g16), ds = pack8<TR>(s, g16);
#pragma unroll
                for (int db = 0; db < G::DB; ++db) {
                    dv[db] = TR::mfma32(tr_frag<TR>(smem + G::TILE, 32 * half + 16 * g16, db, lane), pk, dv[db]);
                    dk[db] = TR::mfma32(tr_frag<TR>(smem, 32 * half + 16 * g16, db, lane), ds, dk[db]);
                }
            }
        }
        __syncthreads();
    }
    if (krow < p.n) {
        store_row<TR, D>(static_cast<E*>(p.dk) + b * p.gsb + (int64_t)krow * p.gsn + h * p.gsh, dk, p.scale, hh);
        store_row<TR, D>(static_cast<E*>(p.dv) + b * p.gsb + (int64_t)krow * p.gsn + h * p.gsh, dv, 1.f, hh);
    }
}

template <class TR, int D>
int launch_fwd(const GenParams& p, hipStream_t st) {
    const int grid = p.B * p.H * ceil_div(p.n, 128);
    hipLaunchKernelGGL((attn_gen_fwd_kernel<TR, D>), dim3(grid), dim3(256), 0, st, p);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
template <class TR, int D>
int launch_bwd(const GenParams& p, int passes, hipStream_t st) {
    const int grid = p.B * p.H * ceil_div(p.n, 128);
    if (passes & 1) hipLaunchKernelGGL((attn_gen_dq_kernel<TR, D>), dim3(grid), dim3(256), 0, st, p);
    if (passes & 2) hipLaunchKernelGGL((attn_gen_dkdv_kernel<TR, D>), dim3(grid), dim3(256), 0, st, p);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

GenParams make_params(const void* q, const void* k, const void* v, int B, int n, int H, int64_t sb, int64_t sn, int64_t sh, int64_t osb,
                      int64_t osn, int64_t osh, float scale) {
    GenParams p{};
    p.q = q; p.k = k; p.v = v;
    p.B = B; p.n = n; p.H = H;
    p.sb = sb; p.sn = sn; p.sh = sh;
    p.osb = osb; p.osn = osn; p.osh = osh;
    p.scale = scale; p.scale_log2 = scale * kLog2e;
    return p;
}

}  // namespace

bool attn_gen_supported(int d) { return d == 32 || d == 64 || d == 128; }

int attn_gen_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int n, int H, int d, int64_t sb, int64_t sn,
                 int64_t sh, int64_t osb, int64_t osn, int64_t osh, float scale, int dtype, void* stream) {
    if (!attn_gen_supported(d) || (dtype != NPCD_BF16 && dtype != NPCD_F16)) return NPCD_ERR_UNSUPPORTED;
    GenParams p = make_params(q, k, v, B, n, H, sb, sn, sh, osb, osn, osh, scale);
    p.o_w = out; p.lse = lse;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NPCD_GEN_FWD(DD) (dtype == NPCD_BF16 ? launch_fwd<BF16, DD>(p, st) : launch_fwd<F16, DD>(p, st))
    return d == 32 ? NPCD_GEN_FWD(32) : d == 64 ? NPCD_GEN_FWD(64) : NPCD_GEN_FWD(128);
#undef NPCD_GEN_FWD
}

int attn_gen_bwd(int passes, const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, void* dq, void* dk,
                 void* dv, float* delta, int B, int n, int H, int d, int64_t sb, int64_t sn, int64_t sh, int64_t osb, int64_t osn, int64_t osh,
                 int64_t gsb, int64_t gsn, int64_t gsh, float scale, int dtype, void* stream) {
    if (!attn_gen_supported(d) || (dtype != NPCD_BF16 && dtype != NPCD_F16)) return NPCD_ERR_UNSUPPORTED;
    GenParams p = make_params(q, k, v, B, n, H, sb, sn, sh, osb, osn, osh, scale);
    p.out = out; p.dout = dout; p.lse = const_cast<float*>(lse); p.delta = delta;
    p.dq = dq; p.dk = dk; p.dv = dv;
    p.gsb = gsb; p.gsn = gsn; p.gsh = gsh;
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NPCD_GEN_BWD(DD) (dtype == NPCD_BF16 ? launch_bwd<BF16, DD>(p, passes, st) : launch_bwd<F16, DD>(p, passes, st))
    return d == 32 ? NPCD_GEN_BWD(32) : d == 64 ? NPCD_GEN_BWD(64) : NPCD_GEN_BWD(128);
#undef NPCD_GEN_BWD
}

}  // namespace npcd
