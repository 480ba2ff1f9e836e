// Fused PointNeRF shading, kernel A in its "rows" form: the four non-linear per-pair layers of the aggregator MLP and the
// inverse-distance aggregation (aggregators/mlp.py:62-125 of the reference) with the ACTIVATIONS IN REGISTERS.
//
// shade.hip's first form keeps a 128-row activation tile in LDS: every layer is LDS reads -> MFMA -> barrier -> LDS writes ->
// barrier, and the matrix pipe is busy 0.375 of the time.  Here a wave owns 64 (point, neighbour) rows from the
// gather to the aggregation:
//   * its activations are the B operand of H_out^T[out][row] = W[out][in] . H_in^T[in][row], held as 16 k-steps x 2 row blocks of
//     8 fp16 per lane (128 registers).  A 32x32 accumulator tile gives lane (row, g) the outputs 8 j + 4 g + b of its row; a B
//     operand wants 8 k-slots per lane.  The contraction order is free, so k-slot (s, g, e) of a hidden layer is DEFINED as
//     feature 32 (s/2) + 8 (2 (s%2) + e/4) + 4 g + e%4: then accumulator values 8 jp .. 8 jp + 7 of output block mb, after
//     LeakyReLU and conversion, ARE fragment s = 2 mb + jp of the next layer -- no LDS, no cross-lane traffic, no barrier.  The
//     weights are packed with the same permutation of their input columns (npcd_shade_pack_weights);
//   * the last non-linear layer runs with the operands swapped (rows on M, features on N): its accumulators hold 16 ROWS per lane
//     for one feature, which is the k-layout of one more matrix product, G^T[feature][point] = H^T[feature][row] . A[row][point]
//     with A = the normalised inverse-distance weights of the window's <= 64 points: the aggregation is 32 MFMAs;
//   * the weights (432 KB per tile of 256 rows for 32 feature channels) stream through a 64-KB LDS ring as 4-KB slabs (two k-steps
//     of two output blocks) by LDS-DMA, 12 slabs ahead, one counted vmcnt wait + one barrier per four slabs; all four waves of the
//     workgroup read every slab (ds_read_b128 of contiguous 1-KB fragments: conflict-free), each fragment feeds two MFMAs;
//   * one wave per SIMD (the register file of a SIMD belongs to one wave: 128 + 128 activation registers, 64 + 64 accumulators):
//     the epilogue of a quarter layer (two output blocks) is spread over the k-steps of the next quarter so that its ~2.5 vector
//     instructions per matrix instruction hide behind the matrix pipe.
// Rows: the valid (point, neighbour) pairs are numbered by a prefix sum over the points (two small kernels); window w takes the
// points whose first row lies in [stride w, stride (w + 1)), stride = 65 - k: at most 64 rows, whole points only.
#include <type_traits>

#include "common.h"
#include "shade_common.h"

namespace npcd {
namespace {

constexpr int kSlab = 4096;                     // 2 k-steps x 2 output blocks x 1 KiB
constexpr int kRingSlabs = 16;
#ifndef NPCD_ROWS_SYNC_STEPS
#define NPCD_ROWS_SYNC_STEPS 4
#endif
constexpr int kSyncSteps = NPCD_ROWS_SYNC_STEPS;   // steps (slabs) per counted wait + barrier: 2 or 4 (every layer is a multiple of 4 steps long)
constexpr int kAhead = 12;                      // slabs in flight ahead of the one being read; <= kRingSlabs - 3
constexpr int kRingBytes = kRingSlabs * kSlab;
constexpr int kBiasBytes = 4 * kHidden * 4;
constexpr int kWaveScratch = 3584;              // per-wave row / point tables (below)
constexpr int kScanPts = 1024;                  // points per workgroup of the two index kernels
constexpr int kRowsLds = kRingBytes + kBiasBytes + 4 * kWaveScratch + 4 * 32 * kRowBytes;

struct RowsWs {
    int32_t* hdr;      // [4]  J = points with at least one neighbour, Q = rows
    int32_t* e_point;  // [max_points]      j -> point
    int32_t* e_row;    // [max_points + 1]  j -> first row; e_row[J] = Q
    int32_t* win;      // [windows + 1]     w -> first j of the window
    int32_t* blk;      // [2 nblk]          per index-kernel workgroup: (non-empty points, rows)
};

__device__ __forceinline__ uint32_t lds_addr32(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void wave_sync() {       // LDS tables written by some lanes of a wave, read by others
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int count_valid(const int32_t* nb, int k) {
    int c = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < k) c += nb[t] >= 0 ? 1 : 0;
    return c;
}
__device__ __forceinline__ int wave_add(int x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// ---- index kernels --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_count_kernel(ShadeArgs a, RowsWs w) {
    __shared__ int red[2][4];
    const int tid = threadIdx.x, P = min(*a.n_points, a.max_points);
    int ne = 0, rows = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = blockIdx.x * kScanPts + tid * 4 + i;
        if (p < P) {
            const int c = count_valid(a.nb_idx + (int64_t)p * a.k, a.k);
            ne += c > 0;
            rows += c;
        }
    }
    ne = wave_add(ne);
    rows = wave_add(rows);
    if ((tid & 63) == 0) { red[0][tid >> 6] = ne; red[1][tid >> 6] = rows; }
    __syncthreads();
    if (tid == 0) {
        w.blk[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        w.blk[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
}

__global__ __launch_bounds__(256) void rows_index_kernel(ShadeArgs a, RowsWs w, int stride) {
    __shared__ int red[2][4], wsum[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, P = min(*a.n_points, a.max_points);
    int one = 0, orow = 0;
    for (int i = tid; i < (int)blockIdx.x; i += 256) { one += w.blk[2 * i]; orow += w.blk[2 * i + 1]; }
    one = wave_add(one);
    orow = wave_add(orow);
    if (lane == 0) { red[0][wave] = one; red[1][wave] = orow; }
    int c[4], ne = 0, rows = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = blockIdx.x * kScanPts + tid * 4 + i;
        c[i] = p < P ? count_valid(a.nb_idx + (int64_t)p * a.k, a.k) : 0;
        ne += c[i] > 0;
        rows += c[i];
    }
    int ine = ne, irows = rows;              // inclusive scan over the lanes of the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int x = __shfl_up(ine, o, 64), y = __shfl_up(irows, o, 64);
        if (lane >= o) { ine += x; irows += y; }
    }
    if (lane == 63) { wsum[0][wave] = ine; wsum[1][wave] = irows; }
    __syncthreads();
    int j = red[0][0] + red[0][1] + red[0][2] + red[0][3] + ine - ne;
    int row = red[1][0] + red[1][1] + red[1][2] + red[1][3] + irows - rows;
    for (int v = 0; v < wave; ++v) { j += wsum[0][v]; row += wsum[1][v]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int p = blockIdx.x * kScanPts + tid * 4 + i;
        if (p >= P) continue;
        if (c[i] > 0) {
            w.e_point[j] = p;
            w.e_row[j] = row;
            const int nrow = row + c[i];
            if (row / stride != nrow / stride) w.win[nrow / stride] = j + 1;   // the next point is the first one of that window
            ++j;
            row = nrow;
        } else {                                  // a point without neighbours aggregates to zero
            u32x4* gp = reinterpret_cast<u32x4*>(a.G + (int64_t)p * kHidden);
            const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int q = 0; q < kHidden * 2 / 16; ++q) gp[q] = z;
        }
    }
    if (blockIdx.x == 0 && tid == 0) w.win[0] = 0;
    if (blockIdx.x == gridDim.x - 1 && tid == 255) {      // j, row are the totals now
        w.hdr[0] = j;
        w.hdr[1] = row;
        w.e_row[j] = row;
        w.win[(row + stride - 1) / stride] = j;
    }
}

// ---- the rows kernel ------------------------------------------------------------------------------------------------------
struct Ring {
    const unsigned char* src;   // the slab stream (period: one tile)
    uint32_t lds;               // LDS byte address of this wave's quarter of slab 0
    uint32_t lane_off;          // byte offset of this lane inside a slab
    int issue, islot, rslot;    // next slab of the stream to fetch, ring slot it goes to, ring slot to read next
};

// values 8 jp .. 8 jp + 7 of an accumulator tile -> LeakyReLU -> one operand fragment of the next product
// (2 conversions, then LeakyReLU on the packed pair: max(h, 0.01 h) in fp16 -- 1.5 vector instructions per value instead of 2.5;
// the negative side is rounded twice, to fp16 and after the scaling)
__device__ __forceinline__ f16x8 epi_frag(const f32x16& acc, int jp) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    u32x4 r;
#ifdef NPCD_DIAG_NO_EPI              // DIAGNOSTIC builds only (wrong results): no conversion / activation work
    for (int e = 0; e < 4; ++e) r[e] = __float_as_uint(acc[8 * jp + e]);
    return __builtin_bit_cast(f16x8, r);
#endif
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const f32x2 f = {acc[8 * jp + 2 * e], acc[8 * jp + 2 * e + 1]};
        const f16x2 h = __builtin_convertvector(f, f16x2);
        const f16x2 sc = {(_Float16)kLeaky, (_Float16)kLeaky};
        r[e] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(h, h * sc));
    }
    return __builtin_bit_cast(f16x8, r);
}

// two accumulator values -> fp16 pair -> LeakyReLU
__device__ __forceinline__ uint32_t epi_pair(float x0, float x1) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#ifdef NPCD_DIAG_NO_EPI              // DIAGNOSTIC builds only (wrong results): no conversion / activation work
    return __float_as_uint(x0);
#endif
    const f32x2 f = {x0, x1};
    const f16x2 h = __builtin_convertvector(f, f16x2);
    const f16x2 sc = {(_Float16)kLeaky, (_Float16)kLeaky};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(h, h * sc));
}
__device__ __forceinline__ void set_dword(f16x8& frag, int e, uint32_t v) {
    u32x4 t = __builtin_bit_cast(u32x4, frag);
    t[e] = v;
    frag = __builtin_bit_cast(f16x8, t);
}

// first fragment (of 8) of a quarter's epilogue that step i of the following quarter converts: one per step, or (`early`: a hidden
// layer's first quarter reads the last fragments of the layer below in its steps 6 and 7) everything within six steps
__host__ __device__ constexpr int epi_first(int i, int nsteps, bool early) {
    return nsteps < 6 ? (i * 8 + nsteps - 1) / nsteps : !early ? (i * 8) / nsteps : (i < 2 ? 2 * i : (i + 2 < 8 ? i + 2 : 8));
}

#ifdef NPCD_SHADE_TL
__device__ long long g_rows_tl[16];
#define NPCD_RTL(i) do { if (blockIdx.x == NPCD_SHADE_TL && wave == 0 && tile == (int)blockIdx.x + 2 * (int)gridDim.x) tl[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NPCD_RTL(i) do { } while (0)
#endif

template <int FEAT>
__global__ __launch_bounds__(256, 1) void shade_rows_kernel(ShadeArgs a, RowsWs ws, int stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    constexpr int K0 = FEAT + kEncBlock, KS0 = K0 / 16, KSF = FEAT / 16;
    constexpr int kPeriod = 4 * (KS0 / 2 + 3 * 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n = lane & 31, g = lane >> 5;
    unsigned char* ring = dsmem;
    float* biasL = reinterpret_cast<float*>(dsmem + kRingBytes);
    unsigned char* scr = dsmem + kRingBytes + kBiasBytes + wave * kWaveScratch;
    int* rowgi = reinterpret_cast<int*>(scr);            // [64] row -> neighbour (global index), -1 past the window's rows
    int* rowpt = rowgi + 64;                             // [64] row -> local point
    float* relv = reinterpret_cast<float*>(rowpt + 64);  // [64][4] relative position
    float* wrow = relv + 256;                            // [64] inverse distance
    float* wn = wrow + 64;                               // [64] normalised weight
    int* ptp = reinterpret_cast<int*>(wn + 64);          // [64] local point -> point
    int* prs = ptp + 64;                                 // [64] its first row in the window
    int* pcn = prs + 64;                                 // [64] its number of rows
    float* pinv = reinterpret_cast<float*>(pcn + 64);    // [64] 1 / sum of its rows' inverse distances
    unsigned char* gst = dsmem + kRingBytes + kBiasBytes + 4 * kWaveScratch + wave * (32 * kRowBytes);   // [32 points][512 + 16 B] aggregated features on their way out
    const ShadeLayout L = shade_layout(FEAT);
    const int Q = ws.hdr[1];
    const int nwin = (Q + stride - 1) / stride, ntiles = (nwin + 3) >> 2;
    if ((int)blockIdx.x >= ntiles) return;
#pragma unroll
    for (int l = 0; l < 4; ++l) biasL[l * kHidden + tid] = reinterpret_cast<const float*>(a.wpack + L.bias[l])[tid];
    __syncthreads();

    Ring rg;
    rg.src = a.wpack + L.rows;
    rg.lds = lds_addr32(ring) + wave * 1024;
    rg.lane_off = wave * 1024 + lane * 16;
    rg.issue = 0; rg.islot = 0; rg.rslot = 0;
    auto issue_slab = [&]() __attribute__((always_inline)) {
        const void* sb = rg.src + (int64_t)rg.issue * kSlab;
        const uint64_t v = reinterpret_cast<uint64_t>(sb);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        const void* sbu = reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
        const uint32_t dst = __builtin_amdgcn_readfirstlane(rg.lds + rg.islot * kSlab);
#ifndef NPCD_DIAG_NO_DMA             // DIAGNOSTIC builds only (wrong results): the stream without its loads
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(rg.lane_off), "s"(sbu), "s"(dst) : "memory");
#endif
        rg.issue = rg.issue + 1 == kPeriod ? 0 : rg.issue + 1;
        rg.islot = (rg.islot + 1) & (kRingSlabs - 1);
    };
    const uint32_t ring_rd = lds_addr32(ring) + lane * 16;
    // fragments of the next slab: hand-issued LDS reads (the compiler would sink them to the end of the step, and the wait that
    // follows the next barrier would then cover their whole latency); `land` is their wait, at the end of the step
    // (the current fragments pass through the statement so that the matrix instructions that use them follow the reads)
    auto read_slab2 = [&](f16x8 (&A)[4], f16x8 (&Cur)[4]) __attribute__((always_inline)) {
        const uint32_t ad = ring_rd + rg.rslot * kSlab;
#ifdef NPCD_DIAG_NO_LDSREAD          // DIAGNOSTIC builds only (wrong results): the fragments are not re-read
        asm volatile("" : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3]), "+v"(Cur[0]), "+v"(Cur[1]), "+v"(Cur[2]), "+v"(Cur[3]) : "v"(ad) : "memory");
        return;
#endif
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072"
                     : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3]), "+v"(Cur[0]), "+v"(Cur[1]), "+v"(Cur[2]), "+v"(Cur[3]) : "v"(ad) : "memory");
        rg.rslot = (rg.rslot + 1) & (kRingSlabs - 1);
    };
    auto read_slab = [&](f16x8 (&A)[4]) __attribute__((always_inline)) {
        const uint32_t ad = ring_rd + rg.rslot * kSlab;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                     : "=&v"(A[0]), "=&v"(A[1]), "=&v"(A[2]), "=&v"(A[3]) : "v"(ad) : "memory");
        rg.rslot = (rg.rslot + 1) & (kRingSlabs - 1);
    };
    // (the step's accumulators are operands so that its matrix instructions stay between the reads and this wait)
    auto land = [&](f16x8 (&A)[4], f32x16 (&acc)[2][2]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3]), "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1])::"memory");
    };
    auto land0 = [&](f16x8 (&A)[4]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0]), "+v"(A[1]), "+v"(A[2]), "+v"(A[3])::"memory");
    };
    // one step of the stream.  Every kSyncSteps-th step: the next kSyncSteps slabs have landed for every wave (counted wait on this
    // wave's own DMA + barrier) and nobody reads the slabs that are overwritten; the other steps only read.
    auto advance = [&](f16x8 (&An)[4], f16x8 (&Cur)[4], bool sync) __attribute__((always_inline)) {
        if (sync) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kAhead - kSyncSteps) : "memory");
#ifndef NPCD_DIAG_NO_BARRIER         // DIAGNOSTIC builds only (races)
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < kSyncSteps; ++u) issue_slab();
        }
        read_slab2(An, Cur);
    };
#pragma unroll 1
    for (int i = 0; i <= kAhead; ++i) issue_slab();
    f16x8 Ac[4];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kAhead) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_slab(Ac);
    land0(Ac);

#ifdef NPCD_SHADE_TL
    long long tl[16];
    for (int i = 0; i < 16; ++i) tl[i] = 0;
#endif
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        NPCD_RTL(0);
        // ---- the window of this wave: points, rows, weights ---------------------------------------------------------
        const int w = tile * 4 + wave;
        int j_lo = 0, npts = 0, row0 = 0, nrows = 0;
        if (w < nwin) {
            j_lo = ws.win[w];
            const int j_hi = ws.win[w + 1];
            row0 = ws.e_row[j_lo];
            npts = min(j_hi - j_lo, 64);
            nrows = min(ws.e_row[j_hi] - row0, 64);
        }
        {
            int p = -1, rs = 0, cnt = 0;
            if (lane < npts) {
                p = ws.e_point[j_lo + lane];
                rs = ws.e_row[j_lo + lane] - row0;
                cnt = ws.e_row[j_lo + lane + 1] - row0 - rs;
                cnt = max(0, min(cnt, 64 - rs));
            }
            ptp[lane] = p; prs[lane] = rs; pcn[lane] = cnt;
            rowgi[lane] = -1;
            wave_sync();
            if (p >= 0) {
                int gi[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) gi[t] = t < a.k ? a.nb_idx[(int64_t)p * a.k + t] : -1;
                int rank = 0;
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (gi[t] >= 0 && rank < cnt) { rowgi[rs + rank] = gi[t]; rowpt[rs + rank] = lane; ++rank; }
            }
            wave_sync();
            const bool rv = lane < nrows && rowgi[lane] >= 0;
            float rel[3] = {0.f, 0.f, 0.f}, wgt = 0.f;
            int mypt = 0;
            if (rv) {
                const int mygi = rowgi[lane];
                mypt = rowpt[lane];
                const int pp = ptp[mypt];
#pragma unroll
                for (int c = 0; c < 3; ++c) rel[c] = a.pts[(int64_t)pp * 3 + c] - a.kp_pos[(int64_t)mygi * 3 + c];
                wgt = 1.f / (sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]) + 1e-5f);
            }
            *reinterpret_cast<f32x4*>(relv + 4 * lane) = f32x4{rel[0], rel[1], rel[2], 0.f};
            wrow[lane] = wgt;
            wave_sync();
            float inv = 0.f;
            if (lane < npts) {
                float s = 0.f;
                for (int i = 0; i < cnt; ++i) s += wrow[rs + i];
                inv = s > 0.f ? 1.f / s : 0.f;
            }
            pinv[lane] = inv;
            wave_sync();
            wn[lane] = rv ? wgt * pinv[mypt] : 0.f;
            wave_sync();
        }
        NPCD_RTL(1);
        // ---- layer-0 operand: [features | 30 x (sin on g = 0, cos on g = 1) | x, y on g = 0; z, 0 on g = 1] -----------
        f16x8 B0[KS0][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int r = 32 * nb + n, gi = rowgi[r];
            const f32x4 rl = *reinterpret_cast<const f32x4*>(relv + 4 * r);
            const float rr[3] = {rl[0], rl[1], rl[2]};
#pragma unroll
            for (int s = 0; s < KSF; ++s) {
                f16x8 v;
                if (gi >= 0) {
                    const float* fp = a.kp_feat + (int64_t)gi * FEAT + 16 * s + 8 * g;
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(fp), x1 = *reinterpret_cast<const f32x4*>(fp + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v[e] = (_Float16)x0[e]; v[4 + e] = (_Float16)x1[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
                }
                B0[s][nb] = v;
            }
            const float phase = g ? 0.25f : 0.f;            // cos(2 pi u) = sin(2 pi (u + 1/4)); v_sin takes revolutions
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                f16x8 v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int t = 8 * s2 + e;
                    float x;
                    if (t < 30) {
                        const int c = t / 10, i = t % 10;
                        const float u = rr[c] * (0.5f * (float)(1 << i));   // sin(x 2^i pi) = sin(2 pi u), u = x 2^(i-1) (exact)
                        x = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(__builtin_amdgcn_fractf(u) + phase));
                    } else if (t == 30) {
                        x = g ? rr[2] : rr[0];
                    } else {
                        x = g ? 0.f : rr[1];
                    }
                    v[e] = (_Float16)(gi >= 0 ? x : 0.f);
                }
                B0[KSF + s2][nb] = v;
            }
        }

        NPCD_RTL(2);
        // ---- the four layers ----------------------------------------------------------------------------------------
        f16x8 X[16][2], Y[16][2];
        f32x16 pend[2][2];
        // one layer: 4 quarters (two output blocks each) x KS / 2 steps (two k-steps each); `pend` = the accumulators of the
        // previous quarter, converted into DST while this quarter's matrix instructions run
        auto layer = [&](auto ks_tag, auto layer_tag, auto final_tag, auto haspend_tag, const auto& Bin, auto& BinW, auto& Bout) __attribute__((always_inline)) {
            constexpr int KS = decltype(ks_tag)::value, LAYER = decltype(layer_tag)::value;
            constexpr bool FINAL = decltype(final_tag)::value, HASPEND = decltype(haspend_tag)::value;
            constexpr int NST = KS / 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x16 acc[2][2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    f32x16 init;
                    if (FINAL) {
                        const float b = biasL[LAYER * kHidden + 32 * (2 * q + m) + n];
#pragma unroll
                        for (int v = 0; v < 16; ++v) init[v] = b;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x4 b4 = *reinterpret_cast<const f32x4*>(biasL + LAYER * kHidden + 32 * (2 * q + m) + 8 * j + 4 * g);
#pragma unroll
                            for (int b = 0; b < 4; ++b) init[4 * j + b] = b4[b];
                        }
                    }
                    acc[m][0] = init;
                    acc[m][1] = init;
                }
                static_assert((4 * (KS / 2)) % kSyncSteps == 0, "a layer is a whole number of synchronisation periods of the slab stream");
#pragma unroll
                for (int i = 0; i < NST; ++i) {
                    f16x8 An[4];
                    advance(An, Ac, ((q * NST + i) % kSyncSteps) == 0);
                    // a single wave feeds the SIMD: vector instructions hide behind a matrix instruction only if they sit in its
                    // own 32-cycle slot (<= 5 of them).  The step is therefore written as 8 x (1 matrix instruction, its share of
                    // the pending epilogue: pairs of values -> convert, scale, max), with nothing allowed across the slots.
                    const bool has_epi = q > 0 || HASPEND;
                    const int pq = q > 0 ? q - 1 : 3;
                    const int f0 = has_epi ? epi_first(i, NST, q == 0) : 0, nfr = has_epi ? epi_first(i + 1, NST, q == 0) - f0 : 0;
                    u32x4 fr[3];                                   // the fragments this step converts (whole ones)
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int sl = t >> 2, m = (t >> 1) & 1, nb = t & 1;
                        acc[m][nb] = FINAL ? F16::mfma32(Bin[2 * i + sl][nb], Ac[2 * sl + m], acc[m][nb])
                                           : F16::mfma32(Ac[2 * sl + m], Bin[2 * i + sl][nb], acc[m][nb]);
#pragma unroll
                        for (int pr = (t * 4 * nfr) / 8; pr < ((t + 1) * 4 * nfr) / 8; ++pr) {
                            const int f = f0 + (pr >> 2), e = pr & 3, fm = f >> 2, jp = (f >> 1) & 1, fnb = f & 1;
                            fr[pr >> 2][e] = epi_pair(pend[fm][fnb][8 * jp + 2 * e], pend[fm][fnb][8 * jp + 2 * e + 1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int u = 0; u < nfr; ++u) {
                        const int f = f0 + u, fm = f >> 2, jp = (f >> 1) & 1, fnb = f & 1;
                        const f16x8 v = __builtin_bit_cast(f16x8, fr[u]);
                        if (q == 0) BinW[4 * pq + 2 * fm + jp][fnb] = v;                       // the layer below ends here
                        else if (FINAL) Bout[2 * (2 * pq + fm) + fnb][jp] = v;                 // [feature block][row k-step]
                        else Bout[4 * pq + 2 * fm + jp][fnb] = v;
                    }
                    land(An, acc);
#pragma unroll
                    for (int f = 0; f < 4; ++f) Ac[f] = An[f];
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) pend[m][nb] = acc[m][nb];
            }
        };
        using std::integral_constant;
        f16x8 dummy[16][2];
        layer(integral_constant<int, KS0>{}, integral_constant<int, 0>{}, integral_constant<bool, false>{}, integral_constant<bool, false>{}, B0, dummy, Y);
        NPCD_RTL(3);
        layer(integral_constant<int, 16>{}, integral_constant<int, 1>{}, integral_constant<bool, false>{}, integral_constant<bool, true>{}, Y, Y, X);
        NPCD_RTL(4);
        layer(integral_constant<int, 16>{}, integral_constant<int, 2>{}, integral_constant<bool, false>{}, integral_constant<bool, true>{}, X, X, Y);
        NPCD_RTL(5);
        layer(integral_constant<int, 16>{}, integral_constant<int, 3>{}, integral_constant<bool, true>{}, integral_constant<bool, true>{}, Y, Y, X);
        NPCD_RTL(6);
#pragma unroll
        for (int f = 0; f < 8; ++f) {                          // the last quarter's accumulators
            const int m = f >> 2, jp = (f >> 1) & 1, nb = f & 1;
            X[2 * (2 * 3 + m) + nb][jp] = epi_frag(pend[m][nb], jp);
        }

        NPCD_RTL(7);
        // ---- aggregation: G^T[feature][point] = H^T[feature][row] . A[row][point] -----------------------------------
        // X[2 mb + (s >> 1)][s & 1] = rows-k-step s of feature block mb; k-slot (s, g, e) = row 32 (s/2) + 8 (2 (s%2) + e/4) + 4 g + e%4
#pragma unroll 1
        for (int pm = 0; pm * 32 < npts; ++pm) {
            const int lp = 32 * pm + n;                              // this lane's point of the window
            const int rs = prs[lp], re = rs + pcn[lp];
            f16x8 Ag[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int rho = 32 * (s >> 1) + 8 * (2 * (s & 1) + (e >> 2)) + 4 * g + (e & 3);
                    const float wv = wn[rho];
                    Ag[s][e] = (_Float16)((rho >= rs && rho < re) ? wv : 0.f);
                }
            NPCD_RTL(9);
            wave_sync();                                             // the previous pass has left the staging rows
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) {
                f32x16 ga;
#pragma unroll
                for (int v = 0; v < 16; ++v) ga[v] = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) ga = F16::mfma32(X[2 * mb + (s >> 1)][s & 1], Ag[s], ga);
#pragma unroll
                for (int j = 0; j < 4; ++j) {                        // lane (point n, g): features 32 mb + 8 j + 4 g ..
                    f16x4 o;
#pragma unroll
                    for (int b = 0; b < 4; ++b) o[b] = (_Float16)ga[4 * j + b];
                    *reinterpret_cast<f16x4*>(gst + n * kRowBytes + (32 * mb + 8 * j + 4 * g) * 2) = o;
                }
            }
            NPCD_RTL(10);
            wave_sync();
            // whole 512-byte rows: one point per half wave
            int pp[16];
            u32x4 rowv[16];
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                pp[it] = ptp[32 * pm + 2 * it + g];
                rowv[it] = *reinterpret_cast<const u32x4*>(gst + (2 * it + g) * kRowBytes + 16 * n);
            }
            uint32_t bad = 0;        // range guard (shade_common.h): an fp16 exponent field of all ones in any stored aggregated value
#pragma unroll
            for (int it = 0; it < 16; ++it)
                if (32 * pm + 2 * it + g < npts && pp[it] >= 0) {
                    *reinterpret_cast<u32x4*>(a.G + (int64_t)pp[it] * kHidden + 8 * n) = rowv[it];
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) bad |= (rowv[it][c4] & 0x7c007c00u) + 0x04000400u;     // 0x7c00 + 0x0400 sets bit 15
                }
            if (a.status && (bad & 0x80008000u)) atomicOr(a.status, kShadeNonfinitePairs);
        }
        NPCD_RTL(8);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef NPCD_SHADE_TL
    if (blockIdx.x == NPCD_SHADE_TL && wave == 0 && lane == 0)
        for (int i = 0; i < 16; ++i) g_rows_tl[i] = tl[i];
#endif
}

}  // namespace

#ifdef NPCD_SHADE_TL
int rows_debug_read(long long* out, int count) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rows_tl), sizeof(long long) * count); }
#endif

int64_t shade_rows_workspace_bytes(int max_points) {
    const int64_t nblk = (max_points + kScanPts - 1) / kScanPts, nwin = ((int64_t)max_points * 8) / 57 + 2;
    return (16 + 4 * (int64_t)max_points + 4 * ((int64_t)max_points + 1) + 4 * (nwin + 1) + 8 * nblk + 255) / 256 * 256;
}

int shade_rows_launch(const ShadeArgs& a, void* rows_ws, hipStream_t st) {
    const int mp = a.max_points;
    const int nblk = (mp + kScanPts - 1) / kScanPts;
    const int stride = 65 - a.k;
    const int64_t nwin = ((int64_t)mp * a.k + stride - 1) / stride + 1;
    RowsWs w;
    w.hdr = static_cast<int32_t*>(rows_ws);
    w.e_point = w.hdr + 4;
    w.e_row = w.e_point + mp;
    w.win = w.e_row + mp + 1;
    w.blk = w.win + ((int64_t)mp * 8) / 57 + 3;
    static DynLds lds32, lds128;
    NPCD_HIP_CHECK(lds32.ensure(reinterpret_cast<const void*>(shade_rows_kernel<32>), kRowsLds));
    NPCD_HIP_CHECK(lds128.ensure(reinterpret_cast<const void*>(shade_rows_kernel<128>), kRowsLds));
    hipLaunchKernelGGL(rows_count_kernel, dim3(nblk), dim3(256), 0, st, a, w);
    hipLaunchKernelGGL(rows_index_kernel, dim3(nblk), dim3(256), 0, st, a, w, stride);
    const int64_t tiles = (nwin + 3) / 4;
    const int grid = (int)(tiles < 256 ? tiles : 256);
    if (a.feat_dim == 32) hipLaunchKernelGGL(shade_rows_kernel<32>, dim3(grid), dim3(256), kRowsLds, st, a, w, stride);
    else hipLaunchKernelGGL(shade_rows_kernel<128>, dim3(grid), dim3(256), kRowsLds, st, a, w, stride);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

}  // namespace npcd
