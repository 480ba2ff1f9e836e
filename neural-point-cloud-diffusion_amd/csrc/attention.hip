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
//   fwd  : one wave = 32 query rows, workgroup = 4 waves = 128 rows; 64-key K / V tiles stream through a 3-slot LDS ring
//          (LDS-DMA) shared by the 4 waves.
//   dq   : same decomposition; per key half S^T, dP^T = V dO^T, dQ^T += K^T dS^T; also writes the per-row constants of the
//          dK/dV pass (-lse/scale, -rowsum(dO * O)).
//   dkdv : one wave = 32 keys (on the lanes), workgroup = 128 keys; Q / dO tiles of 64 rows stream through the ring;
//          S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.  No atomics: dq and dk/dv come from two passes that each
//          own their outputs (deterministic).
// All three run the same software pipeline (a stage = one 32-row half tile; the products consuming the previous half's P / dS
// are issued behind the next half's score products) and the same ring protocol (see kv_mid); DESIGN.md 5.1 has the details
// and the measurements.
#include <math.h>

#include <type_traits>

#include "attention_gen.h"
#include "common.h"

namespace npcd {

#if defined(NPCD_TIMELINE64) && !defined(NPCD_TIMELINE)
#define NPCD_TIMELINE NPCD_TIMELINE64      // (the stamps of the 64-row forward: same buffer, same read-back entry point)
#endif
#ifdef NPCD_TIMELINE
// diagnostic build only: s_memtime stamps of one wave (block NPCD_TIMELINE, wave 0), read back with npcd_debug_read
__device__ long long g_timeline[80];
#define NPCD_TS(i)                                                                       \
    do {                                                                                 \
        if (tl_on) tl[(i)] = __builtin_amdgcn_s_memtime();                               \
    } while (0)
#else
#define NPCD_TS(i) do { } while (0)
#endif

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
    // optional by-product of the backward (npcd_attn_bwd_colsum): per-wave column sums of the stored dq / dk / dv rows, one fp32 row of
    // 3 H 64 columns (the packed c_qkv order: h, {q, k, v}, d) per (batch, 128-row block, wave) [+ one per batch for the edge token]
    float* colsum;
    // optional scratch of the forward (npcd_attn_fwd_ws): partial softmax states of the LAST query row of a sequence of 256 j + 1
    // tokens, one record of kRowxFloats floats per (batch, head, 64-key tile); see rowx_tile
    float* rowx;
};

// Rows past the end of a sequence are CLAMPED to the last valid row instead of being predicated (the duplicated rows are
// neutralised downstream: masked scores / P = 0 / outputs never stored).
// ---- LDS-DMA staging ----------------------------------------------------------------------------
// Two 64x64 tiles (A at buf, B at buf+8192) are filled by 16 global_load_lds_dwordx4 pieces of 1 KiB
// (8 rows x 128 B); wave w issues 4 of them (waves 0,1: tile A, waves 2,3: tile B).  The LDS destination of
// a piece is linear (wave-uniform base + lane*16), so the XOR swizzle is applied on the SOURCE side: the
// lane that fills (row, chunk') fetches logical chunk chunk' ^ tile_swz(row).  No VGPRs, no ds_write, and
// the loads stay in flight across barriers (counted s_waitcnt vmcnt + raw s_barrier, never __syncthreads).
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// One LDS-DMA wave-instruction in the saddr + voffset form: wave-uniform 64-bit base in scalar registers, one 32-bit
// byte offset per lane, LDS destination (wave-uniform byte address; lane l lands at +16 l / +4 l) through M0.
// Hand-issued so that a tile costs two address VGPRs in total instead of a 64-bit pointer per piece.
__device__ __forceinline__ uint32_t lds_addr(const unsigned char* p);
__device__ __forceinline__ const void* uniform_ptr(const void* p) {     // pin a wave-uniform pointer into scalar registers
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ void dma16_issue(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    sbase = uniform_ptr(sbase);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4_issue(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    sbase = uniform_ptr(sbase);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <class E>
__device__ __forceinline__ void dma_tile_pair(unsigned char* buf, const E* a_base, int64_t a_stride, const E* b_base, int64_t b_stride,
                                              int row0, int nrows, int wave, int lane) {
    const bool second = wave >= 2;          // wave-uniform
    const E* base = second ? b_base : a_base;
    const int64_t stride = second ? b_stride : a_stride;
    const uint32_t tile = lds_addr(buf) + (second ? 8192 : 0);
    const int w2 = wave & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int prow = (w2 * 4 + i) * 8 + (lane >> 3);              // row inside the tile
        const int grow = min(row0 + prow, nrows - 1);                  // clamp: duplicated rows are neutralised downstream
        const uint32_t voff = (uint32_t)((grow * stride + (((lane & 7) ^ tile_swz(prow)) << 3)) * (int64_t)sizeof(E));
        dma16_issue(base, voff, __builtin_amdgcn_readfirstlane(tile + (w2 * 4 + i) * 1024));
    }
}
// barrier that does NOT drain the vector-memory counter (keeps later tiles' DMA in flight)
#define NPCD_DMA_WAIT_BARRIER(N)                                   \
    do {                                                           \
        asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");      \
        __builtin_amdgcn_s_barrier();                              \
        asm volatile("" ::: "memory");                             \
    } while (0)

// Fast form for tiles that lie completely inside the sequence: one wave-uniform base per tile (scalar
// registers) + a loop-invariant 32-bit lane offset, i.e. the saddr + voffset form of global_load_lds.  The
// XOR swizzle of a piece only depends on the piece's parity (tile_swz uses row bits 1..3), so two lane offsets
// serve all pieces.  q, k and v are views of one c_qkv output and share their strides.
struct DmaLane {
    uint32_t off[2];
};
template <class E>
__device__ __forceinline__ DmaLane dma_lane(int64_t stride, int lane) {
    DmaLane d;
    const int rowin = lane >> 3;
#pragma unroll
    for (int par = 0; par < 2; ++par)
        d.off[par] = (uint32_t)((rowin * stride + (((lane & 7) ^ tile_swz(8 * par + rowin)) << 3)) * (int64_t)sizeof(E));
    return d;
}
template <class E, int SLOT>
__device__ __forceinline__ void dma_tile_pair_fast(unsigned char* smem, const E* a_base, const E* b_base, int64_t stride, int row0,
                                                   int wave /* wave-uniform (scalar) */, const DmaLane& dl) {
    const bool second = wave >= 2;
    const int w2 = wave & 1;
    const char* sbase = reinterpret_cast<const char*>((second ? b_base : a_base) + (int64_t)(row0 + w2 * 32) * stride);
    const uint32_t dst0 = lds_addr(smem) + SLOT * 16384 + (second ? 8192 : 0) + w2 * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        dma16_issue(sbase + (int64_t)i * 8 * stride * (int64_t)sizeof(E), dl.off[i & 1], __builtin_amdgcn_readfirstlane(dst0 + i * 1024));
}

// Per-lane LDS byte addresses of the MFMA fragments of a 64x64 tile at offset 0 of the ring, computed once per
// kernel.  Everything else (ring slot, K or V image, 32-key half, 16-row group) is a compile-time constant
// that goes into the instructions' 16-bit offset field: the swizzle only uses row bits 1..3.
struct FragAddr {
    uint32_t row[4];    // ds_read_b128: row (lane&31), logical chunk 2s + (lane>>5)
    uint32_t tr[2][2];  // ds_read_b64_tr_b16: [db][second read, 8 rows further]
};
__device__ __forceinline__ uint32_t lds_addr(const unsigned char* p);
__device__ __forceinline__ FragAddr frag_addr(const unsigned char* smem, int lane) {
    FragAddr f;
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) f.row[s2] = lds_addr(smem) + tile_off(r, 2 * s2 + hh);
    const int grp = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = grp >> 1;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        const int col = db * 32 + 16 * (grp & 1) + 4 * pp;
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) f.tr[db][hi] = lds_addr(smem) + tile_off(4 * h + q + 8 * hi, col >> 3) + (col & 7) * 2;
    }
    return f;
}
// hand-issued form (the caller waits with tr_wait() before the first use): keeps the issue order of a batch of reads
template <int OFF>
__device__ __forceinline__ u32x4 lds_b128_issue(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <class TR, int OFF>
__device__ __forceinline__ typename TR::vec8 lds_frag_at(uint32_t addr) {
    typedef __attribute__((address_space(3))) const typename TR::vec8 lds_v8;
    return *reinterpret_cast<lds_v8*>((uintptr_t)(addr + OFF));
}

// Transposed fragment straight from a ROW-MAJOR tile (no second, transposed LDS image): the A operand
// T^T[d = db*32 + (lane&31)][k] of a 32x32x16 MFMA whose 16 k-indices are rows 16*g16 .. 16*g16+15 of
// the tile, in the accumulator-row order (element j of lane-half h <-> row 16 g16 + 8 (j>>2) + 4 h + (j&3)),
// i.e. exactly the order in which a preceding 32x32 accumulator is handed over as the B operand.
// Two ds_read_b64_tr_b16 per fragment: each returns 4 consecutive rows at one column per lane.
// The reads are issued as inline asm: hipcc (ROCm 7.2) treats the ds_read_tr builtin as possibly aliasing
// an in-flight LDS-DMA and drains it with s_waitcnt vmcnt(0).  The caller must execute tr_wait() before the
// first use of any tr_issue() result (cdna_hip_programming.md §5.4 rule 18, §5.7).
__device__ __forceinline__ uint32_t lds_addr(const unsigned char* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
__device__ __forceinline__ u32x2 tr_issue_one(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
struct TrPair {
    u32x2 lo, hi;
};
template <int OFF>
__device__ __forceinline__ u32x2 tr_issue_imm(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// fragment (g16, db) of the tile at byte offset TILE of the ring
template <int TILE, int G16>
__device__ __forceinline__ TrPair tr_issue_at(const FragAddr& fa, int db) {
    TrPair t;
    t.lo = tr_issue_imm<TILE + G16 * 2048>(fa.tr[db][0]);
    t.hi = tr_issue_imm<TILE + G16 * 2048>(fa.tr[db][1]);
    return t;
}
__device__ __forceinline__ void tr_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <class TR>
__device__ __forceinline__ typename TR::vec8 tr_vec(const TrPair& t) {
    const u32x4 x = {t.lo[0], t.lo[1], t.hi[0], t.hi[1]};
    return __builtin_bit_cast(typename TR::vec8, x);
}

// two fp32 -> one packed 16-bit pair (v_cvt_pk_*)
template <class TR>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef typename TR::elem E2 __attribute__((ext_vector_type(2)));
    typedef float F2 __attribute__((ext_vector_type(2)));
    const F2 f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, E2));     // one vector conversion -> one v_cvt_pk
}

// Same result as store_rows, but through a wave-private 4 KiB LDS stage so that every global store instruction
// writes 8 complete 128-byte rows (16 B per lane) instead of 8-byte pieces of 32 different rows: 4 store
// instructions per matrix instead of 8, and whole cache lines (cdna_hip_programming.md, attention forward: 'O staged
// through LDS and stored as whole rows').  `stage` must not be in use by any other wave; rows >= rows_valid are
// not written.  The 16-byte chunk index is XOR-ed with the row so that the 8-byte writes spread over the banks.
__device__ __forceinline__ float half_sum(float x);
template <class TR>
__device__ __forceinline__ void store_rows_staged(unsigned char* stage, typename TR::elem* row0_ptr, int64_t row_stride, int rows_valid,
                                                  const f32x16& a0, const f32x16& a1, float mul, int lane, float* colsum64 = nullptr) {
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        u32x2 x, y;
        x[0] = pack2<TR>(a0[4 * g] * mul, a0[4 * g + 1] * mul);
        x[1] = pack2<TR>(a0[4 * g + 2] * mul, a0[4 * g + 3] * mul);
        y[0] = pack2<TR>(a1[4 * g] * mul, a1[4 * g + 1] * mul);
        y[1] = pack2<TR>(a1[4 * g + 2] * mul, a1[4 * g + 3] * mul);
        *reinterpret_cast<u32x2*>(stage + r * 128 + (((g) ^ (r & 7)) << 4) + hh * 8) = x;        // elements 8g + 4hh ..
        *reinterpret_cast<u32x2*>(stage + r * 128 + (((4 + g) ^ (r & 7)) << 4) + hh * 8) = y;    // elements 32 + 8g + 4hh ..
    }
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + (lane >> 3);
        const u32x4 v = *reinterpret_cast<const u32x4*>(stage + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
        if (row < rows_valid) {
            *reinterpret_cast<u32x4*>(row0_ptr + row * row_stride + (lane & 7) * 8) = v;
            if (colsum64) {         // (wave-uniform pointer) the ROUNDED values, as a later column sum of the stored matrix would see them
                const typename TR::vec8 e = __builtin_bit_cast(typename TR::vec8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) cs[j] += (float)e[j];
            }
        }
    }
    if (colsum64) {
        // lanes with the same lane & 7 hold the same 8 columns for different rows: add over lane bits 3, 4, 5 (fixed order)
        // on the vector ALU, no LDS crossbar: DPP row rotation by 8, v_permlane16_swap, v_permlane32_swap
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = cs[j];
            x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
            const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
            x = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
            cs[j] = half_sum(x);
        }
        if (lane < 8) {
            *reinterpret_cast<f32x4*>(colsum64 + 8 * lane) = f32x4{cs[0], cs[1], cs[2], cs[3]};
            *reinterpret_cast<f32x4*>(colsum64 + 8 * lane + 4) = f32x4{cs[4], cs[5], cs[6], cs[7]};
        }
    }
}

// ============================================================================================
// forward
// ============================================================================================
// exchange between lanes l and l^32 on the VALU (v_permlane32_swap), no LDS crossbar
__device__ __forceinline__ float half_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// The running maximum is only advanced when a row's tile maximum exceeds it by more than kDeferLog2
// (in log2 units): P is then bounded by 2^kDeferLog2 instead of 1, which costs nothing in bf16/f16
// relative precision and removes the O-wide rescale from (almost) every tile after the first.
// (Tried and rejected in round 1: subtracting m through an extra MFMA k-step and taking the row sum
// from an all-ones MFMA block -- fewer vector instructions on paper, but hipcc's register allocation
// for the retry loop spilled 51 VGPRs and the kernel ran 2x slower.)
constexpr float kDeferLog2 = 8.f;

// ---- the single key behind the last full tile (sequence length 64 j + 1) -----------------------------------------------
// Lane (r, hh) of a wave owns, of its row r, the head dimensions 16 s + 8 hh + j of the four MFMA k-steps (the operand
// fragments qf / dof / kf ...), and of a transposed accumulator pair (o0, o1) the dimensions 8 g + 4 hh + j and 32 + 8 g + 4 hh + j
// (store_rows_staged).
#ifndef NPCD_SEED_TAIL
#define NPCD_SEED_TAIL 1
#endif
// All three seeds run on the matrix pipe, which has room (the kernels are bound by vector-instruction issue):
//   row_bcast_issue: the row as an MFMA operand in which EVERY row (column) of the 32-wide block is that row -- four 16-byte
//               loads from one address per lane-half; mfma_dot(that, f) is then x . (row of each lane) in all 16 accumulators;
//   outer_seed: acc[d][lane's row] += x[d] * w(lane's row), one matrix instruction per 32 dimensions with a single non-zero
//               k-index (A = x[d] at k = 0, B = w at k = 0; w passes through the 16-bit type like every P / dS).
template <class TR>
__device__ __forceinline__ float mfma_dot(const typename TR::vec8 (&a)[4], const typename TR::vec8 (&b)[4]) {
    f32x16 acc = {0};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = TR::mfma32(a[s], b[s], acc);
    return acc[0];
}
template <class TR>
__device__ __forceinline__ void outer_seed(uint32_t x0_bits, uint32_t x1_bits, float w, int lane, f32x16& a0, f32x16& a1) {
    using V8 = typename TR::vec8;
    using E = typename TR::elem;
    const bool lo = lane < 32;
    V8 bw = V8{0}, x0 = V8{0}, x1 = V8{0};
    bw[0] = lo ? (E)w : (E)0.f;
    x0[0] = lo ? __builtin_bit_cast(E, (uint16_t)x0_bits) : (E)0.f;
    x1[0] = lo ? __builtin_bit_cast(E, (uint16_t)x1_bits) : (E)0.f;
    a0 = TR::mfma32(x0, bw, a0);
    a1 = TR::mfma32(x1, bw, a1);
}

// ---- prologue loads that do not wait for the LDS-DMA --------------------------------------------------------------------
// vmcnt retires in order and the compiler's own wait insertion does not see the hand-issued LDS-DMA: an ordinary load issued
// after the first tiles' DMA makes its first use wait for those tiles, and whatever is computed from it (delta, the seeds) then
// runs AFTER the tiles have landed instead of under their flight.  The per-row operands and the seed rows are therefore loaded
// by hand-issued instructions BEFORE the DMA and awaited with a counted vmcnt that leaves exactly the DMA outstanding
// (cdna_hip_programming.md: "=v" loads, a wait-only statement, counted waits only).  NPCD_ARRIVED ties a value to the wait.
__device__ __forceinline__ u32x4 gload16(const void* ptr) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t gload_u16(const void* ptr) {
    uint32_t v;
    asm volatile("global_load_ushort %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
__device__ __forceinline__ float gload_f32(const void* ptr) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
#define NPCD_ARRIVED(x) asm volatile("" : "+v"(x))
template <class V8>
__device__ __forceinline__ void arrived4(u32x4 (&raw)[4], V8 (&f)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        NPCD_ARRIVED(raw[s]);
        f[s] = __builtin_bit_cast(V8, raw[s]);
    }
}
// the row as an MFMA operand in which every row (column) of the 32-wide block is that row: four 16-byte loads per lane-half
template <class E>
__device__ __forceinline__ void row_bcast_issue(const E* row, int hh, u32x4 (&raw)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) raw[s] = gload16(row + 16 * s + 8 * hh);
}

// column-sum by-product (AttnParams::colsum): this wave's 64-float segment for matrix `which` (0 q, 1 k, 2 v) of head h
__device__ __forceinline__ float* colsum_seg(const AttnParams& p, int64_t row_id, int h, int which) {
    return p.colsum ? p.colsum + row_id * (3 * 64 * (int64_t)p.H) + h * 192 + which * 64 : nullptr;
}
__device__ __forceinline__ void colsum_zero(float* seg, int lane) {      // a wave without rows contributes zeros
    if (seg && lane < 16) *reinterpret_cast<f32x4*>(seg + 4 * lane) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ---- the edge token of a sequence of 128 j + 1 tokens in the BACKWARD ---------------------------------------------------------
// With the last key / query seeded as above, the only work left for a fifth workgroup per (batch, head) would be ONE row: dQ of
// the last query in the dQ pass, dK / dV of the last key in the dK/dV pass.  Those workgroups are not launched (NPCD_EDGE_TOKEN):
//   * every dQ-pass row already holds P and dS against the last key (the seed); the workgroup's 128 rows are summed into partial
//     dV_E = sum_i P_i dO_i and dK_E = sum_i dS_i q_i,
//   * every dK/dV-pass key already holds dS of the last query against it; the workgroup's 128 keys give partial dQ_E = sum_j dS_j k_j,
// each WAVE on its own 32 rows: four matrix instructions per sum over a wave-private row-major LDS image of the register-resident
// operand rows (edge_reduce; no barrier), and attn_bwd_edge_kernel adds the partials of a (batch, head) in (workgroup, wave) order
// and writes the three rows.  Scratch: kEdgeFloats floats per (batch, head, 32-row block) behind the row-constant planes of
// `delta` (npcd_attn_bwd_workspace_floats).
#ifndef NPCD_EDGE_TOKEN
#define NPCD_EDGE_TOKEN 1
#endif
// DIAGNOSTIC builds only (wrong results, timing): what the seeds / the edge reductions cost
#ifndef NPCD_DIAG_NO_SEED
#define NPCD_DIAG_NO_SEED 0
#endif
#ifndef NPCD_DIAG_NO_EDGE_REDUCE
#define NPCD_DIAG_NO_EDGE_REDUCE 0
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int kEdgeFloats = 192;       // [dK_E | dV_E | dQ_E] x 64
__host__ __device__ inline bool edge_mode(int n) { return NPCD_SEED_TAIL && NPCD_EDGE_TOKEN && (n & 127) == 1 && n > 128; }

// this wave's 32 operand rows as a row-major 32 x 64 image at LDS byte offset img_off (wave-private: no barrier)
template <class V8>
__device__ __forceinline__ void edge_put_rows(const FragAddr& fa, uint32_t img_off, const V8 (&f)[4]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
        asm volatile("ds_write_b128 %0, %1" ::"v"(fa.row[s] + img_off), "v"(__builtin_bit_cast(u32x4, f[s])) : "memory");
}
// The weights sit in LDS in the 16-bit type, permuted so that the eight a lane-half needs for one 16-row k-step are contiguous:
// the transposed fragment's k order is the accumulator-row order, rows 16 g + 4 hh + (0..3) and 16 g + 8 + 4 hh + (0..3).
__device__ __forceinline__ int edge_w_slot(int i) { return (i & ~15) | ((i & 4) << 1) | ((i & 8) >> 1) | (i & 3); }
// lo[m] / hi[m] = sum over the wave's 32 rows i of w[i] * image[i][m] / image[i][32 + m]: four matrix instructions in two
// independent chains; every lane returns the 16 sums of the rows m = acc_row(i, hh) (all 32 columns of the product are equal).
// w_addr: LDS byte address of the wave's 32 permuted 16-bit weights.
template <class TR>
__device__ __forceinline__ void edge_reduce(const FragAddr& fa, uint32_t img_off, uint32_t w_addr, int hh, f32x16& lo, f32x16& hi) {
    using V8 = typename TR::vec8;
    TrPair t[2][2];
    u32x4 wv[2];
#pragma unroll
    for (int db = 0; db < 2; ++db) {
        t[db][0].lo = tr_issue_one(fa.tr[db][0] + img_off);
        t[db][0].hi = tr_issue_one(fa.tr[db][1] + img_off);
        t[db][1].lo = tr_issue_one(fa.tr[db][0] + img_off + 2048);
        t[db][1].hi = tr_issue_one(fa.tr[db][1] + img_off + 2048);
    }
    wv[0] = lds_b128_issue<0>(w_addr + hh * 16);
    wv[1] = lds_b128_issue<32>(w_addr + hh * 16);
    tr_wait();
    lo = TR::mfma32(tr_vec<TR>(t[0][0]), __builtin_bit_cast(V8, wv[0]), f32x16{0});
    hi = TR::mfma32(tr_vec<TR>(t[1][0]), __builtin_bit_cast(V8, wv[0]), f32x16{0});
    lo = TR::mfma32(tr_vec<TR>(t[0][1]), __builtin_bit_cast(V8, wv[1]), lo);
    hi = TR::mfma32(tr_vec<TR>(t[1][1]), __builtin_bit_cast(V8, wv[1]), hi);
}
// Lanes 0 and 32 hold a whole 32-dim block between them; the sums go through a wave-private LDS scratch (256 B per 64 sums) so
// that the wave writes them as ONE coalesced store of whole cache lines (stored as 16-byte pieces from two lanes they cost
// 7 us per launch: 16 K waves x 16 partial-line stores).
__device__ __forceinline__ void edge_stage(unsigned char* scratch64, int lane, const f32x16& lo, const f32x16& hi) {
    if ((lane & 31) == 0) {
        const int hh = lane >> 5;
        float* dst64 = reinterpret_cast<float*>(scratch64);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4v x = {lo[4 * g], lo[4 * g + 1], lo[4 * g + 2], lo[4 * g + 3]};
            const f32x4v y = {hi[4 * g], hi[4 * g + 1], hi[4 * g + 2], hi[4 * g + 3]};
            *reinterpret_cast<f32x4v*>(dst64 + 8 * g + 4 * hh) = x;
            *reinterpret_cast<f32x4v*>(dst64 + 32 + 8 * g + 4 * hh) = y;
        }
    }
}

// One 32-key half tile: S^T (4 MFMAs) -> online softmax on 16 scores per lane -> O^T += V^T P^T (4 MFMAs).
// Working in 32-key halves keeps the live register set small enough for 3 waves per SIMD.  The kernel is bound
// by vector-instruction ISSUE (MI355X_MICROARCH.md, per-instruction cycle constants): every address in here
// is a loop-invariant register + an immediate.
template <class TR, bool MASK, int SLOT, int KB>
__device__ __forceinline__ void fwd_half(const FragAddr& fa, const typename TR::vec8 (&qf)[4], f32x16& o0, f32x16& o1, float& m, float& l,
                                         float c, int key0, int n, int hh) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    constexpr int KT = SLOT * 16384 + KB * 4096, VT = SLOT * 16384 + 8192;
    f32x16 s0 = {0};
    s0 = TR::mfma32(lds_frag_at<TR, KT>(fa.row[0]), qf[0], s0);
    s0 = TR::mfma32(lds_frag_at<TR, KT>(fa.row[1]), qf[1], s0);
    s0 = TR::mfma32(lds_frag_at<TR, KT>(fa.row[2]), qf[2], s0);
    s0 = TR::mfma32(lds_frag_at<TR, KT>(fa.row[3]), qf[3], s0);
    TrPair vt[2][2];
    vt[0][0] = tr_issue_at<VT, KB * 2>(fa, 0);
    vt[0][1] = tr_issue_at<VT, KB * 2>(fa, 1);
    vt[1][0] = tr_issue_at<VT, KB * 2 + 1>(fa, 0);
    vt[1][1] = tr_issue_at<VT, KB * 2 + 1>(fa, 1);
    if (MASK) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (key0 + KB * 32 + acc_row(i, hh) >= n) s0[i] = -INFINITY;
    }
    float mx = fmaxf(s0[0], s0[1]);
#pragma unroll
    for (int i = 2; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s0[i]), s0[i + 1]);
    mx = half_max(mx) * c;   // c = scale * log2(e) > 0: m lives in the exp2 domain
    if (__any(mx > m + kDeferLog2)) {
        const float mn = (mx > m + kDeferLog2) ? mx : m;
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            o0[i] *= alpha;
            o1[i] *= alpha;
        }
    }
    float rs = 0.f;
    V8 pf[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[j], c, -m)), b2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[8 + j], c, -m));
        rs += a + b2;
        pf[0][j] = (E)a;
        pf[1][j] = (E)b2;
    }
    l += rs;
    tr_wait();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        o0 = TR::mfma32(tr_vec<TR>(vt[g][0]), pf[g], o0);
        o1 = TR::mfma32(tr_vec<TR>(vt[g][1]), pf[g], o1);
    }
}

// K/V ring of the pipelined forward / dQ kernels: tile t+1 is awaited and tile t+2 requested (into the slot of tile
// t-1) in the MIDDLE of tile t, after the first stage of tile t has read the last fragments of tile t-1.  At most
// one tile is in flight; it has a whole tile time to land.
template <class E, int SLOT>
__device__ __forceinline__ void kv_mid(unsigned char* smem, const E* kb, const E* vb, int64_t sn, int t, int nt, int n, int wave, int lane,
                                       const DmaLane& dl) {
    NPCD_DMA_WAIT_BARRIER(0);
    if (t + 2 < nt) {
        if ((t + 3) * 64 <= n) dma_tile_pair_fast<E, (SLOT + 2) % 3>(smem, kb, vb, sn, (t + 2) * 64, wave, dl);
        else dma_tile_pair(smem + ((SLOT + 2) % 3) * 16384, kb, sn, vb, sn, (t + 2) * 64, n, wave, lane);
    }
}
// a wave without query rows (last query tile of a ragged sequence) only keeps the K/V stream and the barriers going
template <class E>
__device__ __forceinline__ void kv_idle_loop(unsigned char* smem, const E* kb, const E* vb, int64_t sn, int nfull, int nt, int n, int wave,
                                             int lane, const DmaLane& dl) {
    for (int t = 0; t < nfull; ++t) {
        if (t % 3 == 0) kv_mid<E, 0>(smem, kb, vb, sn, t, nt, n, wave, lane, dl);
        else if (t % 3 == 1) kv_mid<E, 1>(smem, kb, vb, sn, t, nt, n, wave, lane, dl);
        else kv_mid<E, 2>(smem, kb, vb, sn, t, nt, n, wave, lane, dl);
    }
    if (nfull == 0) NPCD_DMA_WAIT_BARRIER(0);
    __builtin_amdgcn_s_barrier();       // the one before the row stores
}

// online-softmax update of one 32-key half from its raw scores; returns the 16-bit P operands.
// Round 6 (NPCD_FWD32_TRIGGER, default on): the form of the 64-row kernel's blk_softmax -- P = exp2(c S - m) and its lane sum FIRST, against
// the stored maximum; only when a lane sum leaves (0, 128] (a score more than ~2^7 above the stored maximum, or the very first tile of
// an unseeded row: exp2(+inf)) the exact row maximum is taken, O / l are rescaled and P is recomputed.  The steady-state stage loses the
// eight v_max3, the half-wave exchange, the multiply and the compare-with-margin of the per-tile maximum: ~60 instead of ~72 vector
// instructions per 32 x 32 score block (the kernel is bound by vector issue).  P <= 128 on the fast path; 16-bit relative
// precision is scale-free.  NPCD_FWD32_TRIGGER=0 builds the round-1..5 form (maximum per tile, advanced when exceeded by 2^8).
#ifndef NPCD_FWD32_TRIGGER
#define NPCD_FWD32_TRIGGER 1
#endif
constexpr float kTrigger32 = 128.f;
template <class TR>
__device__ __forceinline__ float fwd_exp_block(const f32x16& s0, float m, float c, u32x4 (&pw)[2]) {
    float pr[16];
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        pr[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[j], c, -m));
        rs += pr[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) pw[j >> 2][j & 3] = pack2<TR>(pr[2 * j], pr[2 * j + 1]);
    return rs;
}
template <class TR>
__device__ __forceinline__ void fwd_softmax(const f32x16& s0, f32x16& o0, f32x16& o1, float& m, float& l, float c, u32x4 (&pw)[2]) {
#if NPCD_FWD32_TRIGGER
    float rs = fwd_exp_block<TR>(s0, m, c, pw);
    if (__any(!(rs <= kTrigger32))) {                 // wave-uniform; !(<=) also catches inf / nan
        float mx = fmaxf(s0[0], s0[1]);
#pragma unroll
        for (int i = 2; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s0[i]), s0[i + 1]);
        const float mn = fmaxf(m, half_max(mx) * c);  // c = scale * log2(e) > 0: m lives in the exp2 domain
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            o0[i] *= alpha;
            o1[i] *= alpha;
        }
        rs = fwd_exp_block<TR>(s0, m, c, pw);
    }
    l += rs;
#else
    float mx = fmaxf(s0[0], s0[1]);
#pragma unroll
    for (int i = 2; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s0[i]), s0[i + 1]);
    mx = half_max(mx) * c;   // c = scale * log2(e) > 0: m lives in the exp2 domain
    if (__any(mx > m + kDeferLog2)) {
        const float mn = (mx > m + kDeferLog2) ? mx : m;
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            o0[i] *= alpha;
            o1[i] *= alpha;
        }
    }
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[2 * j], c, -m)), b2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[2 * j + 1], c, -m));
        rs += a + b2;
        pw[j >> 2][j & 3] = pack2<TR>(a, b2);
    }
    l += rs;
#endif
}

// One pipelined stage of the forward over FULL key tiles: the score products of half (SLOT, KB), then -- behind them on
// the matrix pipe -- O^T += V^T P^T of the PREVIOUS half (PSLOT, PKB, its P in `pw`), while the vector ALU runs the
// online softmax of the new scores.  A rescale of O (rare, deferred maximum) waits for those products by data dependence.
template <class TR, int SLOT, int KB, int PSLOT, int PKB, bool ACC>
__device__ __forceinline__ void fwd_stage(const FragAddr& fa, const typename TR::vec8 (&qf)[4], f32x16& o0, f32x16& o1, float& m, float& l,
                                          float c, u32x4 (&pw)[2]) {
    using V8 = typename TR::vec8;
    constexpr int KT = SLOT * 16384 + KB * 4096, PV = PSLOT * 16384 + 8192;
    u32x4 kr[4];
    kr[0] = lds_b128_issue<KT>(fa.row[0]);
    kr[1] = lds_b128_issue<KT>(fa.row[1]);
    kr[2] = lds_b128_issue<KT>(fa.row[2]);
    kr[3] = lds_b128_issue<KT>(fa.row[3]);
    TrPair vt[2][2];
    if (ACC) {
        vt[0][0] = tr_issue_at<PV, PKB * 2>(fa, 0);     vt[0][1] = tr_issue_at<PV, PKB * 2>(fa, 1);
        vt[1][0] = tr_issue_at<PV, PKB * 2 + 1>(fa, 0); vt[1][1] = tr_issue_at<PV, PKB * 2 + 1>(fa, 1);
    }
    tr_wait();
    f32x16 s0 = {0};
#ifdef NPCD_DIAG_HALF_MFMA
    // DIAGNOSTIC BUILD ONLY (tools/probes/gpu_dev_fp8_bound.py): half of the matrix instructions of every stage are dropped (the
    // operands they would have used are kept alive, all loads, all vector work and the whole control flow stay) -- wrong
    // results, timing only: what a matrix pipe of TWICE the rate (block-scaled fp8, v_mfma_scale_f32_32x32x64_f8f6f4) could
    // buy this kernel at best, before the cost of producing fp8 operands.
#pragma unroll
    for (int s = 0; s < 2; ++s) s0 = TR::mfma32(__builtin_bit_cast(V8, kr[s]), qf[s], s0);
    asm volatile("" ::"v"(kr[2]), "v"(kr[3]));
    if (ACC) {
        o0 = TR::mfma32(tr_vec<TR>(vt[0][0]), __builtin_bit_cast(V8, pw[0]), o0);
        o1 = TR::mfma32(tr_vec<TR>(vt[0][1]), __builtin_bit_cast(V8, pw[0]), o1);
        asm volatile("" ::"v"(vt[1][0].lo), "v"(vt[1][0].hi), "v"(vt[1][1].lo), "v"(vt[1][1].hi), "v"(pw[1]));
    }
#else
#pragma unroll
    for (int s = 0; s < 4; ++s) s0 = TR::mfma32(__builtin_bit_cast(V8, kr[s]), qf[s], s0);
    if (ACC) {
        o0 = TR::mfma32(tr_vec<TR>(vt[0][0]), __builtin_bit_cast(V8, pw[0]), o0);
        o1 = TR::mfma32(tr_vec<TR>(vt[0][1]), __builtin_bit_cast(V8, pw[0]), o1);
        o0 = TR::mfma32(tr_vec<TR>(vt[1][0]), __builtin_bit_cast(V8, pw[1]), o0);
        o1 = TR::mfma32(tr_vec<TR>(vt[1][1]), __builtin_bit_cast(V8, pw[1]), o1);
    }
#endif
    u32x4 nw[2];
    fwd_softmax<TR>(s0, o0, o1, m, l, c, nw);
    pw[0] = nw[0];
    pw[1] = nw[1];
}
template <class TR, int PSLOT, int PKB>
__device__ __forceinline__ void fwd_flush(const FragAddr& fa, f32x16& o0, f32x16& o1, const u32x4 (&pw)[2]) {
    using V8 = typename TR::vec8;
    constexpr int PV = PSLOT * 16384 + 8192;
    TrPair vt[2][2];
    vt[0][0] = tr_issue_at<PV, PKB * 2>(fa, 0);     vt[0][1] = tr_issue_at<PV, PKB * 2>(fa, 1);
    vt[1][0] = tr_issue_at<PV, PKB * 2 + 1>(fa, 0); vt[1][1] = tr_issue_at<PV, PKB * 2 + 1>(fa, 1);
    tr_wait();
    o0 = TR::mfma32(tr_vec<TR>(vt[0][0]), __builtin_bit_cast(V8, pw[0]), o0);
    o1 = TR::mfma32(tr_vec<TR>(vt[0][1]), __builtin_bit_cast(V8, pw[0]), o1);
    o0 = TR::mfma32(tr_vec<TR>(vt[1][0]), __builtin_bit_cast(V8, pw[1]), o0);
    o1 = TR::mfma32(tr_vec<TR>(vt[1][1]), __builtin_bit_cast(V8, pw[1]), o1);
}
constexpr int kRowxFloats = 68;        // O[64], m, l, pad (272 B: records stay 16-byte aligned)
__host__ __device__ inline bool rowx_mode(int n) { return NPCD_SEED_TAIL && n > 256 && ((n - 1) & 255) == 0; }
template <class TR, int SLOT, int RING, bool LEAN> __device__ __forceinline__ void rowx_tile(unsigned char* smem, float c, int lane_in, int wave, float* rec);
template <class TR, int SLOT>
__device__ __forceinline__ void fwd_step(unsigned char* smem, const FragAddr& fa, const DmaLane& dl, const typename TR::elem* kb,
                                         const typename TR::elem* vb, int64_t sn, int t, int nt, int n, int wave, int lane,
                                         const typename TR::vec8 (&qf)[4], f32x16& o0, f32x16& o1, float& m, float& l, float c, u32x4 (&pw)[2],
                                         int xt, float* xrec) {
    constexpr int PREV = (SLOT + 2) % 3;
    fwd_stage<TR, SLOT, 0, PREV, 1, true>(fa, qf, o0, o1, m, l, c, pw);
    kv_mid<typename TR::elem, SLOT>(smem, kb, vb, sn, t, nt, n, wave, lane, dl);
    fwd_stage<TR, SLOT, 1, SLOT, 0, true>(fa, qf, o0, o1, m, l, c, pw);
    if (xt >= 0 && t == xt) rowx_tile<TR, SLOT, 3, true>(smem, c, lane, wave, xrec);      // (wave-uniform) this wave's share of the last query row
}
// the ragged last key tile (fewer than 64 keys): simple, masked, not pipelined
template <class TR, int SLOT>
__device__ __forceinline__ void fwd_tail(const FragAddr& fa, const typename TR::vec8 (&qf)[4], f32x16& o0, f32x16& o1, float& m, float& l, float c,
                                         int key0, int n, int hh) {
    fwd_half<TR, true, SLOT, 0>(fa, qf, o0, o1, m, l, c, key0, n, hh);
    if (key0 + 32 < n) fwd_half<TR, true, SLOT, 1>(fa, qf, o0, o1, m, l, c, key0, n, hh);
}

#ifndef NPCD_FWD_WAVES
#define NPCD_FWD_WAVES 2
#endif
// ROWX (opt-in, NPCD_ATTN_ROWX32=1): see `rowx` below.  Measured at cfg-D: 142-145 us against 128-130 us without it (same process,
// same box, event-bracketed) -- the fifth workgroup it removes is cheap (one active wave, no matrix work) while the row's
// vector work lands on eight busy waves and the merge is one more launch; as a template parameter so that the default
// instantiation keeps its 166 registers (three waves per SIMD: 128 against 148 us at two).
template <class TR, bool ROWX>
__global__ __launch_bounds__(256, NPCD_FWD_WAVES) void attn_fwd_kernel(AttnParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 16384 + 1024 + 256];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // A sequence of 64 j + 1 tokens (the denoiser's: 512 points + the timestep token) would end in a key tile with ONE key.
    // That key never enters the ring: its score and its value row SEED the online softmax (m = s, l = 1, O = v) from a few
    // direct loads and six matrix instructions while the first tiles are still in flight (row_bcast / outer_seed), and the stream
    // below covers nk = n - 1 keys, all in full tiles.
    const bool seeded = NPCD_SEED_TAIL && (p.n & 63) == 1 && p.n > 64;          // kernel-uniform
    const int n = p.n, nk = seeded ? n - 1 : n, nt = (nk + 63) >> 6, nfull = nk >> 6;
    // ROWX: 256 j + 1 tokens with scratch (round 4; the 64-row form had it): no FIFTH workgroup for the single last query row -- at
    // n = 513 it holds a slot for a whole pass over K / V, 1,024 of the 5,120 workgroups of a launch -- the first nt waves of the
    // (batch, head) split that row's keys instead (rowx_tile, one resident key tile each), attn_fwd_rowx_merge_kernel combines them
    const bool rowx = ROWX && p.rowx != nullptr && rowx_mode(p.n);               // kernel-uniform
    const int nqt = rowx ? nk >> 7 : (n + 127) >> 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = bid % nqt, bh = bid / nqt, h = bh % p.H, b = bh / p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const int q0 = qt * 128 + wave * 32;
    const bool wave_active = q0 < n;
    const int qrow = q0 + r;
    const float c = p.scale_log2;
    const DmaLane dl = dma_lane<E>(p.sn, lane);
    const int xt = (ROWX && rowx && qt * 4 + wave < nt) ? qt * 4 + wave : -1;   // the key tile this wave takes for the last query row
    float* xrec = (ROWX && xt >= 0) ? p.rowx + ((int64_t)bh * nt + xt) * kRowxFloats : nullptr;

    // Q fragments stay unscaled (scores are scaled in fp32 after the MFMA, identically in fwd and bwd, so
    // that P recomputed in the backward matches the forward's LSE even for very large logits)
    u32x4 qraw[4], keraw[4] = {};
    uint32_t vx0 = 0, vx1 = 0;
    row_bcast_issue(qb + (int64_t)min(qrow, n - 1) * p.sn, hh, qraw);            // (one row per lane here, not a broadcast)
    const bool seed = seeded && !NPCD_DIAG_NO_SEED;
    if (seed) {
        row_bcast_issue(kb + (int64_t)(n - 1) * p.sn, hh, keraw);
        vx0 = gload_u16(vb + (int64_t)(n - 1) * p.sn + r);
        vx1 = gload_u16(vb + (int64_t)(n - 1) * p.sn + 32 + r);
    }
    // 3-deep LDS ring filled by LDS-DMA (4 wave-instructions per tile pair and wave)
    if (nt > 0) dma_tile_pair(smem, kb, p.sn, vb, p.sn, 0, nk, wave, lane);
    if (nt > 1) dma_tile_pair(smem + 16384, kb, p.sn, vb, p.sn, 64, nk, wave, lane);
    if (nt > 1) vm_wait<8>();
    else if (nt > 0) vm_wait<4>();
    else vm_wait<0>();
    V8 qf[4];
    arrived4(qraw, qf);
    f32x16 o0 = {0}, o1 = {0};
    float m = -INFINITY, l = 0.f;
    if (seed) {
        V8 ke[4];
        arrived4(keraw, ke);
        NPCD_ARRIVED(vx0);
        NPCD_ARRIVED(vx1);
        m = mfma_dot<TR>(ke, qf) * c;                 // the exp2 domain, like every later maximum
        l = 0.5f;                                     // P = 1; the two half-wave partial sums are added at the end
        outer_seed<TR>(vx0, vx1, 1.f, lane, o0, o1);
    }
    if (rowx && wave == 0)       // the last query row -> LDS (128 B, lanes 32..63 repeat lanes 0..31): lands with the first tiles
        dma4_issue(qb + (int64_t)(n - 1) * p.sn, (uint32_t)((lane & 31) * 4), __builtin_amdgcn_readfirstlane(lds_addr(smem) + 3 * 16384 + 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!wave_active) {                                          // wave-uniform
        kv_idle_loop<E>(smem, kb, vb, p.sn, nfull, nt, nk, wave, lane, dl);
        return;
    }
    const FragAddr fa = frag_addr(smem, lane);
    u32x4 pw[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    if (nfull > 0) {
        // tile 0 (slot 0) is peeled: its first stage has no predecessor
        fwd_stage<TR, 0, 0, 2, 1, false>(fa, qf, o0, o1, m, l, c, pw);
        kv_mid<E, 0>(smem, kb, vb, p.sn, 0, nt, nk, wave, lane, dl);
        fwd_stage<TR, 0, 1, 0, 0, true>(fa, qf, o0, o1, m, l, c, pw);
        if (ROWX && xt == 0) rowx_tile<TR, 0, 3, true>(smem, c, lane, wave, xrec);
        for (int t = 1; t < nfull; t += 3) {
            fwd_step<TR, 1>(smem, fa, dl, kb, vb, p.sn, t, nt, nk, wave, lane, qf, o0, o1, m, l, c, pw, xt, xrec);
            if (t + 1 < nfull) fwd_step<TR, 2>(smem, fa, dl, kb, vb, p.sn, t + 1, nt, nk, wave, lane, qf, o0, o1, m, l, c, pw, xt, xrec);
            if (t + 2 < nfull) fwd_step<TR, 0>(smem, fa, dl, kb, vb, p.sn, t + 2, nt, nk, wave, lane, qf, o0, o1, m, l, c, pw, xt, xrec);
        }
        const int last = (nfull - 1) % 3;
        if (last == 0) fwd_flush<TR, 0, 1>(fa, o0, o1, pw);
        else if (last == 1) fwd_flush<TR, 1, 1>(fa, o0, o1, pw);
        else fwd_flush<TR, 2, 1>(fa, o0, o1, pw);
    } else {
        NPCD_DMA_WAIT_BARRIER(0);
    }
    if (nfull < nt) {        // ragged last tile: landed at the mid-point of tile nfull-1 (or in the prologue)
        const int slot = nfull % 3;
        if (slot == 0) fwd_tail<TR, 0>(fa, qf, o0, o1, m, l, c, nfull * 64, nk, hh);
        else if (slot == 1) fwd_tail<TR, 1>(fa, qf, o0, o1, m, l, c, nfull * 64, nk, hh);
        else fwd_tail<TR, 2>(fa, qf, o0, o1, m, l, c, nfull * 64, nk, hh);
    }
    l = half_sum(l);
    __builtin_amdgcn_s_barrier();     // every wave has left the ring: 4 KiB of it per wave stage the output rows
    E* orow0 = static_cast<E*>(p.o_w) + b * p.osb + (int64_t)q0 * p.osn + h * p.osh;
    store_rows_staged<TR>(smem + wave * 4096, orow0, p.osn, n - q0, o0, o1, 1.f / l, lane);
    if (qrow < n && hh == 0) p.lse[(int64_t)(b * p.H + h) * n + qrow] = m * kLn2 + logf(l);
}

// ============================================================================================
// forward, second form (default): 64 query rows per wave
// ============================================================================================
// What the counters of the 32-row form say: 12 vector instructions per matrix instruction, every K / V fragment read
// from LDS feeds ONE matrix instruction, one softmax chain per wave.  This form changes all three:
//   * a wave owns TWO 32-row query blocks A and B (workgroup = 4 waves = 256 rows, two workgroups per CU = two waves per SIMD):
//     every K / V fragment feeds two matrix instructions, and the two blocks are two independent chains -- the vector work of one
//     block is issued behind the matrix instructions of the other (stage = [S_A, PV_A(prev) || softmax B(prev)], [S_B, PV_B(prev) ||
//     softmax A]);
//   * the running maximum is not tracked per tile: P = exp2(c S - m) and its lane sum come first, and only when a lane sum leaves
//     (0, kTrigger] (a score more than ~2^7 above the stored maximum, or the very first tile) the exact row maximum is taken, O / l
//     are rescaled and P is recomputed.  P <= kTrigger on the fast path (16-bit relative precision is scale-free).
// Per 32 x 32 score block that leaves 16 fma + 16 exp2 + 16 adds + 8 packed conversions + one compare for 8 matrix instructions
// (the 32-row form: 12 vector instructions per matrix instruction; this one 7).
// (Tried: scores in the exp2 domain straight from the matrix instruction -- q pre-multiplied by scale * log2(e) in the 16-bit type,
//  -m as the C operand of the first score product.  The C operand costs 32 registers the kernel does not have (243 spills), and
//  the pre-multiplied q rounds COHERENTLY for a query whose elements are equal: a logit of 72 moved by 0.2, 15 % in P.  Dropped.)
struct QBlk {
    f32x16 o0, o1;   // O^T accumulators: head dimensions 0..31 / 32..63 x 32 queries
    float m, l;
};
constexpr float kTrigger = 128.f;

// exact maximum of the block's scores -> move b.m up to it and rescale l / O.  b.m starts at -inf: the first tile comes here
// (exp2(s + inf) = inf trips the check below), alpha = exp2(-inf) = 0 multiplies the zero state, m becomes the tile's maximum.
__device__ __forceinline__ void blk_advance_max(const f32x16& s, QBlk& b, float c) {
    float mx = fmaxf(s[0], s[1]);
#pragma unroll
    for (int i = 2; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s[i]), s[i + 1]);
    const float mn = fmaxf(b.m, half_max(mx) * c);      // c = scale * log2(e) > 0: m lives in the exp2 domain
    const float alpha = __builtin_amdgcn_exp2f(b.m - mn);
    b.m = mn;
    b.l *= alpha;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        b.o0[i] *= alpha;
        b.o1[i] *= alpha;
    }
}

// scores (exp2 domain) -> exp2(s - m) as packed 16-bit P operands + this lane's partial row sum; no state is touched
#ifndef NPCD_DIAG_F64
#define NPCD_DIAG_F64 0      // DIAGNOSTIC builds (wrong results, timing only): 1 no softmax arithmetic, 2 no matrix instructions in the
#endif                       // stages, 3 no LDS fragment reads in the stages, 4 no ring refill / barrier per tile
template <class TR>
__device__ __forceinline__ float blk_exp(const f32x16& s, float m, float c, u32x4 (&pw)[2]) {
#if NPCD_DIAG_F64 == 1
#pragma unroll
    for (int j = 0; j < 8; ++j) pw[j >> 2][j & 3] = pack2<TR>(s[2 * j], s[2 * j + 1]);
    return 1.f;
#endif
    float pr[16];
    float rs = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        pr[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j], c, -m));
        rs += pr[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) pw[j >> 2][j & 3] = pack2<TR>(pr[2 * j], pr[2 * j + 1]);
    return rs;
}
// one block, checked on its own (ragged tail, flush)
template <class TR>
__device__ __forceinline__ void blk_softmax(const f32x16& s, QBlk& b, float c, u32x4 (&pw)[2], bool force) {
    if (force) blk_advance_max(s, b, c);
    float rs = blk_exp<TR>(s, b.m, c, pw);
    if (__any(!(rs <= kTrigger))) {                   // wave-uniform; !(<=) also catches inf / nan
        blk_advance_max(s, b, c);
        rs = blk_exp<TR>(s, b.m, c, pw);
    }
    b.l += rs;
}

// One stage = one 32-key half tile (SLOT, KB) for both blocks: the two score chains share the K fragments, PV of the PREVIOUS half
// (PSLOT, PKB; its P operands in pwA / pwB) shares the V^T fragments and rides on the matrix pipe while the vector ALU turns the new
// scores into the next P operands.  The fast path is ONE basic block (one check for both blocks at its end), so that the
// scheduler can lay the vector work of one block beside the matrix instructions of the other.
template <class TR, int SLOT, int KB, int PSLOT, int PKB, bool ACC>
__device__ __forceinline__ void fwd64_stage(const FragAddr& fa, const typename TR::vec8 (&qA)[4], const typename TR::vec8 (&qB)[4], QBlk& A, QBlk& B,
                                            u32x4 (&pwA)[2], u32x4 (&pwB)[2], float c) {
    using V8 = typename TR::vec8;
    constexpr int KT = SLOT * 16384 + KB * 4096, PV = PSLOT * 16384 + 8192;
    u32x4 kr[4];
    TrPair vt[2][2];
#if NPCD_DIAG_F64 == 3
    for (int s = 0; s < 4; ++s) kr[s] = __builtin_bit_cast(u32x4, qA[s]);
    for (int g = 0; g < 2; ++g)
        for (int h2 = 0; h2 < 2; ++h2) { vt[g][h2].lo = u32x2{pwA[0][0], pwA[0][1]}; vt[g][h2].hi = u32x2{pwB[0][0], pwB[0][1]}; }
#else
    kr[0] = lds_b128_issue<KT>(fa.row[0]);
    kr[1] = lds_b128_issue<KT>(fa.row[1]);
    kr[2] = lds_b128_issue<KT>(fa.row[2]);
    kr[3] = lds_b128_issue<KT>(fa.row[3]);
    if (ACC) {
        vt[0][0] = tr_issue_at<PV, PKB * 2>(fa, 0);     vt[0][1] = tr_issue_at<PV, PKB * 2>(fa, 1);
        vt[1][0] = tr_issue_at<PV, PKB * 2 + 1>(fa, 0); vt[1][1] = tr_issue_at<PV, PKB * 2 + 1>(fa, 1);
    }
    tr_wait();
#endif
    f32x16 sA = {0}, sB = {0};
#if NPCD_DIAG_F64 == 2
    for (int i = 0; i < 16; ++i) { sA[i] = __uint_as_float(kr[i & 3][i >> 2]) * 1e-30f; sB[i] = __uint_as_float(kr[(i + 1) & 3][i >> 2]) * 1e-30f; }
    if (ACC) { asm volatile("" ::"v"(vt[0][0].lo), "v"(vt[0][1].lo), "v"(vt[1][0].lo), "v"(vt[1][1].lo), "v"(vt[0][0].hi), "v"(vt[0][1].hi), "v"(vt[1][0].hi), "v"(vt[1][1].hi)); }
    if (false) {
#else
#pragma unroll
    for (int s = 0; s < 4; ++s) sA = TR::mfma32(__builtin_bit_cast(V8, kr[s]), qA[s], sA);
#pragma unroll
    for (int s = 0; s < 4; ++s) sB = TR::mfma32(__builtin_bit_cast(V8, kr[s]), qB[s], sB);
    if (ACC) {
#endif
        A.o0 = TR::mfma32(tr_vec<TR>(vt[0][0]), __builtin_bit_cast(V8, pwA[0]), A.o0);
        A.o1 = TR::mfma32(tr_vec<TR>(vt[0][1]), __builtin_bit_cast(V8, pwA[0]), A.o1);
        A.o0 = TR::mfma32(tr_vec<TR>(vt[1][0]), __builtin_bit_cast(V8, pwA[1]), A.o0);
        A.o1 = TR::mfma32(tr_vec<TR>(vt[1][1]), __builtin_bit_cast(V8, pwA[1]), A.o1);
        B.o0 = TR::mfma32(tr_vec<TR>(vt[0][0]), __builtin_bit_cast(V8, pwB[0]), B.o0);
        B.o1 = TR::mfma32(tr_vec<TR>(vt[0][1]), __builtin_bit_cast(V8, pwB[0]), B.o1);
        B.o0 = TR::mfma32(tr_vec<TR>(vt[1][0]), __builtin_bit_cast(V8, pwB[1]), B.o0);
        B.o1 = TR::mfma32(tr_vec<TR>(vt[1][1]), __builtin_bit_cast(V8, pwB[1]), B.o1);
    }
    u32x4 nA[2], nB[2];
    float rsA = blk_exp<TR>(sA, A.m, c, nA);
    float rsB = blk_exp<TR>(sB, B.m, c, nB);
    if (__any(!(rsA <= kTrigger) || !(rsB <= kTrigger))) {      // rare (always on the first tile): exact maxima, rescale, redo
        blk_advance_max(sA, A, c);
        blk_advance_max(sB, B, c);
        rsA = blk_exp<TR>(sA, A.m, c, nA);
        rsB = blk_exp<TR>(sB, B.m, c, nB);
    }
    A.l += rsA;
    B.l += rsB;
    pwA[0] = nA[0]; pwA[1] = nA[1];
    pwB[0] = nB[0]; pwB[1] = nB[1];
}
// after the last full stage: PV of the last half for both blocks
template <class TR, int PSLOT, int PKB>
__device__ __forceinline__ void fwd64_flush(const FragAddr& fa, QBlk& A, QBlk& B, const u32x4 (&pwA)[2], const u32x4 (&pwB)[2]) {
    using V8 = typename TR::vec8;
    constexpr int PV = PSLOT * 16384 + 8192;
    TrPair vt[2][2];
    vt[0][0] = tr_issue_at<PV, PKB * 2>(fa, 0);     vt[0][1] = tr_issue_at<PV, PKB * 2>(fa, 1);
    vt[1][0] = tr_issue_at<PV, PKB * 2 + 1>(fa, 0); vt[1][1] = tr_issue_at<PV, PKB * 2 + 1>(fa, 1);
    tr_wait();
    A.o0 = TR::mfma32(tr_vec<TR>(vt[0][0]), __builtin_bit_cast(V8, pwA[0]), A.o0);
    A.o1 = TR::mfma32(tr_vec<TR>(vt[0][1]), __builtin_bit_cast(V8, pwA[0]), A.o1);
    A.o0 = TR::mfma32(tr_vec<TR>(vt[1][0]), __builtin_bit_cast(V8, pwA[1]), A.o0);
    A.o1 = TR::mfma32(tr_vec<TR>(vt[1][1]), __builtin_bit_cast(V8, pwA[1]), A.o1);
    B.o0 = TR::mfma32(tr_vec<TR>(vt[0][0]), __builtin_bit_cast(V8, pwB[0]), B.o0);
    B.o1 = TR::mfma32(tr_vec<TR>(vt[0][1]), __builtin_bit_cast(V8, pwB[0]), B.o1);
    B.o0 = TR::mfma32(tr_vec<TR>(vt[1][0]), __builtin_bit_cast(V8, pwB[1]), B.o0);
    B.o1 = TR::mfma32(tr_vec<TR>(vt[1][1]), __builtin_bit_cast(V8, pwB[1]), B.o1);
}
// ---- the last query row of a sequence of 256 j + 1 tokens, without a workgroup of its own ----------------------------------------
// A workgroup for that single row costs as much slot time as a full one (it streams every K / V tile: +26 us of 119 at cfg-D).
// Instead, the nk / 64 waves that work on a (batch, head) anyway split the row's KEYS: wave w takes the 64 keys of tile w while that
// tile sits in the ring -- a matrix-VECTOR problem, done on the vector ALU from the LDS-resident tile: lane j holds key j (64-deep
// dot product, v_dot2c), wave maximum / sum by DPP-free shuffles (once per item), then lane (d-pair, key parity) accumulates
// sum_j p_j v_j over its 32 keys.  ~250 instructions per wave per item, 2 registers of state.  The wave's partial state
// (m, l, O[64]) goes to scratch; attn_fwd_rowx_merge_kernel combines the partials of a (batch, head), adds the pair (x, x) and writes
// the row and its LSE.

// wave-wide max / sum on the vector ALU (no LDS crossbar): four DPP row rotations inside the 16-lane rows, then v_permlane16_swap
// and v_permlane32_swap across them; every lane ends with the result
template <class Op>
__device__ __forceinline__ float wave_reduce64(float x, Op op) {
    x = op(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128 /* row_ror:8 */, 0xf, 0xf, false)));
    x = op(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124 /* row_ror:4 */, 0xf, 0xf, false)));
    x = op(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x122 /* row_ror:2 */, 0xf, 0xf, false)));
    x = op(x, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121 /* row_ror:1 */, 0xf, 0xf, false)));
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = op(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return op(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
}
__device__ __forceinline__ float wave_max64(float x) { return wave_reduce64(x, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ float wave_sum64(float x) { return wave_reduce64(x, [](float a, float b) { return a + b; }); }
// LEAN: half-size batches of LDS reads (the 32-row forward has ~70 registers free between two stages, the 64-row one ~100)
// rowx_core: `base` + KT / VT = the LDS byte addresses of the K / V tile, `scr` = 256 bytes of wave-private scratch (the row's P),
// `qxa` = the 128-byte copy of the last query row; rowx_tile = the ring form of it (ring slot SLOT, scratch behind the RING slots)
template <class TR, int KT, int VT, bool LEAN>
__device__ __forceinline__ void rowx_core(uint32_t base, uint32_t scr, uint32_t qxa, float c, int lane_in, float* rec) {
    // Everything lane-dependent in here is derived from an OPAQUE copy of the lane number: otherwise the compiler hoists the ~100
    // loop-invariant LDS addresses of the three ring-slot instances out of the tile loop and spills them (82 spilled registers,
    // the whole kernel 20 % slower).  Ring slot and row numbers go into the instructions' immediate offsets.
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    // scores: lane j = key j of the tile; its 128-byte row against the last query row (a 128-byte LDS copy made by the prologue,
    // every lane reads the same address), two 16-byte chunks at a time so that the main loop's state stays in registers
    const uint32_t krow = base + lane * 128, ksw = (uint32_t)tile_swz(lane) << 4;
    float s0 = 0.f, s1 = 0.f;
    if constexpr (LEAN) {
#pragma unroll
        for (int c2 = 0; c2 < 8; c2 += 2) {
            u32x4 kv[2], qv[2];
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) kv[ch] = lds_b128_issue<KT>(krow + (((uint32_t)(c2 + ch) << 4) ^ ksw));
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qv[0]) : "v"(qxa), "n"(c2 * 16) : "memory");
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qv[1]) : "v"(qxa), "n"(c2 * 16 + 16) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kv[0]), "+v"(kv[1]), "+v"(qv[0]), "+v"(qv[1])::"memory");
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                s0 = TR::dot2(kv[ch][0], qv[ch][0], s0);
                s1 = TR::dot2(kv[ch][1], qv[ch][1], s1);
                s0 = TR::dot2(kv[ch][2], qv[ch][2], s0);
                s1 = TR::dot2(kv[ch][3], qv[ch][3], s1);
            }
        }
    } else {
#pragma unroll
    for (int c4 = 0; c4 < 8; c4 += 4) {                   // (between two stages ~100 registers are free: 32 of them here)
        u32x4 kv[4], qv[4];
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) kv[ch] = lds_b128_issue<KT>(krow + (((uint32_t)(c4 + ch) << 4) ^ ksw));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qv[0]) : "v"(qxa), "n"(c4 * 16) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qv[1]) : "v"(qxa), "n"(c4 * 16 + 16) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qv[2]) : "v"(qxa), "n"(c4 * 16 + 32) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qv[3]) : "v"(qxa), "n"(c4 * 16 + 48) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kv[0]), "+v"(kv[1]), "+v"(kv[2]), "+v"(kv[3]), "+v"(qv[0]), "+v"(qv[1]), "+v"(qv[2]), "+v"(qv[3])::"memory");
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            s0 = TR::dot2(kv[ch][0], qv[ch][0], s0);
            s1 = TR::dot2(kv[ch][1], qv[ch][1], s1);
            s0 = TR::dot2(kv[ch][2], qv[ch][2], s0);
            s1 = TR::dot2(kv[ch][3], qv[ch][3], s1);
        }
    }
    }
    const float sc = (s0 + s1) * c;                       // exp2 domain
    const float m = wave_max64(sc);
    const float pj = __builtin_amdgcn_exp2f(sc - m);
    const float l = wave_sum64(pj);
    asm volatile("ds_write_b32 %0, %1" ::"v"(scr + lane * 4), "v"(pj) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // O[d] = sum_j p_j v_j[d]: lane = (key parity, d-pair); V row j = 2 i + parity, elements 2 dp, 2 dp + 1.  tile_swz(j) only
    // depends on i, so a read's address is one lane register XOR a constant + an immediate.  Eight keys per batch in flight.
    const int dp = lane & 31, par = lane >> 5;
    const uint32_t vrow = base + par * 128 + (dp & 3) * 4, vch = (uint32_t)(dp >> 2) << 4, prow = scr + par * 4;
    float o0 = 0.f, o1 = 0.f;
    constexpr int VB = LEAN ? 4 : 16;
#pragma unroll
    for (int i0 = 0; i0 < 32; i0 += VB) {
        uint32_t vv[16];
        float pp[16];
#pragma unroll
        for (int i = 0; i < VB; ++i) {
            const int ii = i0 + i;
            const uint32_t swz = (uint32_t)((((ii & 1) << 2) | ((ii >> 1) & 3)) << 4);       // tile_swz(2 ii + parity) << 4
            const uint32_t va = vrow + (vch ^ swz);
            switch (ii) {     // (immediates must be literal)
#define NPCD_RX(I) case I: asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(vv[i]) : "v"(va), "n"(VT + (I) * 256) : "memory"); \
                           asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(pp[i]) : "v"(prow), "n"((I) * 8) : "memory"); break;
                NPCD_RX(0) NPCD_RX(1) NPCD_RX(2) NPCD_RX(3) NPCD_RX(4) NPCD_RX(5) NPCD_RX(6) NPCD_RX(7) NPCD_RX(8) NPCD_RX(9) NPCD_RX(10)
                NPCD_RX(11) NPCD_RX(12) NPCD_RX(13) NPCD_RX(14) NPCD_RX(15) NPCD_RX(16) NPCD_RX(17) NPCD_RX(18) NPCD_RX(19) NPCD_RX(20)
                NPCD_RX(21) NPCD_RX(22) NPCD_RX(23) NPCD_RX(24) NPCD_RX(25) NPCD_RX(26) NPCD_RX(27) NPCD_RX(28) NPCD_RX(29) NPCD_RX(30)
                NPCD_RX(31)
#undef NPCD_RX
            }
        }
        if constexpr (LEAN)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(pp[0]), "+v"(pp[1]), "+v"(pp[2]), "+v"(pp[3])::"memory");
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(vv[0]), "+v"(vv[1]), "+v"(vv[2]), "+v"(vv[3]), "+v"(vv[4]), "+v"(vv[5]), "+v"(vv[6]), "+v"(vv[7]), "+v"(pp[0]), "+v"(pp[1]),
                       "+v"(pp[2]), "+v"(pp[3]), "+v"(pp[4]), "+v"(pp[5]), "+v"(pp[6]), "+v"(pp[7])::"memory");
        if constexpr (!LEAN)
            asm volatile("" : "+v"(vv[8]), "+v"(vv[9]), "+v"(vv[10]), "+v"(vv[11]), "+v"(vv[12]), "+v"(vv[13]), "+v"(vv[14]), "+v"(vv[15]), "+v"(pp[8]), "+v"(pp[9]),
                           "+v"(pp[10]), "+v"(pp[11]), "+v"(pp[12]), "+v"(pp[13]), "+v"(pp[14]), "+v"(pp[15]));
#pragma unroll
        for (int i = 0; i < VB; ++i) {
            o0 = __builtin_fmaf(pp[i], TR::lo(vv[i]), o0);
            o1 = __builtin_fmaf(pp[i], TR::hi(vv[i]), o1);
        }
    }
    o0 = half_sum(o0);
    o1 = half_sum(o1);
    if (lane < 32) *reinterpret_cast<float2*>(rec + 2 * dp) = make_float2(o0, o1);
    if (lane == 0) *reinterpret_cast<float2*>(rec + 64) = make_float2(m, l);
}
template <class TR, int SLOT, int RING, bool LEAN>
__device__ __forceinline__ void rowx_tile(unsigned char* smem, float c, int lane_in, int wave, float* rec) {
    const uint32_t base = lds_addr(smem);
    rowx_core<TR, SLOT * 16384, SLOT * 16384 + 8192, LEAN>(base, base + RING * 16384 + wave * 256, base + RING * 16384 + 1024, c, lane_in, rec);
}
// one 64-thread block per (batch, head): combine the nw partial states, add the pair (x, x), write the row and its LSE
template <class E>
__global__ __launch_bounds__(64) void attn_fwd_rowx_merge_kernel(AttnParams p, int nw) {
    const int bh = blockIdx.x, h = bh % p.H, b = bh / p.H, d = threadIdx.x, x = p.n - 1;
    const E* qx = static_cast<const E*>(p.q) + b * p.sb + (int64_t)x * p.sn + h * p.sh;
    const E* kx = static_cast<const E*>(p.k) + b * p.sb + (int64_t)x * p.sn + h * p.sh;
    const E* vx = static_cast<const E*>(p.v) + b * p.sb + (int64_t)x * p.sn + h * p.sh;
    const float* rec = p.rowx + (int64_t)bh * nw * kRowxFloats;
    const float sxx = wave_sum64((float)qx[d] * (float)kx[d]) * p.scale_log2;
    float M = sxx;
    for (int w = 0; w < nw; ++w) M = fmaxf(M, rec[w * kRowxFloats + 64]);
    const float pxx = __builtin_amdgcn_exp2f(sxx - M);
    float L = pxx, O = pxx * (float)vx[d];
    for (int w = 0; w < nw; ++w) {
        const float a = __builtin_amdgcn_exp2f(rec[w * kRowxFloats + 64] - M);
        L += a * rec[w * kRowxFloats + 65];
        O += a * rec[w * kRowxFloats + d];
    }
    E* orow = static_cast<E*>(p.o_w) + b * p.osb + (int64_t)x * p.osn + h * p.osh;
    orow[d] = (E)(O / L);
    if (d == 0) p.lse[(int64_t)bh * p.n + x] = M * kLn2 + logf(L);
}

template <class TR, int SLOT>
__device__ __forceinline__ void fwd64_step(unsigned char* smem, const FragAddr& fa, const DmaLane& dl, const typename TR::elem* kb,
                                           const typename TR::elem* vb, int64_t sn, int t, int nt, int n, int wave, int lane,
                                           const typename TR::vec8 (&qA)[4], const typename TR::vec8 (&qB)[4], QBlk& A, QBlk& B,
                                           u32x4 (&pwA)[2], u32x4 (&pwB)[2], int xt, float c, float* xrec) {
    constexpr int PREV = (SLOT + 2) % 3;
    fwd64_stage<TR, SLOT, 0, PREV, 1, true>(fa, qA, qB, A, B, pwA, pwB, c);
#if NPCD_DIAG_F64 != 4
    kv_mid<typename TR::elem, SLOT>(smem, kb, vb, sn, t, nt, n, wave, lane, dl);
#endif
    fwd64_stage<TR, SLOT, 1, SLOT, 0, true>(fa, qA, qB, A, B, pwA, pwB, c);
    if (t == xt) rowx_tile<TR, SLOT, 3, false>(smem, c, lane, wave, xrec);      // (wave-uniform) this wave's share of the last query row
}
// one 32-key half of the ragged last key tile for one block: masked, exact maximum, not pipelined
template <class TR, int SLOT, int KB>
__device__ __forceinline__ void fwd64_half_masked(const FragAddr& fa, const typename TR::vec8 (&qf)[4], QBlk& b, float c, int key0, int n, int hh) {
    using V8 = typename TR::vec8;
    constexpr int KT = SLOT * 16384 + KB * 4096, VT = SLOT * 16384 + 8192;
    f32x16 s0 = {0};
    s0 = TR::mfma32(lds_frag_at<TR, KT>(fa.row[0]), qf[0], s0);
    s0 = TR::mfma32(lds_frag_at<TR, KT>(fa.row[1]), qf[1], s0);
    s0 = TR::mfma32(lds_frag_at<TR, KT>(fa.row[2]), qf[2], s0);
    s0 = TR::mfma32(lds_frag_at<TR, KT>(fa.row[3]), qf[3], s0);
    TrPair vt[2][2];
    vt[0][0] = tr_issue_at<VT, KB * 2>(fa, 0);
    vt[0][1] = tr_issue_at<VT, KB * 2>(fa, 1);
    vt[1][0] = tr_issue_at<VT, KB * 2 + 1>(fa, 0);
    vt[1][1] = tr_issue_at<VT, KB * 2 + 1>(fa, 1);
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (key0 + KB * 32 + acc_row(i, hh) >= n) s0[i] = -INFINITY;
    u32x4 pw[2];
    blk_softmax<TR>(s0, b, c, pw, true);
    tr_wait();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        b.o0 = TR::mfma32(tr_vec<TR>(vt[g][0]), __builtin_bit_cast(V8, pw[g]), b.o0);
        b.o1 = TR::mfma32(tr_vec<TR>(vt[g][1]), __builtin_bit_cast(V8, pw[g]), b.o1);
    }
}
template <class TR, int SLOT>
__device__ __forceinline__ void fwd64_tail(const FragAddr& fa, const typename TR::vec8 (&qA)[4], const typename TR::vec8 (&qB)[4], QBlk& A, QBlk& B,
                                           float c, int key0, int n, int hh) {
    fwd64_half_masked<TR, SLOT, 0>(fa, qA, A, c, key0, n, hh);
    fwd64_half_masked<TR, SLOT, 0>(fa, qB, B, c, key0, n, hh);
    if (key0 + 32 < n) {
        fwd64_half_masked<TR, SLOT, 1>(fa, qA, A, c, key0, n, hh);
        fwd64_half_masked<TR, SLOT, 1>(fa, qB, B, c, key0, n, hh);
    }
}

template <class TR>
__global__ __launch_bounds__(256, 2) void attn_fwd64_kernel(AttnParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 16384 + 1024 + 256];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool seeded = NPCD_SEED_TAIL && (p.n & 63) == 1 && p.n > 64;          // kernel-uniform (see attn_fwd_kernel)
    const int n = p.n, nk = seeded ? n - 1 : n, nt = (nk + 63) >> 6, nfull = nk >> 6;
    // 256 j + 1 tokens with scratch: no workgroup for the last query row, the waves of the (batch, head) split its keys (rowx_tile)
    const bool rowx = p.rowx != nullptr && rowx_mode(p.n);                       // kernel-uniform
    const int nqt = rowx ? nk >> 8 : (n + 255) >> 8;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = bid % nqt, bh = bid / nqt, h = bh % p.H, b = bh / p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const int q0 = qt * 256 + wave * 64;
    const bool wave_active = q0 < n;
    const float c = p.scale_log2;
    const DmaLane dl = dma_lane<E>(p.sn, lane);
    const int xt = rowx ? qt * 4 + wave : -1;                   // the key tile this wave takes for the last query row
    float* xrec = rowx ? p.rowx + ((int64_t)bh * nt + xt) * kRowxFloats : nullptr;

    u32x4 qrawA[4], qrawB[4], keraw[4] = {};
    uint32_t vx0 = 0, vx1 = 0;
    row_bcast_issue(qb + (int64_t)min(q0 + r, n - 1) * p.sn, hh, qrawA);            // (one row per lane, not a broadcast)
    row_bcast_issue(qb + (int64_t)min(q0 + 32 + r, n - 1) * p.sn, hh, qrawB);
    if (seeded) {
        row_bcast_issue(kb + (int64_t)(n - 1) * p.sn, hh, keraw);
        vx0 = gload_u16(vb + (int64_t)(n - 1) * p.sn + r);
        vx1 = gload_u16(vb + (int64_t)(n - 1) * p.sn + 32 + r);
    }
#ifdef NPCD_TIMELINE64
    const bool tl_on = (blockIdx.x == NPCD_TIMELINE64) && wave == 0;
    long long tl[40];
    for (int i = 0; i < 40; ++i) tl[i] = 0;
    NPCD_TS(0);
#endif
    if (nt > 0) dma_tile_pair(smem, kb, p.sn, vb, p.sn, 0, nk, wave, lane);
    if (nt > 1) dma_tile_pair(smem + 16384, kb, p.sn, vb, p.sn, 64, nk, wave, lane);
#ifdef NPCD_TIMELINE64
    NPCD_TS(1);
#endif
    if (nt > 1) vm_wait<8>();
    else if (nt > 0) vm_wait<4>();
    else vm_wait<0>();
    V8 qA[4], qB[4];
    arrived4(qrawA, qA);
    arrived4(qrawB, qB);
#ifdef NPCD_TIMELINE64
    NPCD_TS(2);
#endif
    QBlk A, B;
    A.o0 = A.o1 = B.o0 = B.o1 = f32x16{0};
    A.m = B.m = -INFINITY;
    A.l = B.l = 0.f;
    if (seeded) {
        V8 ke[4];
        arrived4(keraw, ke);
        NPCD_ARRIVED(vx0);
        NPCD_ARRIVED(vx1);
        A.m = mfma_dot<TR>(ke, qA) * c;               // the exp2 domain, like every later maximum
        B.m = mfma_dot<TR>(ke, qB) * c;
        A.l = B.l = 0.5f;                             // P = 1; the two half-wave partial sums are added at the end
        outer_seed<TR>(vx0, vx1, 1.f, lane, A.o0, A.o1);
        outer_seed<TR>(vx0, vx1, 1.f, lane, B.o0, B.o1);
    }
    if (rowx && wave == 0)       // the last query row -> LDS (128 B, lanes 32..63 repeat lanes 0..31): lands with the first tiles
        dma4_issue(qb + (int64_t)(n - 1) * p.sn, (uint32_t)((lane & 31) * 4), __builtin_amdgcn_readfirstlane(lds_addr(smem) + 3 * 16384 + 1024));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!wave_active) {                                          // wave-uniform
        kv_idle_loop<E>(smem, kb, vb, p.sn, nfull, nt, nk, wave, lane, dl);
        return;
    }
    const FragAddr fa = frag_addr(smem, lane);
    u32x4 pwA[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}}, pwB[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
#ifdef NPCD_TIMELINE64
    NPCD_TS(3);
#define NPCD_TS_TILE(t) do { if (tl_on && (t) < 14) { __builtin_amdgcn_sched_barrier(0); tl[4 + (t)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define NPCD_TS_TILE(t) do { } while (0)
#endif
    if (nfull > 0) {
        fwd64_stage<TR, 0, 0, 2, 1, false>(fa, qA, qB, A, B, pwA, pwB, c);
        kv_mid<E, 0>(smem, kb, vb, p.sn, 0, nt, nk, wave, lane, dl);
        fwd64_stage<TR, 0, 1, 0, 0, true>(fa, qA, qB, A, B, pwA, pwB, c);
        if (xt == 0) rowx_tile<TR, 0, 3, false>(smem, c, lane, wave, xrec);
        NPCD_TS_TILE(0);
        for (int t = 1; t < nfull; t += 3) {
            fwd64_step<TR, 1>(smem, fa, dl, kb, vb, p.sn, t, nt, nk, wave, lane, qA, qB, A, B, pwA, pwB, xt, c, xrec);
            NPCD_TS_TILE(t);
            if (t + 1 < nfull) fwd64_step<TR, 2>(smem, fa, dl, kb, vb, p.sn, t + 1, nt, nk, wave, lane, qA, qB, A, B, pwA, pwB, xt, c, xrec);
            NPCD_TS_TILE(t + 1);
            if (t + 2 < nfull) fwd64_step<TR, 0>(smem, fa, dl, kb, vb, p.sn, t + 2, nt, nk, wave, lane, qA, qB, A, B, pwA, pwB, xt, c, xrec);
            NPCD_TS_TILE(t + 2);
        }
        const int last = (nfull - 1) % 3;
        if (last == 0) fwd64_flush<TR, 0, 1>(fa, A, B, pwA, pwB);
        else if (last == 1) fwd64_flush<TR, 1, 1>(fa, A, B, pwA, pwB);
        else fwd64_flush<TR, 2, 1>(fa, A, B, pwA, pwB);
    } else {
        NPCD_DMA_WAIT_BARRIER(0);
    }
    if (nfull < nt) {        // ragged last tile: landed at the mid-point of tile nfull-1 (or in the prologue)
        const int slot = nfull % 3;
        if (slot == 0) fwd64_tail<TR, 0>(fa, qA, qB, A, B, c, nfull * 64, nk, hh);
        else if (slot == 1) fwd64_tail<TR, 1>(fa, qA, qB, A, B, c, nfull * 64, nk, hh);
        else fwd64_tail<TR, 2>(fa, qA, qB, A, B, c, nfull * 64, nk, hh);
    }
#ifdef NPCD_TIMELINE64
    NPCD_TS(20);
#endif
    A.l = half_sum(A.l);
    B.l = half_sum(B.l);
    __builtin_amdgcn_s_barrier();     // every wave has left the ring: 4 KiB of it per wave stage the output rows
#ifdef NPCD_TIMELINE64
    NPCD_TS(21);
#endif
    E* orow0 = static_cast<E*>(p.o_w) + b * p.osb + (int64_t)q0 * p.osn + h * p.osh;
    store_rows_staged<TR>(smem + wave * 4096, orow0, p.osn, n - q0, A.o0, A.o1, 1.f / A.l, lane);
    if (q0 + 32 < n) store_rows_staged<TR>(smem + wave * 4096, orow0 + 32 * p.osn, p.osn, n - q0 - 32, B.o0, B.o1, 1.f / B.l, lane);
    float* lrow = p.lse + (int64_t)(b * p.H + h) * n;
    if (hh == 0) {
        if (q0 + r < n) lrow[q0 + r] = A.m * kLn2 + logf(A.l);
        if (q0 + 32 + r < n) lrow[q0 + 32 + r] = B.m * kLn2 + logf(B.l);
    }
#ifdef NPCD_TIMELINE64
    NPCD_TS(22);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NPCD_TS(23);
    if (tl_on && lane == 0)
        for (int i = 0; i < 40; ++i) g_timeline[i] = tl[i];
#endif
}

// ============================================================================================
// forward, third form: K and V of a (batch, head) RESIDENT in LDS (sequences of 513 tokens: the denoiser's)
// ============================================================================================
// What the stage-loop probe says (tools/probes/attn_shape_probe.hip, round 5): the pipelined stage of the kernels above runs at
// 408 cycles per 32 x 32 score block and SIMD with three waves per SIMD, 499 with two, 780 with one -- 1.0 / 0.88 / 0.70 PFLOP/s --
// while the forward at n = 513 delivers 0.59-0.65: more than 40 % of its time is NOT the loop.  It is the shape of the launch:
// five workgroups per (batch, head), each streaming all of K / V through a ring for 128 (or one!) query rows, a prologue and an
// epilogue per eight tiles, a barrier + DMA wait per tile, 6.67 rounds of workgroups on the chip.
// Here ONE workgroup of eight waves owns a whole (batch, head): its 512 keys x (K, V) = 128 KB are requested by LDS-DMA up front, tile
// after tile, and stay; a wave owns 64 query rows (the two-block stage of the 64-row form: every fragment read feeds two matrix
// instructions) and walks the eight tiles as they land -- one counted wait + barrier per tile, nothing to refill, no ring; the
// 513th key seeds the softmax state as before; the 513th QUERY row is split over the eight waves (wave w takes the keys of tile w:
// rowx_core) and merged by wave 0 through LDS -- no fifth workgroup, no scratch in HBM, no second launch.  K / V are read from HBM
// exactly once per (batch, head); 1,024 workgroups = four rounds of one per CU (two waves per SIMD).
// LDS: [0, 16 KB) the last query row, the eight partial records of its softmax, per-wave scratch; [16 KB + 16 KB t) tile t (K | V).
// The fragment addresses point ONE TILE BELOW the current tile and are advanced by 16 KB per tile, so that every stage is the same
// instantiation (ring slot 1, previous slot 0) and every offset fits the instructions' 16-bit immediate.
constexpr int kResMisc = 16384, kResRecOff = 256, kResScrOff = 4096, kResTiles = 8;
constexpr int kResLds = kResMisc + kResTiles * 16384;
__host__ __device__ inline bool fwd_res_shape(int n) { return n == 64 * kResTiles + 1; }

template <class E>
__device__ __forceinline__ void rowx_merge_lds(const AttnParams& p, int b, int h, const float* rec, int nw, int d) {
    const int x = p.n - 1;
    const E* qx = static_cast<const E*>(p.q) + b * p.sb + (int64_t)x * p.sn + h * p.sh;
    const E* kx = static_cast<const E*>(p.k) + b * p.sb + (int64_t)x * p.sn + h * p.sh;
    const E* vx = static_cast<const E*>(p.v) + b * p.sb + (int64_t)x * p.sn + h * p.sh;
    const float sxx = wave_sum64((float)qx[d] * (float)kx[d]) * p.scale_log2;
    float M = sxx;
    for (int w = 0; w < nw; ++w) M = fmaxf(M, rec[w * kRowxFloats + 64]);
    const float pxx = __builtin_amdgcn_exp2f(sxx - M);
    float L = pxx, O = pxx * (float)vx[d];
    for (int w = 0; w < nw; ++w) {
        const float a = __builtin_amdgcn_exp2f(rec[w * kRowxFloats + 64] - M);
        L += a * rec[w * kRowxFloats + 65];
        O += a * rec[w * kRowxFloats + d];
    }
    E* orow = static_cast<E*>(p.o_w) + b * p.osb + (int64_t)x * p.osn + h * p.osh;
    orow[d] = (E)(O / L);
    if (d == 0) p.lse[(int64_t)(b * p.H + h) * p.n + x] = M * kLn2 + logf(L);
}

template <int N> struct ResWait { static __device__ __forceinline__ void at(int t) { if (t == N) vm_wait<2 * (6 - N)>(); else ResWait<N + 1>::at(t); } };
template <> struct ResWait<7> { static __device__ __forceinline__ void at(int) {} };

template <class TR>
__global__ __launch_bounds__(512, 2) void attn_fwd_res_kernel(AttnParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char rsmem[];
    unsigned char* tiles = rsmem + kResMisc;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = p.n, nk = n - 1;
    const int bh = xcd_remap(blockIdx.x, gridDim.x), h = bh % p.H, b = bh / p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const int q0 = wave * 64;
    const float c = p.scale_log2;
    const DmaLane dl = dma_lane<E>(p.sn, lane);
    // ---- prologue: the wave's query rows and the seed rows (oldest), the last query row -> LDS, then ALL tiles ------------------
    u32x4 qrawA[4], qrawB[4], keraw[4];
    row_bcast_issue(qb + (int64_t)(q0 + r) * p.sn, hh, qrawA);             // (one row per lane, not a broadcast)
    row_bcast_issue(qb + (int64_t)(q0 + 32 + r) * p.sn, hh, qrawB);
    row_bcast_issue(kb + (int64_t)nk * p.sn, hh, keraw);
    uint32_t vx0 = gload_u16(vb + (int64_t)nk * p.sn + r), vx1 = gload_u16(vb + (int64_t)nk * p.sn + 32 + r);
    // (every wave issues the 128-byte copy of the last query row -- same bytes, same place -- so that all waves count the same DMAs)
    dma4_issue(qb + (int64_t)nk * p.sn, (uint32_t)((lane & 31) * 4), __builtin_amdgcn_readfirstlane(lds_addr(rsmem)));
    {
        // wave w: rows 16 (w & 3) .. + 15 of every K (w < 4) or V (w >= 4) tile = two 1-KiB pieces per tile
        const bool second = wave >= 4;
        const int w4 = wave & 3;
        const char* sbase = reinterpret_cast<const char*>((second ? vb : kb) + (int64_t)(w4 * 16) * p.sn);
        const uint32_t dst0 = lds_addr(tiles) + (second ? 8192 : 0) + w4 * 2048;
#pragma unroll
        for (int t = 0; t < kResTiles; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                dma16_issue(sbase + ((int64_t)t * 64 + i * 8) * p.sn * (int64_t)sizeof(E), dl.off[i & 1], __builtin_amdgcn_readfirstlane(dst0 + t * 16384 + i * 1024));
    }
    vm_wait<2 * kResTiles + 1>();                     // the register loads have landed; the DMAs may all still be in flight
    V8 qA[4], qB[4], ke[4];
    arrived4(qrawA, qA);
    arrived4(qrawB, qB);
    arrived4(keraw, ke);
    NPCD_ARRIVED(vx0);
    NPCD_ARRIVED(vx1);
    QBlk A, B;
    A.o0 = A.o1 = B.o0 = B.o1 = f32x16{0};
    A.m = mfma_dot<TR>(ke, qA) * c;                   // the 513th key seeds the state (see attn_fwd_kernel): m = its score, l = 1, O = its value row
    B.m = mfma_dot<TR>(ke, qB) * c;
    A.l = B.l = 0.5f;                                 // (the two half-wave partial sums are added at the end)
    outer_seed<TR>(vx0, vx1, 1.f, lane, A.o0, A.o1);
    outer_seed<TR>(vx0, vx1, 1.f, lane, B.o0, B.o1);
    vm_wait<2 * (kResTiles - 1)>();                   // this wave's pieces of tile 0 and the last query row
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    FragAddr fa = frag_addr(rsmem, lane);             // one tile below tile 0
    u32x4 pwA[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}}, pwB[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    const uint32_t scr = lds_addr(rsmem) + kResScrOff + wave * 256, qxa = lds_addr(rsmem);
    float* rec = reinterpret_cast<float*>(rsmem + kResRecOff) + wave * kRowxFloats;
    // tile 0 (peeled: its first stage has no predecessor)
    fwd64_stage<TR, 1, 0, 0, 1, false>(fa, qA, qB, A, B, pwA, pwB, c);
    vm_wait<2 * (kResTiles - 2)>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    fwd64_stage<TR, 1, 1, 1, 0, true>(fa, qA, qB, A, B, pwA, pwB, c);
    if (wave == 0) rowx_core<TR, 0, 8192, false>(lds_addr(tiles), scr, qxa, c, lane, rec);
#pragma unroll 1
    for (int t = 1; t < kResTiles; ++t) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) fa.row[s2] += 16384;
#pragma unroll
        for (int db = 0; db < 2; ++db) { fa.tr[db][0] += 16384; fa.tr[db][1] += 16384; }
        fwd64_stage<TR, 1, 0, 0, 1, true>(fa, qA, qB, A, B, pwA, pwB, c);
        if (t + 1 < kResTiles) {                      // tile t + 1: this wave's pieces have landed -> barrier -> everybody's have
            ResWait<1>::at(t);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        fwd64_stage<TR, 1, 1, 1, 0, true>(fa, qA, qB, A, B, pwA, pwB, c);
        if (t == wave) rowx_core<TR, 0, 8192, false>(lds_addr(tiles) + t * 16384, scr, qxa, c, lane, rec);      // (wave-uniform)
    }
    fwd64_flush<TR, 1, 1>(fa, A, B, pwA, pwB);
    A.l = half_sum(A.l);
    B.l = half_sum(B.l);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the partial record of the last query row is written
    __builtin_amdgcn_s_barrier();     // every wave has left the tiles: 4 KiB of them per wave stage the output rows
    asm volatile("" ::: "memory");
    E* orow0 = static_cast<E*>(p.o_w) + b * p.osb + (int64_t)q0 * p.osn + h * p.osh;
    store_rows_staged<TR>(tiles + wave * 4096, orow0, p.osn, 64, A.o0, A.o1, 1.f / A.l, lane);
    store_rows_staged<TR>(tiles + wave * 4096, orow0 + 32 * p.osn, p.osn, 32, B.o0, B.o1, 1.f / B.l, lane);
    float* lrow = p.lse + (int64_t)(b * p.H + h) * n;
    if (hh == 0) {
        lrow[q0 + r] = A.m * kLn2 + logf(A.l);
        lrow[q0 + 32 + r] = B.m * kLn2 + logf(B.l);
    }
    if (wave == kResTiles - 1) rowx_merge_lds<E>(p, b, h, reinterpret_cast<const float*>(rsmem + kResRecOff), kResTiles, lane);
}

// ============================================================================================
// backward, pass 1: dQ (+ delta)
// ============================================================================================
// One pipelined stage of the dQ pass over FULL key tiles (no masking): the score products of half (SLOT, KB), then --
// behind them on the matrix pipe -- dQ^T += K^T dS^T of the PREVIOUS half (PSLOT, PKB, its dS in `dw`), while the
// vector ALU turns the new scores into the next dS.
template <class TR, int SLOT, int KB, int PSLOT, int PKB, bool ACC>
__device__ __forceinline__ void dq_stage(const FragAddr& fa, const typename TR::vec8 (&qf)[4], const typename TR::vec8 (&dof)[4],
                                         f32x16& dq0, f32x16& dq1, float c, float lse2, float delta, u32x4 (&dw)[2]) {
    using V8 = typename TR::vec8;
    constexpr int KT = SLOT * 16384 + KB * 4096, VT = KT + 8192, PK = PSLOT * 16384;
    u32x4 kr[4], vr[4];
    kr[0] = lds_b128_issue<KT>(fa.row[0]); vr[0] = lds_b128_issue<VT>(fa.row[0]);
    kr[1] = lds_b128_issue<KT>(fa.row[1]); vr[1] = lds_b128_issue<VT>(fa.row[1]);
    kr[2] = lds_b128_issue<KT>(fa.row[2]); vr[2] = lds_b128_issue<VT>(fa.row[2]);
    kr[3] = lds_b128_issue<KT>(fa.row[3]); vr[3] = lds_b128_issue<VT>(fa.row[3]);
    tr_wait();
    f32x16 s0 = {0}, d0 = {0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        s0 = TR::mfma32(__builtin_bit_cast(V8, kr[s]), qf[s], s0);
        d0 = TR::mfma32(__builtin_bit_cast(V8, vr[s]), dof[s], d0);
    }
    TrPair kt[2][2];
    if (ACC) {
        __builtin_amdgcn_sched_barrier(0);
        kt[0][0] = tr_issue_at<PK, PKB * 2>(fa, 0);     kt[0][1] = tr_issue_at<PK, PKB * 2>(fa, 1);
        kt[1][0] = tr_issue_at<PK, PKB * 2 + 1>(fa, 0); kt[1][1] = tr_issue_at<PK, PKB * 2 + 1>(fa, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 2; s < 4; ++s) {
        s0 = TR::mfma32(__builtin_bit_cast(V8, kr[s]), qf[s], s0);
        d0 = TR::mfma32(__builtin_bit_cast(V8, vr[s]), dof[s], d0);
    }
    if (ACC) {
        tr_wait();
        dq0 = TR::mfma32(tr_vec<TR>(kt[0][0]), __builtin_bit_cast(V8, dw[0]), dq0);
        dq1 = TR::mfma32(tr_vec<TR>(kt[0][1]), __builtin_bit_cast(V8, dw[0]), dq1);
        dq0 = TR::mfma32(tr_vec<TR>(kt[1][0]), __builtin_bit_cast(V8, dw[1]), dq0);
        dq1 = TR::mfma32(tr_vec<TR>(kt[1][1]), __builtin_bit_cast(V8, dw[1]), dq1);
    }
    u32x4 nw[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[2 * j], c, -lse2)), p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[2 * j + 1], c, -lse2));
        nw[j >> 2][j & 3] = pack2<TR>(p0 * (d0[2 * j] - delta), p1 * (d0[2 * j + 1] - delta));
    }
    dw[0] = nw[0];
    dw[1] = nw[1];
}
template <class TR, int PSLOT, int PKB>
__device__ __forceinline__ void dq_flush(const FragAddr& fa, f32x16& dq0, f32x16& dq1, const u32x4 (&dw)[2]) {
    using V8 = typename TR::vec8;
    constexpr int PK = PSLOT * 16384;
    TrPair kt[2][2];
    kt[0][0] = tr_issue_at<PK, PKB * 2>(fa, 0);     kt[0][1] = tr_issue_at<PK, PKB * 2>(fa, 1);
    kt[1][0] = tr_issue_at<PK, PKB * 2 + 1>(fa, 0); kt[1][1] = tr_issue_at<PK, PKB * 2 + 1>(fa, 1);
    tr_wait();
    dq0 = TR::mfma32(tr_vec<TR>(kt[0][0]), __builtin_bit_cast(V8, dw[0]), dq0);
    dq1 = TR::mfma32(tr_vec<TR>(kt[0][1]), __builtin_bit_cast(V8, dw[0]), dq1);
    dq0 = TR::mfma32(tr_vec<TR>(kt[1][0]), __builtin_bit_cast(V8, dw[1]), dq0);
    dq1 = TR::mfma32(tr_vec<TR>(kt[1][1]), __builtin_bit_cast(V8, dw[1]), dq1);
}

template <class TR, int SLOT>
__device__ __forceinline__ void dq_step(unsigned char* smem, const FragAddr& fa, const DmaLane& dl, const typename TR::elem* kb,
                                        const typename TR::elem* vb, int64_t sn, int t, int nt, int n, int wave, int lane,
                                        const typename TR::vec8 (&qf)[4], const typename TR::vec8 (&dof)[4], f32x16& dq0, f32x16& dq1,
                                        float c, float lse2, float delta, u32x4 (&dw)[2]) {
    constexpr int PREV = (SLOT + 2) % 3;
    dq_stage<TR, SLOT, 0, PREV, 1, true>(fa, qf, dof, dq0, dq1, c, lse2, delta, dw);
    kv_mid<typename TR::elem, SLOT>(smem, kb, vb, sn, t, nt, n, wave, lane, dl);
    dq_stage<TR, SLOT, 1, SLOT, 0, true>(fa, qf, dof, dq0, dq1, c, lse2, delta, dw);
}

// the ragged last key tile (fewer than 64 keys): simple, masked, not pipelined
template <class TR, int SLOT, int KB>
__device__ __forceinline__ void dq_tail_half(const FragAddr& fa, const typename TR::vec8 (&qf)[4], const typename TR::vec8 (&dof)[4],
                                             f32x16& dq0, f32x16& dq1, float c, float lse2, float delta, int key0, int n, int hh) {
    using V8 = typename TR::vec8;
    constexpr int KT = SLOT * 16384 + KB * 4096, VT = KT + 8192, KTR = SLOT * 16384;
    f32x16 s0 = {0}, d0 = {0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        s0 = TR::mfma32(lds_frag_at<TR, KT>(fa.row[s]), qf[s], s0);
        d0 = TR::mfma32(lds_frag_at<TR, VT>(fa.row[s]), dof[s], d0);
    }
    TrPair kt[2][2];
    kt[0][0] = tr_issue_at<KTR, KB * 2>(fa, 0);     kt[0][1] = tr_issue_at<KTR, KB * 2>(fa, 1);
    kt[1][0] = tr_issue_at<KTR, KB * 2 + 1>(fa, 0); kt[1][1] = tr_issue_at<KTR, KB * 2 + 1>(fa, 1);
    u32x4 dw[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[2 * j], c, -lse2)), p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[2 * j + 1], c, -lse2));
        if (key0 + KB * 32 + acc_row(2 * j, hh) >= n) p0 = 0.f;
        if (key0 + KB * 32 + acc_row(2 * j + 1, hh) >= n) p1 = 0.f;
        dw[j >> 2][j & 3] = pack2<TR>(p0 * (d0[2 * j] - delta), p1 * (d0[2 * j + 1] - delta));
    }
    tr_wait();
    dq0 = TR::mfma32(tr_vec<TR>(kt[0][0]), __builtin_bit_cast(V8, dw[0]), dq0);
    dq1 = TR::mfma32(tr_vec<TR>(kt[0][1]), __builtin_bit_cast(V8, dw[0]), dq1);
    dq0 = TR::mfma32(tr_vec<TR>(kt[1][0]), __builtin_bit_cast(V8, dw[1]), dq0);
    dq1 = TR::mfma32(tr_vec<TR>(kt[1][1]), __builtin_bit_cast(V8, dw[1]), dq1);
}
template <class TR, int SLOT>
__device__ __forceinline__ void dq_tail(const FragAddr& fa, const typename TR::vec8 (&qf)[4], const typename TR::vec8 (&dof)[4], f32x16& dq0,
                                        f32x16& dq1, float c, float lse2, float delta, int key0, int n, int hh) {
    dq_tail_half<TR, SLOT, 0>(fa, qf, dof, dq0, dq1, c, lse2, delta, key0, n, hh);
    if (key0 + 32 < n) dq_tail_half<TR, SLOT, 1>(fa, qf, dof, dq0, dq1, c, lse2, delta, key0, n, hh);
}

#ifndef NPCD_DQ_WAVES
#define NPCD_DQ_WAVES 2
#endif
template <class TR>
__global__ __launch_bounds__(256, NPCD_DQ_WAVES) void attn_bwd_dq_kernel(AttnParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * 16384 + 512];       // ring + [dS_E | P_E] of the 128 rows (edge token), 16-bit
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool seeded = NPCD_SEED_TAIL && (p.n & 63) == 1 && p.n > 64;          // the single last key is folded into the initial dQ (see attn_fwd_kernel)
    const int n = p.n, nk = seeded ? n - 1 : n, nt = (nk + 63) >> 6, nfull = nk >> 6;
    const bool edge = edge_mode(p.n);          // the last row has no workgroup of its own: its dQ comes from the dK/dV pass
    const int nqt = edge ? n >> 7 : (n + 127) >> 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = bid % nqt, bh = bid / nqt, h = bh % p.H, b = bh / p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const E* ob = static_cast<const E*>(p.out) + b * p.osb + h * p.osh;
    const E* dob = static_cast<const E*>(p.dout) + b * p.osb + h * p.osh;
    const int q0 = qt * 128 + wave * 32;
    const bool wave_active = q0 < n;
    const int qrow = q0 + r;
    const bool row_ok = qrow < n;
    const float c = p.scale_log2;
    const DmaLane dl = dma_lane<E>(p.sn, lane);

    // the per-row operands (and the seed rows) first, then the K/V stream; delta and the seeds are computed under its flight
    const int qclamp = min(qrow, n - 1);
    const bool seed = seeded && !NPCD_DIAG_NO_SEED;
    u32x4 qraw[4], doraw[4], oraw[4], keraw[4] = {}, veraw[4] = {};
    uint32_t kx0 = 0, kx1 = 0;
    row_bcast_issue(qb + (int64_t)qclamp * p.sn, hh, qraw);
    row_bcast_issue(dob + (int64_t)qclamp * p.osn, hh, doraw);
    row_bcast_issue(ob + (int64_t)qclamp * p.osn, hh, oraw);
    float lse_row = gload_f32(p.lse + (int64_t)(b * p.H + h) * n + qclamp);   // rows past the end duplicate the last row; never stored
    if (seed) {
        row_bcast_issue(kb + (int64_t)(n - 1) * p.sn, hh, keraw);
        row_bcast_issue(vb + (int64_t)(n - 1) * p.sn, hh, veraw);
        kx0 = gload_u16(kb + (int64_t)(n - 1) * p.sn + r);
        kx1 = gload_u16(kb + (int64_t)(n - 1) * p.sn + 32 + r);
    }
    if (nt > 0) dma_tile_pair(smem, kb, p.sn, vb, p.sn, 0, nk, wave, lane);
    if (nt > 1) dma_tile_pair(smem + 16384, kb, p.sn, vb, p.sn, 64, nk, wave, lane);
    if (nt > 1) vm_wait<8>();
    else if (nt > 0) vm_wait<4>();
    else vm_wait<0>();

    V8 qf[4], dof[4], of[4];
    arrived4(qraw, qf);
    arrived4(doraw, dof);
    arrived4(oraw, of);
    NPCD_ARRIVED(lse_row);
    float delta = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) delta += (float)dof[s][j] * (float)of[s][j];
    delta = half_sum(delta);
    {   // row constants of the dK/dV pass (its initial accumulators), planes [2][B][H][npad]; the pad rows of the last
        // 64-row tile get -inf / 0 so that their P and dS vanish there without masking
        const int npad = ((n + 63) >> 6) << 6;
        if (hh == 0 && qrow < npad) {
            const int64_t at = (int64_t)(b * p.H + h) * npad + qrow;
            p.delta[at] = row_ok ? -lse_row / p.scale : -INFINITY;
            p.delta[(int64_t)p.B * p.H * npad + at] = row_ok ? -delta : 0.f;
        }
    }
    const float lse2 = lse_row * kLog2e;
    f32x16 dq0 = {0}, dq1 = {0};
    if (seed) {       // dQ of the last key: dS = P (dP - delta), dQ = dS k
        V8 ke[4], ve[4];
        arrived4(keraw, ke);
        arrived4(veraw, ve);
        NPCD_ARRIVED(kx0);
        NPCD_ARRIVED(kx1);
        const float s1 = mfma_dot<TR>(ke, qf), dp1 = mfma_dot<TR>(ve, dof);
        const float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(s1, c, -lse2)), ds1 = pe * (dp1 - delta);
        if (edge && hh == 0) {
            E* we = reinterpret_cast<E*>(smem + 3 * 16384) + wave * 64;
            we[edge_w_slot(r)] = (E)ds1;
            we[32 + edge_w_slot(r)] = (E)pe;
        }
        outer_seed<TR>(kx0, kx1, ds1, lane, dq0, dq1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!wave_active) {      // wave-uniform: no query rows (ragged last query tile): keep the stream and the barriers going
        colsum_zero(colsum_seg(p, ((int64_t)b * nqt + qt) * 4 + wave, h, 0), lane);
        kv_idle_loop<E>(smem, kb, vb, p.sn, nfull, nt, nk, wave, lane, dl);
        return;
    }
    const FragAddr fa = frag_addr(smem, lane);
    u32x4 dw[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    if (nfull > 0) {
        // tile 0 (slot 0) is peeled: its first stage has no predecessor
        dq_stage<TR, 0, 0, 2, 1, false>(fa, qf, dof, dq0, dq1, c, lse2, delta, dw);
        kv_mid<E, 0>(smem, kb, vb, p.sn, 0, nt, nk, wave, lane, dl);
        dq_stage<TR, 0, 1, 0, 0, true>(fa, qf, dof, dq0, dq1, c, lse2, delta, dw);
        for (int t = 1; t < nfull; t += 3) {
            dq_step<TR, 1>(smem, fa, dl, kb, vb, p.sn, t, nt, nk, wave, lane, qf, dof, dq0, dq1, c, lse2, delta, dw);
            if (t + 1 < nfull) dq_step<TR, 2>(smem, fa, dl, kb, vb, p.sn, t + 1, nt, nk, wave, lane, qf, dof, dq0, dq1, c, lse2, delta, dw);
            if (t + 2 < nfull) dq_step<TR, 0>(smem, fa, dl, kb, vb, p.sn, t + 2, nt, nk, wave, lane, qf, dof, dq0, dq1, c, lse2, delta, dw);
        }
        const int last = (nfull - 1) % 3;
        if (last == 0) dq_flush<TR, 0, 1>(fa, dq0, dq1, dw);
        else if (last == 1) dq_flush<TR, 1, 1>(fa, dq0, dq1, dw);
        else dq_flush<TR, 2, 1>(fa, dq0, dq1, dw);
    } else {
        NPCD_DMA_WAIT_BARRIER(0);
    }
    if (nfull < nt) {        // ragged last tile: landed at the mid-point of tile nfull-1 (or just above)
        const int slot = nfull % 3;
        if (slot == 0) dq_tail<TR, 0>(fa, qf, dof, dq0, dq1, c, lse2, delta, nfull * 64, nk, hh);
        else if (slot == 1) dq_tail<TR, 1>(fa, qf, dof, dq0, dq1, c, lse2, delta, nfull * 64, nk, hh);
        else dq_tail<TR, 2>(fa, qf, dof, dq0, dq1, c, lse2, delta, nfull * 64, nk, hh);
    }
    __builtin_amdgcn_s_barrier();     // every wave has left the ring: 4 KiB of it per wave stage the gradient rows
    f32x16 k0 = {0}, k1 = {0}, v0 = {0}, v1 = {0};
    if (edge && !NPCD_DIAG_NO_EDGE_REDUCE) {       // partial dK_E (Q rows weighted by dS_E) and dV_E (dO rows weighted by P_E) of this wave's 32 rows
        const uint32_t img = 16384 + wave * 8192, wa = lds_addr(smem) + 3 * 16384 + wave * 128;
        edge_put_rows(fa, img, qf);
        edge_put_rows(fa, img + 4096, dof);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        edge_reduce<TR>(fa, img, wa, hh, k0, k1);
        edge_reduce<TR>(fa, img + 4096, wa + 64, hh, v0, v1);
    }
    E* grow0 = static_cast<E*>(p.dq) + b * p.gsb + (int64_t)q0 * p.gsn + h * p.gsh;
    store_rows_staged<TR>(smem + wave * 4096, grow0, p.gsn, n - q0, dq0, dq1, p.scale, lane, colsum_seg(p, ((int64_t)b * nqt + qt) * 4 + wave, h, 0));
    if (edge && !NPCD_DIAG_NO_EDGE_REDUCE) {
        float* part = p.delta + 2 * (int64_t)p.B * p.H * (((n + 63) >> 6) << 6) + (((int64_t)bh * nqt + qt) * 4 + wave) * kEdgeFloats;
        unsigned char* scratch = smem + 16384 + wave * 8192;        // the wave's own image, read out by edge_reduce
        edge_stage(scratch, lane, k0, k1);
        edge_stage(scratch + 256, lane, v0, v1);
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        reinterpret_cast<f32x2v*>(part)[lane] = reinterpret_cast<const f32x2v*>(scratch)[lane];
    }
}

// ============================================================================================
// backward, pass 2: dK, dV
// ============================================================================================
// Ring slot of the dK/dV pass: Q tile, dO tile (row-major, also read transposed), then the two per-row
// constants of the 64 query rows written by the dQ pass: -lse/scale [64] and -delta [64] (fp32).
constexpr int kDkdvSlot = 2 * 8192 + 512;


// per-wave LDS-DMA context of the dK/dV pass: waves 0,1 stream the Q tile, waves 2,3 the dO tile (32 rows each);
// waves with an even / odd index also fetch the 64 -lse/scale / -delta values of the tile (4 bytes per lane).
template <class E>
struct QdoStream {
    const E* base;        // q or dout of this (batch, head), advanced to the wave's first row
    const E* q;           // for the clamped (ragged last tile) form
    const E* dout;
    const float* stat;    // row-constant plane of this wave
    int64_t stride, qstride, dostride;
    DmaLane dl;
    int dst;              // byte offset of the wave's first piece inside a slot
    int stat_dst;
};

template <class E>
__device__ __forceinline__ void qdo_prefetch(unsigned char* slot, const QdoStream<E>& qs, int t, int n, int wave, int lane) {
    const int row0 = t * 64;
    const uint32_t sl = lds_addr(slot);
    if (row0 + 64 <= n) {
        const char* sbase = reinterpret_cast<const char*>(qs.base + (int64_t)row0 * qs.stride);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dma16_issue(sbase + (int64_t)i * 8 * qs.stride * (int64_t)sizeof(E), qs.dl.off[i & 1],
                        __builtin_amdgcn_readfirstlane(sl + qs.dst + i * 1024));
    } else {
        dma_tile_pair(slot, qs.q, qs.qstride, qs.dout, qs.dostride, row0, n, wave, lane);
    }
    dma4_issue(qs.stat, (uint32_t)(row0 + lane) * 4u, __builtin_amdgcn_readfirstlane(sl + qs.stat_dst));
}


struct DkdvState {
    f32x16 dk0, dk1, dv0, dv1;
#ifdef NPCD_TIMELINE
    long long* tl;
    bool tl_on;
#endif
};
#ifdef NPCD_TIMELINE
#ifndef NPCD_TL_TILE
#define NPCD_TL_TILE 3
#endif
#define NPCD_TS_STEP(i)                                                                        \
    do {                                                                                       \
        if (a.tl_on && t == NPCD_TL_TILE) { __builtin_amdgcn_sched_barrier(0); a.tl[(i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } \
    } while (0)
#else
#define NPCD_TS_STEP(i) do { } while (0)
#endif

// One stage of the dK/dV pass = one 32-row query sub-block (SLOT, SUB) of the ring, keys on the lanes:
//   A  one batch of LDS reads: the sub-block's row constants and Q / dO row fragments, plus (ACC) the transposed
//      dO / Q fragments of the PREVIOUS sub-block (PSLOT, PSUB), one wait;
//   B  S'^T = K Q^T - lse/scale and dP'^T = V dO^T - delta: the row constants enter as the initial accumulators,
//      so that P = exp2(c S') and dS = P dP' need no subtraction (cdna_hip_programming.md, 'Row constants as
//      the initial accumulator');
//   C  dV^T += dO^T P, dK^T += Q^T dS of the previous sub-block (its P / dS sit in pf / df) -- issued behind B on
//      the matrix pipe while the vector ALU turns this sub-block's scores into the next pf / df.
template <class TR, int SLOT, int SUB, int PSLOT, int PSUB, bool ACC>
__device__ __forceinline__ void dkdv_stage(const FragAddr& fa, uint32_t st_addr, const typename TR::vec8 (&kf)[4],
                                           const typename TR::vec8 (&vf)[4], float c, DkdvState& a,
                                           typename TR::vec8 (&pf)[2], typename TR::vec8 (&df)[2]) {
    using V8 = typename TR::vec8;
    constexpr int QT = SLOT * kDkdvSlot + SUB * 4096, DT = QT + 8192, ST = SLOT * kDkdvSlot + 16384 + SUB * 128;
    constexpr int PQ = PSLOT * kDkdvSlot, PD = PQ + 8192;
    // ---- A
    u32x4 si[4], di[4], qr[4], dr[4];
    si[0] = lds_b128_issue<ST>(st_addr);       si[1] = lds_b128_issue<ST + 32>(st_addr);
    si[2] = lds_b128_issue<ST + 64>(st_addr);  si[3] = lds_b128_issue<ST + 96>(st_addr);
    qr[0] = lds_b128_issue<QT>(fa.row[0]);     qr[1] = lds_b128_issue<QT>(fa.row[1]);
    qr[2] = lds_b128_issue<QT>(fa.row[2]);     qr[3] = lds_b128_issue<QT>(fa.row[3]);
    di[0] = lds_b128_issue<ST + 256>(st_addr); di[1] = lds_b128_issue<ST + 288>(st_addr);
    di[2] = lds_b128_issue<ST + 320>(st_addr); di[3] = lds_b128_issue<ST + 352>(st_addr);
    dr[0] = lds_b128_issue<DT>(fa.row[0]);     dr[1] = lds_b128_issue<DT>(fa.row[1]);
    dr[2] = lds_b128_issue<DT>(fa.row[2]);     dr[3] = lds_b128_issue<DT>(fa.row[3]);
    tr_wait();
    // ---- B (first half), then the transposed fragments of the previous sub-block are requested so that their LDS
    // latency passes under the score products
    f32x16 s, d;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            s[4 * g + bq] = __uint_as_float(si[g][bq]);
            d[4 * g + bq] = __uint_as_float(di[g][bq]);
        }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        s = TR::mfma32(__builtin_bit_cast(V8, qr[ks]), kf[ks], s);
        d = TR::mfma32(__builtin_bit_cast(V8, dr[ks]), vf[ks], d);
    }
    TrPair to[2][2], tq[2][2];
    if (ACC) {
        __builtin_amdgcn_sched_barrier(0);
        to[0][0] = tr_issue_at<PD, 2 * PSUB>(fa, 0);     to[0][1] = tr_issue_at<PD, 2 * PSUB>(fa, 1);
        to[1][0] = tr_issue_at<PD, 2 * PSUB + 1>(fa, 0); to[1][1] = tr_issue_at<PD, 2 * PSUB + 1>(fa, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 2; ks < 4; ++ks) {
        s = TR::mfma32(__builtin_bit_cast(V8, qr[ks]), kf[ks], s);
        d = TR::mfma32(__builtin_bit_cast(V8, dr[ks]), vf[ks], d);
    }
    // ---- C: [dV products] [request Q^T] [first half of the vector work] [dK products] [second half]
    u32x4 pw[2], dw[2];
    auto softmax_half = [&](int h2) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * h2 + jj;
            // rows past the end of the sequence carry -inf / 0 as row constants (written by the dQ pass): P = dS = 0
            const float p0 = __builtin_amdgcn_exp2f(s[2 * j] * c), p1 = __builtin_amdgcn_exp2f(s[2 * j + 1] * c);
            pw[h2][jj] = pack2<TR>(p0, p1);
            dw[h2][jj] = pack2<TR>(p0 * d[2 * j], p1 * d[2 * j + 1]);
        }
    };
    if (ACC) {
        tr_wait();
        a.dv0 = TR::mfma32(tr_vec<TR>(to[0][0]), pf[0], a.dv0);
        a.dv1 = TR::mfma32(tr_vec<TR>(to[0][1]), pf[0], a.dv1);
        a.dv0 = TR::mfma32(tr_vec<TR>(to[1][0]), pf[1], a.dv0);
        a.dv1 = TR::mfma32(tr_vec<TR>(to[1][1]), pf[1], a.dv1);
        __builtin_amdgcn_sched_barrier(0);
        tq[0][0] = tr_issue_at<PQ, 2 * PSUB>(fa, 0);     tq[0][1] = tr_issue_at<PQ, 2 * PSUB>(fa, 1);
        tq[1][0] = tr_issue_at<PQ, 2 * PSUB + 1>(fa, 0); tq[1][1] = tr_issue_at<PQ, 2 * PSUB + 1>(fa, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    softmax_half(0);
    if (ACC) {
        tr_wait();
        a.dk0 = TR::mfma32(tr_vec<TR>(tq[0][0]), df[0], a.dk0);
        a.dk1 = TR::mfma32(tr_vec<TR>(tq[0][1]), df[0], a.dk1);
        a.dk0 = TR::mfma32(tr_vec<TR>(tq[1][0]), df[1], a.dk0);
        a.dk1 = TR::mfma32(tr_vec<TR>(tq[1][1]), df[1], a.dk1);
    }
    softmax_half(1);
    pf[0] = __builtin_bit_cast(V8, pw[0]);
    pf[1] = __builtin_bit_cast(V8, pw[1]);
    df[0] = __builtin_bit_cast(V8, dw[0]);
    df[1] = __builtin_bit_cast(V8, dw[1]);
}

// the accumulation products of the very last sub-block
template <class TR, int PSLOT, int PSUB>
__device__ __forceinline__ void dkdv_flush(const FragAddr& fa, DkdvState& a, const typename TR::vec8 (&pf)[2],
                                           const typename TR::vec8 (&df)[2]) {
    constexpr int PQ = PSLOT * kDkdvSlot, PD = PQ + 8192;
    TrPair to[2][2], tq[2][2];
    to[0][0] = tr_issue_at<PD, 2 * PSUB>(fa, 0);     to[0][1] = tr_issue_at<PD, 2 * PSUB>(fa, 1);
    to[1][0] = tr_issue_at<PD, 2 * PSUB + 1>(fa, 0); to[1][1] = tr_issue_at<PD, 2 * PSUB + 1>(fa, 1);
    tq[0][0] = tr_issue_at<PQ, 2 * PSUB>(fa, 0);     tq[0][1] = tr_issue_at<PQ, 2 * PSUB>(fa, 1);
    tq[1][0] = tr_issue_at<PQ, 2 * PSUB + 1>(fa, 0); tq[1][1] = tr_issue_at<PQ, 2 * PSUB + 1>(fa, 1);
    tr_wait();
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        a.dv0 = TR::mfma32(tr_vec<TR>(to[g][0]), pf[g], a.dv0);
        a.dv1 = TR::mfma32(tr_vec<TR>(to[g][1]), pf[g], a.dv1);
        a.dk0 = TR::mfma32(tr_vec<TR>(tq[g][0]), df[g], a.dk0);
        a.dk1 = TR::mfma32(tr_vec<TR>(tq[g][1]), df[g], a.dk1);
    }
}

// One 64-row query tile t >= 1 in ring slot SLOT.  Ring protocol: tile t+1 is awaited (and tile t+2 requested, into
// the slot of tile t-1) in the MIDDLE of tile t, after the last reads of tile t-1 (the transposed fragments read by the
// first stage of tile t).  Returns false when the tile's second sub-block lies completely past the sequence.
#define NPCD_DKDV_MID() NPCD_DMA_WAIT_BARRIER(0)
template <class TR, int SLOT>
__device__ __forceinline__ bool dkdv_step(unsigned char* smem, const FragAddr& fa, uint32_t st_addr, const QdoStream<typename TR::elem>& qs,
                                          int t, int nt, int n, int wave, int lane, const typename TR::vec8 (&kf)[4],
                                          const typename TR::vec8 (&vf)[4], float c, DkdvState& a, typename TR::vec8 (&pf)[2],
                                          typename TR::vec8 (&df)[2]) {
    constexpr int PREV = (SLOT + 2) % 3;
    NPCD_TS_STEP(22);
    dkdv_stage<TR, SLOT, 0, PREV, 1, true>(fa, st_addr, kf, vf, c, a, pf, df);
    NPCD_TS_STEP(25);
    NPCD_DKDV_MID();                                             // tile t+1 landed; every wave is done with tile t-1
    NPCD_TS_STEP(26);
    if (t + 2 < nt) qdo_prefetch(smem + PREV * kDkdvSlot, qs, t + 2, n, wave, lane);
    NPCD_TS_STEP(27);
    if (t * 64 + 32 >= n) return false;
    dkdv_stage<TR, SLOT, 1, SLOT, 0, true>(fa, st_addr, kf, vf, c, a, pf, df);
    NPCD_TS_STEP(30);
    return true;
}

#ifndef NPCD_DKDV_WAVES
#define NPCD_DKDV_WAVES 2
#endif
template <class TR>
__global__ __launch_bounds__(256, NPCD_DKDV_WAVES) void attn_bwd_dkdv_kernel(AttnParams p) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // A sequence of 64 j + 1 tokens: the single last QUERY row never enters the ring, its dK / dV contributions are the
    // initial accumulators (from direct loads and a few matrix instructions while the first tiles are in flight; cf. attn_fwd_kernel),
    // and the stream below covers nq = n - 1 query rows in full tiles.
    const bool seeded = NPCD_SEED_TAIL && (p.n & 63) == 1 && p.n > 64;          // kernel-uniform
    const int n = p.n, nq = seeded ? n - 1 : n, nt = (nq + 63) >> 6;
    const bool edge = edge_mode(p.n);          // the last key has no workgroup of its own: its dK / dV come from the dQ pass
    const int nkt = edge ? n >> 7 : (n + 127) >> 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int kt = bid % nkt, bh = bid / nkt, h = bh % p.H, b = bh / p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const E* dob = static_cast<const E*>(p.dout) + b * p.osb + h * p.osh;
    const int key0 = kt * 128 + wave * 32;
    const bool wave_active = key0 < n;
    const int key = key0 + r;
    const bool key_ok = key < n;

#ifdef NPCD_TIMELINE
    const bool tl_on = (blockIdx.x == NPCD_TIMELINE) && wave == 0;
    long long tl[40];
    for (int i = 0; i < 40; ++i) tl[i] = 0;
#endif
    NPCD_TS(0);
    QdoStream<E> qs;
    {
        const bool second = wave >= 2;
        const int w2 = wave & 1;
        qs.q = qb; qs.dout = dob; qs.qstride = p.sn; qs.dostride = p.osn;
        qs.stride = second ? p.osn : p.sn;
        qs.base = (second ? dob : qb) + (int64_t)(w2 * 32) * qs.stride;
        qs.dl = dma_lane<E>(qs.stride, lane);
        qs.dst = (second ? 8192 : 0) + w2 * 4096;
        const int npad = ((n + 63) >> 6) << 6;       // row constants: planes [2][B][H][npad], pad rows hold -inf / 0
        qs.stat = p.delta + ((int64_t)w2 * p.B * p.H + (b * p.H + h)) * npad;
        qs.stat_dst = 16384 + w2 * 256;
    }
    if (!wave_active) {      // wave-uniform: no keys in this wave (ragged last key block): keep the stream and the barriers going
        colsum_zero(colsum_seg(p, ((int64_t)b * nkt + kt) * 4 + wave, h, 1), lane);
        colsum_zero(colsum_seg(p, ((int64_t)b * nkt + kt) * 4 + wave, h, 2), lane);
        qdo_prefetch(dsmem, qs, 0, nq, wave, lane);
        if (nt > 1) qdo_prefetch(dsmem + kDkdvSlot, qs, 1, nq, wave, lane);
        if (nt > 1) NPCD_DMA_WAIT_BARRIER(5);
        else NPCD_DMA_WAIT_BARRIER(0);
        for (int t = 0; t < nt; ++t) {
            NPCD_DMA_WAIT_BARRIER(0);
            if (t + 2 < nt) qdo_prefetch(dsmem + ((t + 2) % 3) * kDkdvSlot, qs, t + 2, nq, wave, lane);
        }
        return;
    }

    // this lane's K / V row (and the seed rows) first, then the Q / dO stream (5 wave-instructions per tile and wave): the
    // seeds are computed under its flight (see gload16)
    const bool seed = seeded && !NPCD_DIAG_NO_SEED;
    const int kclamp = min(key, n - 1);
    u32x4 kraw[4], vraw[4], qeraw[4] = {}, doeraw[4] = {}, oeraw[4] = {};
    uint32_t qx0 = 0, qx1 = 0, dx0 = 0, dx1 = 0;
    float lse_e = 0.f;
    row_bcast_issue(kb + (int64_t)kclamp * p.sn, hh, kraw);
    row_bcast_issue(vb + (int64_t)kclamp * p.sn, hh, vraw);
    if (seed) {
        const E* qrow = qb + (int64_t)(n - 1) * p.sn;
        const E* dorow = dob + (int64_t)(n - 1) * p.osn;
        row_bcast_issue(qrow, hh, qeraw);
        row_bcast_issue(dorow, hh, doeraw);
        row_bcast_issue(static_cast<const E*>(p.out) + b * p.osb + h * p.osh + (int64_t)(n - 1) * p.osn, hh, oeraw);
        lse_e = gload_f32(p.lse + (int64_t)(b * p.H + h) * n + (n - 1));
        qx0 = gload_u16(qrow + r);   qx1 = gload_u16(qrow + 32 + r);
        dx0 = gload_u16(dorow + r);  dx1 = gload_u16(dorow + 32 + r);
    }
    qdo_prefetch(dsmem, qs, 0, nq, wave, lane);
    if (nt > 1) qdo_prefetch(dsmem + kDkdvSlot, qs, 1, nq, wave, lane);
    if (nt > 1) vm_wait<10>();
    else vm_wait<5>();

    V8 kf[4], vf[4];
    arrived4(kraw, kf);
    arrived4(vraw, vf);
    if (!key_ok) {          // keys past the end: zero operands (their rows are never stored)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = V8{0};
            vf[s] = V8{0};
        }
    }
    DkdvState a;
    a.dk0 = f32x16{0}; a.dk1 = f32x16{0}; a.dv0 = f32x16{0}; a.dv1 = f32x16{0};
    float ds_edge = 0.f;
    if (seed) {       // the last query row against this lane's key: P = exp2(c (k.q - lse/scale)), dS = P (v.dO - delta)
        V8 qe[4], doe[4], oe[4];
        arrived4(qeraw, qe);
        arrived4(doeraw, doe);
        arrived4(oeraw, oe);
        NPCD_ARRIVED(lse_e);
        NPCD_ARRIVED(qx0); NPCD_ARRIVED(qx1); NPCD_ARRIVED(dx0); NPCD_ARRIVED(dx1);
        // the row constants of the last query (the dQ pass, in edge mode, has no workgroup that would have written them)
        const float c0 = -lse_e / p.scale, c1 = -mfma_dot<TR>(doe, oe);
        const float p1 = __builtin_amdgcn_exp2f((mfma_dot<TR>(qe, kf) + c0) * p.scale_log2);
        const float ds1 = p1 * (mfma_dot<TR>(doe, vf) + c1);
        ds_edge = ds1;
        outer_seed<TR>(dx0, dx1, p1, lane, a.dv0, a.dv1);
        outer_seed<TR>(qx0, qx1, ds1, lane, a.dk0, a.dk1);
    }
    V8 pf[2], df[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        pf[g] = V8{0};
        df[g] = V8{0};
    }
    const float c = p.scale_log2;
    const FragAddr fa = frag_addr(dsmem, lane);
    const uint32_t st_addr = lds_addr(dsmem) + hh * 16;
    NPCD_TS(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // tiles 0 / 1 have arrived
    NPCD_TS(2);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    NPCD_TS(3);
    // tile 0 (slot 0) is peeled: its first stage has no predecessor to accumulate
    bool both = true;      // the last tile ended with its second sub-block
    {
        const int t = 0;
        dkdv_stage<TR, 0, 0, 2, 1, false>(fa, st_addr, kf, vf, c, a, pf, df);
        NPCD_DKDV_MID();
        if (2 < nt) qdo_prefetch(dsmem + 2 * kDkdvSlot, qs, 2, nq, wave, lane);
        if (32 < nq) dkdv_stage<TR, 0, 1, 0, 0, true>(fa, st_addr, kf, vf, c, a, pf, df);
        else both = false;
        (void)t;
    }
    for (int t = 1; t < nt; t += 3) {
        both = dkdv_step<TR, 1>(dsmem, fa, st_addr, qs, t, nt, nq, wave, lane, kf, vf, c, a, pf, df);
        if (t + 1 < nt) both = dkdv_step<TR, 2>(dsmem, fa, st_addr, qs, t + 1, nt, nq, wave, lane, kf, vf, c, a, pf, df);
        if (t + 2 < nt) both = dkdv_step<TR, 0>(dsmem, fa, st_addr, qs, t + 2, nt, nq, wave, lane, kf, vf, c, a, pf, df);
    }
    NPCD_TS(20);
    {
        const int last = (nt - 1) % 3;
        if (both) {
            if (last == 0) dkdv_flush<TR, 0, 1>(fa, a, pf, df);
            else if (last == 1) dkdv_flush<TR, 1, 1>(fa, a, pf, df);
            else dkdv_flush<TR, 2, 1>(fa, a, pf, df);
        } else {
            if (last == 0) dkdv_flush<TR, 0, 0>(fa, a, pf, df);
            else if (last == 1) dkdv_flush<TR, 1, 0>(fa, a, pf, df);
            else dkdv_flush<TR, 2, 0>(fa, a, pf, df);
        }
    }
    f32x16 eq0 = {0}, eq1 = {0};
    if (edge && !NPCD_DIAG_NO_EDGE_REDUCE) {   // partial dQ_E of this wave's 32 keys (K rows weighted by dS_E), in the slot before the last
        // tile's -- released, like the staging slot below, at the last mid-tile barrier
        const uint32_t img = ((nt + 1) % 3) * kDkdvSlot + wave * 4096, wa = lds_addr(dsmem) + ((nt + 1) % 3) * kDkdvSlot + 16384 + wave * 64;
        edge_put_rows(fa, img, kf);
        if (hh == 0) reinterpret_cast<E*>(dsmem + ((nt + 1) % 3) * kDkdvSlot + 16384 + wave * 64)[edge_w_slot(r)] = (E)ds_edge;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        edge_reduce<TR>(fa, img, wa, hh, eq0, eq1);
    }
    {   // the slot after the last tile's was released at the last mid-tile barrier: 4 KiB of it per wave stage the rows
        unsigned char* stage = dsmem + (nt % 3) * kDkdvSlot + wave * 4096;
        E* gk = static_cast<E*>(p.dk) + b * p.gsb + (int64_t)key0 * p.gsn + h * p.gsh;
        E* gv = static_cast<E*>(p.dv) + b * p.gsb + (int64_t)key0 * p.gsn + h * p.gsh;
        store_rows_staged<TR>(stage, gk, p.gsn, n - key0, a.dk0, a.dk1, p.scale, lane, colsum_seg(p, ((int64_t)b * nkt + kt) * 4 + wave, h, 1));
        store_rows_staged<TR>(stage, gv, p.gsn, n - key0, a.dv0, a.dv1, 1.f, lane, colsum_seg(p, ((int64_t)b * nkt + kt) * 4 + wave, h, 2));
    }
    if (edge && !NPCD_DIAG_NO_EDGE_REDUCE) {
        float* part = p.delta + 2 * (int64_t)p.B * p.H * (((n + 63) >> 6) << 6) + (((int64_t)bh * nkt + kt) * 4 + wave) * kEdgeFloats;
        unsigned char* scratch = dsmem + ((nt + 1) % 3) * kDkdvSlot + wave * 4096;      // the wave's own image, read out by edge_reduce
        edge_stage(scratch, lane, eq0, eq1);
        part[128 + lane] = reinterpret_cast<const float*>(scratch)[lane];
    }
#ifdef NPCD_TIMELINE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NPCD_TS(21);
    if (tl_on && lane == 0)
        for (int i = 0; i < 40; ++i) g_timeline[i] = tl[i];
#endif
}

// ============================================================================================
// backward, ONE pass over the scores (5 products: S, dP, dV, dK, dQ) -- attn_bwd_fused_kernel
// ============================================================================================
// One workgroup = one (batch, head), eight waves, 32 keys per wave: a PASS covers 256 keys and streams every query sub-block
// (32 rows) through the ring of the dK/dV kernel above; sequences longer than 256 keys take several passes, one after the
// other in the same workgroup.  Per sub-block and wave, with the keys on the lanes exactly as in the dK/dV kernel:
// S = Q K^T, dP = dO V^T (row constants as initial accumulators), P, dS, dV^T += dO^T P, dK^T += Q^T dS.  New here: dS is
// written once to LDS as a [key][query] image (8 bytes per lane and 4-row group, 8-byte chunks XOR-swizzled with the key) and
// after the workgroup's barrier every wave computes ONE 16 x 16 tile of dQ^T[64 d][32 q] over ALL 256 keys of the pass with
// v_mfma_f32_16x16x32 (A = K^T from a [key][d] image of the pass's keys, B = dS^T, both by transposed LDS reads): the sum
// over keys is the K dimension of the matrix instruction, no cross-wave fp32 reduction and no atomics.  dQ of a sub-block is
// therefore complete for the pass's keys; across passes it is accumulated in a private fp32 slab of the (batch, head) in global
// memory (same lanes, same addresses in every pass: plain read-modify-write), the last pass scales, rounds and stores it.
// The per-row constants (-lse/scale, -rowsum(dO * O)) are computed by the workgroup itself in a prologue.
// Deterministic: fixed summation orders everywhere.
constexpr int kFWaves = 8;
constexpr int kFStage = kFWaves * 2048;                          // dS^T exchange buffer: [wave][32 keys][32 q] 16-bit
constexpr int kFKimg = kFWaves * 4096;                           // K image: [wave][32 keys][64 d] 16-bit, tile_off swizzle
constexpr int kFLds = 3 * kDkdvSlot + 2 * kFStage + kFKimg;      // 116,224 B: one workgroup per CU

template <class E>
struct FusedStream {
    const E* base;        // q (waves 0..3) or dout (waves 4..7) of this (batch, head), advanced to the wave's first row of a tile
    const E* q;
    const E* dout;
    const float* stat;
    int64_t stride, qstride, dostride;
    DmaLane dl;
    int dst, stat_dst, w4;
    bool second;
};
// tile t of the Q / dO stream into ring slot `slot`: eight waves, two 1 KiB pieces each
template <class E>
__device__ __forceinline__ void fused_prefetch(unsigned char* slot, const FusedStream<E>& qs, int t, int n, int lane) {
    const int row0 = t * 64;
    const uint32_t sl = lds_addr(slot);
    if (row0 + 64 <= n) {
        const char* sbase = reinterpret_cast<const char*>(qs.base + (int64_t)row0 * qs.stride);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            dma16_issue(sbase + (int64_t)i * 8 * qs.stride * (int64_t)sizeof(E), qs.dl.off[i & 1],
                        __builtin_amdgcn_readfirstlane(sl + qs.dst + i * 1024));
    } else {
        const E* base = qs.second ? qs.dout : qs.q;
        const int64_t stride = qs.second ? qs.dostride : qs.qstride;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int prow = (qs.w4 * 2 + i) * 8 + (lane >> 3);
            const int grow = min(row0 + prow, n - 1);
            const uint32_t voff = (uint32_t)((grow * stride + (((lane & 7) ^ tile_swz(prow)) << 3)) * (int64_t)sizeof(E));
            dma16_issue(base, voff, __builtin_amdgcn_readfirstlane(sl + qs.dst + i * 1024));
        }
    }
    dma4_issue(qs.stat, (uint32_t)(row0 + lane) * 4u, __builtin_amdgcn_readfirstlane(sl + qs.stat_dst));
}

struct FusedCtx {
    uint32_t stg;          // LDS byte address of the two dS^T exchange buffers
    uint32_t kimg;         // LDS byte address of the K image
    uint32_t wr_off;       // this lane's write offset inside its wave's 2 KiB of an exchange buffer (chunk 0)
    uint32_t a_tr[2];      // transposed-read lane offsets into a 4 KiB K block   (rows 8g + q, 8g + 4 + q; 16 d columns of this wave's tile)
    uint32_t b_tr[2];      // transposed-read lane offsets into a 2 KiB dS^T block (rows 8g + q, 8g + 4 + q; 16 q columns of this wave's tile)
    float* slab;           // fp32 dQ accumulator of this (batch, head): [npad][64]
    void* dq;              // output rows of this (batch, head)
    int64_t gsn;
    int n, nkb, wave, dt, qt;
    bool first_pass, last_pass, key_ok;
    float scale;
};

// slab address of this lane's 4 consecutive d of query row q0 + 16 qt + (lane & 15); nullptr past the end of the sequence
__device__ __forceinline__ float* fused_slab_ptr(const FusedCtx& fc, int q0, int lane) {
    const int q = q0 + 16 * fc.qt + (lane & 15);
    return q < fc.n ? fc.slab + (int64_t)q * 64 + 16 * fc.dt + 4 * (lane >> 4) : nullptr;
}
// dQ^T tile of the sub-block that starts at query row q0 (its dS^T sits in exchange buffer PAR): one MFMA per key block of the
// pass, operands fetched two blocks at a time (8 transposed reads, one wait); `prev` = the tile's running sum from the
// earlier passes (loaded by the caller a stage ago; ignored in the first pass)
template <class TR, int PAR>
__device__ __forceinline__ void fused_dq(const FusedCtx& fc, int q0, f32x4v prev, int lane) {
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < fc.nkb; j0 += 2) {
        u32x2 a0[2], a1[2], b0[2], b1[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int j = min(j0 + jj, fc.nkb - 1);                          // (a block past the end re-reads the last one; its product is dropped)
            const uint32_t ka = fc.kimg + j * 4096, sb = fc.stg + PAR * kFStage + j * 2048;
            a0[jj] = tr_issue_one(ka + fc.a_tr[0]); a1[jj] = tr_issue_one(ka + fc.a_tr[1]);
            b0[jj] = tr_issue_one(sb + fc.b_tr[0]); b1[jj] = tr_issue_one(sb + fc.b_tr[1]);
        }
        tr_wait();
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (j0 + jj < fc.nkb) {
                const u32x4 av = {a0[jj][0], a0[jj][1], a1[jj][0], a1[jj][1]}, bv = {b0[jj][0], b0[jj][1], b1[jj][0], b1[jj][1]};
                if constexpr (std::is_same<typename TR::elem, __bf16>::value)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc, 0, 0, 0);
                else
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), acc, 0, 0, 0);
            }
        }
    }
    const int q = q0 + 16 * fc.qt + (lane & 15), dcol = 16 * fc.dt + 4 * (lane >> 4);
    if (q < fc.n) {
        if (!fc.first_pass) acc += prev;
        if (fc.last_pass) {
            u32x2 o;
            o[0] = pack2<TR>(acc[0] * fc.scale, acc[1] * fc.scale);
            o[1] = pack2<TR>(acc[2] * fc.scale, acc[3] * fc.scale);
            *reinterpret_cast<u32x2*>(static_cast<typename TR::elem*>(fc.dq) + (int64_t)q * fc.gsn + dcol) = o;
        } else {
            *reinterpret_cast<f32x4v*>(fc.slab + (int64_t)q * 64 + dcol) = acc;
        }
    }
}

// One stage = one 32-row query sub-block (SLOT, SUB), cf. dkdv_stage; additionally the dQ tile of the PREVIOUS sub-block
// (its dS^T was published by every wave at the end of the previous stage and the caller has passed a barrier since) and, at
// the end, this sub-block's dS^T into exchange buffer PAR.
template <class TR, int SLOT, int SUB, int PSLOT, int PSUB, bool ACC, int PAR, class PF>
__device__ __forceinline__ void fused_stage(const FragAddr& fa, uint32_t st_addr, const typename TR::vec8 (&kf)[4],
                                            const typename TR::vec8 (&vf)[4], float c, DkdvState& a, typename TR::vec8 (&pf)[2],
                                            typename TR::vec8 (&df)[2], const FusedCtx& fc, int prev_q0, int cur_q0, f32x4v& slab_val,
                                            bool compute, int lane, PF&& prefetch) {
    using V8 = typename TR::vec8;
    constexpr int QT = SLOT * kDkdvSlot + SUB * 4096, DT = QT + 8192, ST = SLOT * kDkdvSlot + 16384 + SUB * 128;
    constexpr int PQ = PSLOT * kDkdvSlot, PD = PQ + 8192;
#if !defined(NPCD_FUSED_ABL) || NPCD_FUSED_ABL < 1
    if (ACC) fused_dq<TR, PAR ^ 1>(fc, prev_q0, slab_val, lane);
#endif
    __builtin_amdgcn_sched_barrier(0);
    // the Q / dO tile two ahead is requested only now: the running dQ sums above are ordinary loads, and an ordinary load's
    // wait (vmcnt(0), the compiler cannot count past hand-issued LDS-DMA) must not fall behind a freshly issued DMA
    prefetch();
#if !defined(NPCD_FUSED_ABL) || NPCD_FUSED_ABL < 1
    if (!fc.first_pass) {                                        // this sub-block's running sum, consumed by the next stage
#else
    if (false) {
#endif
        const float* sp = fused_slab_ptr(fc, cur_q0, lane);
        slab_val = sp ? *reinterpret_cast<const f32x4v*>(sp) : f32x4v{0.f, 0.f, 0.f, 0.f};
    }
    if (!compute) return;                                       // wave without keys in this pass (wave-uniform)
    u32x4 si[4], di[4], qr[4], dr[4];
    si[0] = lds_b128_issue<ST>(st_addr);       si[1] = lds_b128_issue<ST + 32>(st_addr);
    si[2] = lds_b128_issue<ST + 64>(st_addr);  si[3] = lds_b128_issue<ST + 96>(st_addr);
    qr[0] = lds_b128_issue<QT>(fa.row[0]);     qr[1] = lds_b128_issue<QT>(fa.row[1]);
    qr[2] = lds_b128_issue<QT>(fa.row[2]);     qr[3] = lds_b128_issue<QT>(fa.row[3]);
    di[0] = lds_b128_issue<ST + 256>(st_addr); di[1] = lds_b128_issue<ST + 288>(st_addr);
    di[2] = lds_b128_issue<ST + 320>(st_addr); di[3] = lds_b128_issue<ST + 352>(st_addr);
    dr[0] = lds_b128_issue<DT>(fa.row[0]);     dr[1] = lds_b128_issue<DT>(fa.row[1]);
    dr[2] = lds_b128_issue<DT>(fa.row[2]);     dr[3] = lds_b128_issue<DT>(fa.row[3]);
    tr_wait();
    f32x16 s, d;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            s[4 * g + bq] = __uint_as_float(si[g][bq]);
            d[4 * g + bq] = __uint_as_float(di[g][bq]);
        }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        s = TR::mfma32(__builtin_bit_cast(V8, qr[ks]), kf[ks], s);
        d = TR::mfma32(__builtin_bit_cast(V8, dr[ks]), vf[ks], d);
    }
    TrPair to[2][2], tq[2][2];
    if (ACC) {
        __builtin_amdgcn_sched_barrier(0);
        to[0][0] = tr_issue_at<PD, 2 * PSUB>(fa, 0);     to[0][1] = tr_issue_at<PD, 2 * PSUB>(fa, 1);
        to[1][0] = tr_issue_at<PD, 2 * PSUB + 1>(fa, 0); to[1][1] = tr_issue_at<PD, 2 * PSUB + 1>(fa, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 2; ks < 4; ++ks) {
        s = TR::mfma32(__builtin_bit_cast(V8, qr[ks]), kf[ks], s);
        d = TR::mfma32(__builtin_bit_cast(V8, dr[ks]), vf[ks], d);
    }
    u32x4 pw[2], dw[2];
    auto softmax_half = [&](int h2) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = 4 * h2 + jj;
            const float p0 = __builtin_amdgcn_exp2f(s[2 * j] * c), p1 = __builtin_amdgcn_exp2f(s[2 * j + 1] * c);
            pw[h2][jj] = pack2<TR>(p0, p1);
            dw[h2][jj] = pack2<TR>(p0 * d[2 * j], p1 * d[2 * j + 1]);
        }
    };
    if (ACC) {
        tr_wait();
        a.dv0 = TR::mfma32(tr_vec<TR>(to[0][0]), pf[0], a.dv0);
        a.dv1 = TR::mfma32(tr_vec<TR>(to[0][1]), pf[0], a.dv1);
        a.dv0 = TR::mfma32(tr_vec<TR>(to[1][0]), pf[1], a.dv0);
        a.dv1 = TR::mfma32(tr_vec<TR>(to[1][1]), pf[1], a.dv1);
        __builtin_amdgcn_sched_barrier(0);
        tq[0][0] = tr_issue_at<PQ, 2 * PSUB>(fa, 0);     tq[0][1] = tr_issue_at<PQ, 2 * PSUB>(fa, 1);
        tq[1][0] = tr_issue_at<PQ, 2 * PSUB + 1>(fa, 0); tq[1][1] = tr_issue_at<PQ, 2 * PSUB + 1>(fa, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    softmax_half(0);
    if (ACC) {
        tr_wait();
        a.dk0 = TR::mfma32(tr_vec<TR>(tq[0][0]), df[0], a.dk0);
        a.dk1 = TR::mfma32(tr_vec<TR>(tq[0][1]), df[0], a.dk1);
        a.dk0 = TR::mfma32(tr_vec<TR>(tq[1][0]), df[1], a.dk0);
        a.dk1 = TR::mfma32(tr_vec<TR>(tq[1][1]), df[1], a.dk1);
    }
    softmax_half(1);
    pf[0] = __builtin_bit_cast(V8, pw[0]);
    pf[1] = __builtin_bit_cast(V8, pw[1]);
    df[0] = __builtin_bit_cast(V8, dw[0]);
    df[1] = __builtin_bit_cast(V8, dw[1]);
    // publish dS^T of this sub-block: lane = key, registers 4g..4g+3 = four consecutive query rows 8g + 4hh .. -> 8 bytes at
    // [key][q = 8g + 4hh], 8-byte chunk index XOR (key & 7); keys past the end of the sequence publish zeros (their P is not
    // zero -- harmless for dK / dV, whose rows are never stored, but it would leak into every query's dQ)
#if !defined(NPCD_FUSED_ABL) || NPCD_FUSED_ABL < 2
    {
#else
    if (false) {
#endif
        const uint32_t base = fc.stg + PAR * kFStage + fc.wave * 2048 + fc.wr_off;
        const int key = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x2 v = {dw[g >> 1][2 * (g & 1)], dw[g >> 1][2 * (g & 1) + 1]};
            if (!fc.key_ok) v = u32x2{0u, 0u};
            const uint32_t addr = base + ((((2 * g + hh) ^ (key & 7)) & 7) << 3);
            asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
        }
    }
}

template <class TR>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(AttnParams p, float* dq_slab) {
    using E = typename TR::elem;
    using V8 = typename TR::vec8;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    unsigned char* stg = dsmem + 3 * kDkdvSlot;
    unsigned char* kimg = stg + 2 * kFStage;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = p.n, nt = (n + 63) >> 6, npad = nt * 64;
    const int bh = blockIdx.x, h = bh % p.H, b = bh / p.H;
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
    const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
    const E* ob = static_cast<const E*>(p.out) + b * p.osb + h * p.osh;
    const E* dob = static_cast<const E*>(p.dout) + b * p.osb + h * p.osh;
    float* plane0 = p.delta + (int64_t)bh * npad;
    float* plane1 = p.delta + ((int64_t)p.B * p.H + bh) * npad;

    // ---- prologue: the per-row constants of every query of this (batch, head): eight lanes per row, 16 bytes each
    for (int row0 = wave * 8; row0 < npad; row0 += 8 * kFWaves) {
        const int row = row0 + (lane >> 3), chunk = lane & 7;
        float part = 0.f;
        if (row < n) {
            const V8 dv = *reinterpret_cast<const V8*>(dob + (int64_t)row * p.osn + chunk * 8);
            const V8 ov = *reinterpret_cast<const V8*>(ob + (int64_t)row * p.osn + chunk * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) part += (float)dv[j] * (float)ov[j];
        }
        part += __shfl_xor(part, 1, 64);
        part += __shfl_xor(part, 2, 64);
        part += __shfl_xor(part, 4, 64);
        if (chunk == 0) {
            plane0[row] = row < n ? -p.lse[(int64_t)bh * n + row] / p.scale : -INFINITY;
            plane1[row] = row < n ? -part : 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    FusedStream<E> qs;
    {
        qs.second = wave >= 4;
        qs.w4 = wave & 3;
        qs.q = qb; qs.dout = dob; qs.qstride = p.sn; qs.dostride = p.osn;
        qs.stride = qs.second ? p.osn : p.sn;
        qs.base = (qs.second ? dob : qb) + (int64_t)(qs.w4 * 16) * qs.stride;
        qs.dl = dma_lane<E>(qs.stride, lane);
        qs.dst = (qs.second ? 8192 : 0) + qs.w4 * 2048;
        qs.stat = (wave & 1) ? plane1 : plane0;
        qs.stat_dst = 16384 + (wave & 1) * 256;
    }
    const FragAddr fa = frag_addr(dsmem, lane);
    const uint32_t st_addr = lds_addr(dsmem) + hh * 16;
    FusedCtx fc;
    fc.stg = lds_addr(stg);
    fc.kimg = lds_addr(kimg);
    fc.wave = wave;
    fc.dt = wave & 3;
    fc.qt = wave >> 2;
    fc.n = n;
    fc.scale = p.scale;
    fc.slab = dq_slab + (int64_t)bh * npad * 64;
    fc.dq = static_cast<E*>(p.dq) + b * p.gsb + h * p.gsh;
    fc.gsn = p.gsn;
    fc.wr_off = (uint32_t)((lane & 31) * 64);
    {   // transposed reads (ds_read_b64_tr_b16): 16-lane group g = lane >> 4 supplies rows 8g + (0..3) [second read: + 4], lane 4q' + pp
        // of a group the address of row q', columns 4pp .. 4pp + 3 of the group's 16 columns
        const int g = lane >> 4, li = lane & 15, q4 = li >> 2, pp = li & 3;
#pragma unroll
        for (int hi = 0; hi < 2; ++hi) {
            const int krow = 8 * g + 4 * hi + q4;                                   // key inside a 32-key block
            const int dcol = 16 * fc.dt + 4 * pp;                                   // K image: [key][64 d], tile_off swizzle of 16-byte chunks
            fc.a_tr[hi] = (uint32_t)(tile_off(krow, dcol >> 3) + (dcol & 7) * 2);
            const int qc8 = (16 * fc.qt + 4 * pp) >> 2;                             // dS^T image: [key][32 q], 8-byte chunk XOR (key & 7)
            fc.b_tr[hi] = (uint32_t)(krow * 64 + (((qc8 ^ (krow & 7)) & 7) << 3));
        }
    }
    const float c = p.scale_log2;
    const int npass = (n + 255) >> 8;
    for (int pass = 0; pass < npass; ++pass) {
        const int key0 = pass * 256 + wave * 32;
        const bool wave_active = key0 < n;
        const int key = key0 + r;
        const bool key_ok = key < n;
        fc.key_ok = key_ok;
        fc.nkb = min(kFWaves, (n - pass * 256 + 31) >> 5);
        fc.first_pass = pass == 0;
        fc.last_pass = pass == npass - 1;
        V8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32x4 z = {0, 0, 0, 0};
            kf[s] = __builtin_bit_cast(V8, key_ok ? *reinterpret_cast<const u32x4*>(kb + (int64_t)key * p.sn + 16 * s + 8 * hh) : z);
            vf[s] = __builtin_bit_cast(V8, key_ok ? *reinterpret_cast<const u32x4*>(vb + (int64_t)key * p.sn + 16 * s + 8 * hh) : z);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(kf[s]), "+v"(vf[s]));
        __builtin_amdgcn_s_barrier();                             // previous pass (ring, images, exchange buffers) is finished everywhere
        asm volatile("" ::: "memory");
        if (wave < fc.nkb) {
#pragma unroll
            for (int s = 0; s < 4; ++s) *reinterpret_cast<V8*>(kimg + wave * 4096 + tile_off(r, 2 * s + hh)) = kf[s];
        }
        fused_prefetch(dsmem, qs, 0, n, lane);
        if (nt > 1) fused_prefetch(dsmem + kDkdvSlot, qs, 1, n, lane);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");

        DkdvState a;
        a.dk0 = f32x16{0}; a.dk1 = f32x16{0}; a.dv0 = f32x16{0}; a.dv1 = f32x16{0};
        V8 pf[2], df[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) { pf[g] = V8{0}; df[g] = V8{0}; }
        // stage sequence: sub-block i = 2t + sub lives in ring slot t % 3, its dS^T in exchange buffer i & 1
        bool both = true;
#define NPCD_F_STAGE(SLOT, SUB, PSLOT, PSUB, ACCV, PARV, PQ0, CQ0, PREF) \
    fused_stage<TR, SLOT, SUB, PSLOT, PSUB, ACCV, PARV>(fa, st_addr, kf, vf, c, a, pf, df, fc, PQ0, CQ0, slab_val, wave_active, lane, PREF)
        f32x4v slab_val = {0.f, 0.f, 0.f, 0.f};
        auto nopf = [] {};
        {
            NPCD_F_STAGE(0, 0, 2, 1, false, 0, 0, 0, nopf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's dS^T writes have reached the LDS
            NPCD_DMA_WAIT_BARRIER(0);                             // tile 1 landed; dS^T of sub-block 0 published
            if (32 < n) NPCD_F_STAGE(0, 1, 0, 0, true, 1, 0, 32, [&] { if (2 < nt) fused_prefetch(dsmem + 2 * kDkdvSlot, qs, 2, n, lane); });
            else both = false;
        }
        auto step = [&](auto slot_c, int t) {
            constexpr int SLOT = decltype(slot_c)::value, PREV = (SLOT + 2) % 3;
#if !defined(NPCD_FUSED_ABL) || NPCD_FUSED_ABL < 3
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // dS^T of the previous tile's second sub-block published
            asm volatile("" ::: "memory");
#endif
            NPCD_F_STAGE(SLOT, 0, PREV, 1, true, 0, (t - 1) * 64 + 32, t * 64, nopf);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            NPCD_DMA_WAIT_BARRIER(0);                             // tile t+1 landed; every wave done with tile t-1; dS^T published
            if (t * 64 + 32 >= n) {
                if (t + 2 < nt) fused_prefetch(dsmem + PREV * kDkdvSlot, qs, t + 2, n, lane);
                return false;
            }
            NPCD_F_STAGE(SLOT, 1, SLOT, 0, true, 1, t * 64, t * 64 + 32, [&] { if (t + 2 < nt) fused_prefetch(dsmem + PREV * kDkdvSlot, qs, t + 2, n, lane); });
            return true;
        };
        for (int t = 1; t < nt; t += 3) {
            both = step(std::integral_constant<int, 1>{}, t);
            if (t + 1 < nt) both = step(std::integral_constant<int, 2>{}, t + 1);
            if (t + 2 < nt) both = step(std::integral_constant<int, 0>{}, t + 2);
        }
#undef NPCD_F_STAGE
        // ---- flush: the last sub-block's dV / dK products and its dQ tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const int last = (nt - 1) % 3;
            const int lq0 = (nt - 1) * 64 + (both ? 32 : 0);
            if (both) fused_dq<TR, 1>(fc, lq0, slab_val, lane);
            else fused_dq<TR, 0>(fc, lq0, slab_val, lane);
            if (wave_active) {
                if (both) {
                    if (last == 0) dkdv_flush<TR, 0, 1>(fa, a, pf, df);
                    else if (last == 1) dkdv_flush<TR, 1, 1>(fa, a, pf, df);
                    else dkdv_flush<TR, 2, 1>(fa, a, pf, df);
                } else {
                    if (last == 0) dkdv_flush<TR, 0, 0>(fa, a, pf, df);
                    else if (last == 1) dkdv_flush<TR, 1, 0>(fa, a, pf, df);
                    else dkdv_flush<TR, 2, 0>(fa, a, pf, df);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                             // every wave has left the images: the K image stages the gradient rows
        asm volatile("" ::: "memory");
        if (wave_active) {
            unsigned char* stage = kimg + wave * 4096;
            E* gk = static_cast<E*>(p.dk) + b * p.gsb + (int64_t)key0 * p.gsn + h * p.gsh;
            E* gv = static_cast<E*>(p.dv) + b * p.gsb + (int64_t)key0 * p.gsn + h * p.gsh;
            store_rows_staged<TR>(stage, gk, p.gsn, n - key0, a.dk0, a.dk1, p.scale, lane);
            store_rows_staged<TR>(stage, gv, p.gsn, n - key0, a.dv0, a.dv1, 1.f, lane);
        }
    }
}

// ============================================================================================
// fp32 forward (sampling / fp32 inference path: the reference runs DiffusionModel.generate in fp32 with
// the einsum attention, diffusion_model.py:108-133, npcd.py:8 use_flash_attn=False).  Exact fp32 math on
// the vector ALU: no 16-bit rounding anywhere.  One wave = 4 query rows, workgroup = 16 rows; 64-key K/V
// tiles in LDS (rows padded to 65 floats: the per-lane row reads are conflict-free); pass 1 has the key on
// the lane (scores), pass 2 the feature dimension on the lane (P.V with P broadcast from LDS).
// ============================================================================================
// Round 6: templated over the head dim (32 / 64 / 128) -- the fp32 sampler (the reference's default sampling precision) then covers
// every head dim the 16-bit kernels take.  D = 128 needs 77 KB of LDS (dynamic); pass 2 gives a lane the features lane, lane + 64.
template <int D>
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           float* __restrict__ out, float* __restrict__ lse, int B, int n, int H, int64_t sb,
                                                           int64_t sn, int64_t sh, int64_t osb, int64_t osn, int64_t osh, float scale) {
    constexpr int PAD = D + 1, DF = (D + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char f32smem[];
    float* Ks = reinterpret_cast<float*>(f32smem);      // [64][D + 1]
    float* Vs = Ks + 64 * PAD;                          // [64][D]
    float* Qs = Vs + 64 * D;                            // [16][D]
    float* Ps = Qs + 16 * D;                            // [16][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nqt = (n + 15) / 16;
    const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt, h = bh % H, b = bh / H;
    const float* qb = q + b * sb + h * sh;
    const float* kb = k + b * sb + h * sh;
    const float* vb = v + b * sb + h * sh;
    for (int i = tid; i < 16 * D; i += 256) {
        const int row = min(qt * 16 + i / D, n - 1);
        Qs[i] = qb[row * sn + (i % D)] * scale;
    }
    float m[4], l[4], acc[4][DF];
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        m[rq] = -INFINITY; l[rq] = 0.f;
#pragma unroll
        for (int f = 0; f < DF; ++f) acc[rq][f] = 0.f;
    }
    const int nt = (n + 63) / 64;
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        for (int i = tid; i < 64 * D; i += 256) {
            const int row = min(t * 64 + i / D, n - 1);
            Ks[(i / D) * PAD + (i % D)] = kb[row * sn + (i % D)];
            Vs[i] = vb[row * sn + (i % D)];
        }
        __syncthreads();
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        const float* krow = Ks + lane * PAD;
        const float* qrow = Qs + wave * 4 * D;
#pragma unroll 8
        for (int d = 0; d < D; ++d) {
            const float kv = krow[d];
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) s[rq] = fmaf(qrow[rq * D + d], kv, s[rq]);
        }
        const bool key_ok = t * 64 + lane < n;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const float sc = key_ok ? s[rq] : -INFINITY;
            float mx = sc;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
            const float mn = fmaxf(m[rq], mx);
            const float alpha = expf(m[rq] - mn);
            const float pe = expf(sc - mn);
            float ps = pe;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) ps += __shfl_xor(ps, off, 64);
            l[rq] = l[rq] * alpha + ps;
#pragma unroll
            for (int f = 0; f < DF; ++f) acc[rq][f] *= alpha;
            m[rq] = mn;
            Ps[(wave * 4 + rq) * 64 + lane] = pe;
        }
        // same-wave LDS write -> read: ordered within the wave
        const float* prow = Ps + wave * 4 * 64;
#pragma unroll
        for (int f = 0; f < DF; ++f) {
            const int col = lane + 64 * f;
            if (col < D) {
#pragma unroll 8
                for (int kk = 0; kk < 64; ++kk) {
                    const float vv = Vs[kk * D + col];
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) acc[rq][f] = fmaf(prow[rq * 64 + kk], vv, acc[rq][f]);
                }
            }
        }
    }
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
        const int row = qt * 16 + wave * 4 + rq;
#pragma unroll
        for (int f = 0; f < DF; ++f) {
            const int col = lane + 64 * f;
            if (row < n && col < D) out[b * osb + row * osn + h * osh + col] = acc[rq][f] / l[rq];
        }
        if (lse && row < n && lane == 0) lse[(int64_t)(b * H + h) * n + row] = m[rq] + logf(l[rq]);     // natural log, scaled scores
    }
}
template <int D>
static int launch_f32_valu(const float* q, const float* k, const float* v, float* out, float* lse, int B, int n, int H, int64_t sb, int64_t sn,
                           int64_t sh, int64_t osb, int64_t osn, int64_t osh, float scale, hipStream_t st) {
    constexpr size_t lds = (size_t)(64 * (D + 1) + 64 * D + 16 * D + 16 * 64) * sizeof(float);
    static DynLds attr;
    if (lds > 65536) NPCD_HIP_CHECK(attr.ensure(reinterpret_cast<const void*>(attn_fwd_f32_kernel<D>), lds));
    hipLaunchKernelGGL(attn_fwd_f32_kernel<D>, dim3(B * H * ceil_div(n, 16)), dim3(256), lds, st, q, k, v, out, lse, B, n, H, sb, sn, sh, osb, osn,
                       osh, scale);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

// ============================================================================================
// fp32 backward (`--dtype float32` training: the reference differentiates its fp32 einsum attention, transformer.py:76-81).
// fp32 operands on the fp32 matrix instruction v_mfma_f32_32x32x2_f32 (lane (i, g) = (l & 31, l >> 5) gives A[i][g] and B[g][i],
// the accumulator register r of lane (j, g) is C[acc_row(r, g)][j]); nothing is rounded to 16 bits.  Two kernels, as in the
// 16-bit path, both with the OWNED index on the lanes (accumulator columns) and the index that is summed over on the rows:
//   dq kernel    : a wave owns 32 query rows (q and dO fragments in registers, their own q on the lane), loops over 32-key tiles
//                  in LDS:  S^T = K Q^T,  dP^T = V dO^T,  dS^T = P^T (dP^T - delta) scale,  dQ^T += K^T dS^T;
//   dk/dv kernel : a wave owns 32 keys (k and v fragments in registers), loops over 32-query tiles:  S = Q K^T,  dP = dO V^T,
//                  dV^T += dO^T P,  dK^T += Q^T dS.
// The contraction order inside a product is free, which removes every transposition: (1) over the 64 features, lane half g takes
// features 8 u + 4 g + e (u = 0..7, e = 0..3): the register-resident fragment is eight float4 loads of the lane's own row, the LDS
// side one ds_read_b128 per u (rows padded to 68 floats: conflict-free); (2) over the 32 rows of a score block, step r pairs the
// two rows acc_row(r, 0), acc_row(r, 1) -- exactly what the lanes of the two halves hold in accumulator register r, so P / dS
// are the B operand as they stand, and the A side reads row acc_row(r, g) of the LDS tile with the feature on the lane.
// delta = rowsum(dO . O) is formed by the dq kernel (its rows' dO are in registers) and handed to the dk/dv kernel through
// `delta`.  Deterministic: no atomics, fixed summation order.
// ============================================================================================
constexpr int kB32St = 68;   // floats per LDS row of a 32 x 64 tile

struct AttnF32Bwd {
    const float *q, *k, *v, *o, *dout, *lse;
    float *dq, *dk, *dv, *delta;
    int B, n, H;
    int64_t sb, sn, sh, osb, osn, osh, gsb, gsn, gsh;
    float scale;
};

// 32 rows x 64 floats from row0 of a [n][64] strided matrix into LDS (rows past n: zero), 256 threads
__device__ __forceinline__ void f32_tile_load(float* dst, const float* base, int row0, int n, int64_t sn, int tid) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = tid + it * 256, row = c >> 4, col = (c & 15) * 4;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + row < n) x = *reinterpret_cast<const float4*>(base + (int64_t)(row0 + row) * sn + col);
        *reinterpret_cast<float4*>(dst + row * kB32St + col) = x;
    }
}
// the lane's own row as the register-resident operand: features 8 u + 4 g + e
__device__ __forceinline__ void f32_row_frag(float (&f)[32], const float* row, bool ok, int g) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) x = *reinterpret_cast<const float4*>(row + 8 * u + 4 * g);
        f[4 * u + 0] = x.x; f[4 * u + 1] = x.y; f[4 * u + 2] = x.z; f[4 * u + 3] = x.w;
    }
}
// C[tile row][lane's row] += sum over the 64 features of tile[row][.] * frag[.]
__device__ __forceinline__ f32x16 f32_scores(const float* tile, const float (&frag)[32], int i, int g) {
    f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float4 a = *reinterpret_cast<const float4*>(tile + i * kB32St + 8 * u + 4 * g);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, frag[4 * u + 0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, frag[4 * u + 1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, frag[4 * u + 2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, frag[4 * u + 3], c, 0, 0, 0);
    }
    return c;
}
// acc[blk][feature 32 blk + .][lane's row] += sum over the 32 tile rows of tile[row][feature] * w[row][lane's row]
__device__ __forceinline__ void f32_accumulate(f32x16 (&acc)[2], const float* tile, const f32x16& w, int i, int g) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float* rowp = tile + acc_row(r, g) * kB32St + i;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(rowp[0], w[r], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(rowp[32], w[r], acc[1], 0, 0, 0);
    }
}
// the lane's row of a [n][64] gradient from the transposed accumulators: features 32 blk + 8 (r >> 2) + 4 g + (0..3)
__device__ __forceinline__ void f32_store_row(float* row, const f32x16 (&acc)[2], int g) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
            *reinterpret_cast<float4*>(row + 32 * blk + 8 * r4 + 4 * g) =
                make_float4(acc[blk][4 * r4], acc[blk][4 * r4 + 1], acc[blk][4 * r4 + 2], acc[blk][4 * r4 + 3]);
}

// fp32 forward on the same instruction, same layout as the dq kernel: a wave owns 32 query rows (q on the lane), S^T = K Q^T per
// 32-key tile, online softmax down the accumulator COLUMN (16 registers + the other lane half: the running maximum and sum of a
// query are one scalar per lane), O^T += V^T P^T with P^T as the B operand as it stands.  Exact fp32, deterministic.
__global__ __launch_bounds__(256) void attn_fwd_f32_mfma_kernel(AttnF32Bwd p) {
    __shared__ float Ks[32 * kB32St];
    __shared__ float Vs[32 * kB32St];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, g = lane >> 5;
    const int n = p.n, nqt = (n + 127) / 128;
    const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt, h = bh % p.H, b = bh / p.H;
    const int qrow = qt * 128 + wave * 32 + i;
    const bool ok = qrow < n;
    float qf[32];
    f32_row_frag(qf, p.q + b * p.sb + (int64_t)qrow * p.sn + h * p.sh, ok, g);
    const float c = p.scale * kLog2e;
    float m = -INFINITY, l = 0.f;          // l: this lane half's share of the row sum
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    const float* kb = p.k + b * p.sb + h * p.sh;
    const float* vb = p.v + b * p.sb + h * p.sh;
    const int nt = (n + 31) / 32;
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        f32_tile_load(Ks, kb, t * 32, n, p.sn, tid);
        f32_tile_load(Vs, vb, t * 32, n, p.sn, tid);
        __syncthreads();
        f32x16 st = f32_scores(Ks, qf, i, g);             // S^T[key][q]
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = (t * 32 + acc_row(r, g) < n) ? st[r] * c : -INFINITY;
            mx = fmaxf(mx, st[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m, mx);                    // finite from the first tile on (key 0 exists)
        const float alpha = exp2f(m - mn);
        m = mn;
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            st[r] = exp2f(st[r] - mn);
            ps += st[r];
        }
        l = fmaf(l, alpha, ps);
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] *= alpha; acc[1][r] *= alpha; }
        f32_accumulate(acc, Vs, st, i, g);                // O^T[d][q] += V[key][d] P^T[key][q]
    }
    l += __shfl_xor(l, 32, 64);
    if (ok) {
        const float inv = 1.f / l;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[0][r] *= inv; acc[1][r] *= inv; }
        f32_store_row(p.dq + b * p.gsb + (int64_t)qrow * p.gsn + h * p.gsh, acc, g);
        if (p.delta && g == 0) p.delta[(int64_t)bh * n + qrow] = (m + log2f(l)) * kLn2;      // lse, natural log of sum exp(scaled scores)
    }
}

__global__ __launch_bounds__(256) void attn_bwd_f32_dq_kernel(AttnF32Bwd p) {
    __shared__ float Ks[32 * kB32St];
    __shared__ float Vs[32 * kB32St];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, g = lane >> 5;
    const int n = p.n, nqt = (n + 127) / 128;
    const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt, h = bh % p.H, b = bh / p.H;
    const int qrow = qt * 128 + wave * 32 + i;
    const bool ok = qrow < n;
    float qf[32], gf[32];
    f32_row_frag(qf, p.q + b * p.sb + (int64_t)qrow * p.sn + h * p.sh, ok, g);
    f32_row_frag(gf, p.dout + b * p.osb + (int64_t)qrow * p.osn + h * p.osh, ok, g);
    float delta = 0.f;
    {
        float of[32];
        f32_row_frag(of, p.o + b * p.osb + (int64_t)qrow * p.osn + h * p.osh, ok, g);
#pragma unroll
        for (int e = 0; e < 32; ++e) delta = fmaf(gf[e], of[e], delta);
        delta += __shfl_xor(delta, 32, 64);
    }
    const float lse2 = ok ? p.lse[(int64_t)bh * n + qrow] * kLog2e : INFINITY;
    if (ok && g == 0) p.delta[(int64_t)bh * n + qrow] = delta;
    const float c = p.scale * kLog2e;
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    const float* kb = p.k + b * p.sb + h * p.sh;
    const float* vb = p.v + b * p.sb + h * p.sh;
    const int nt = (n + 31) / 32;
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        f32_tile_load(Ks, kb, t * 32, n, p.sn, tid);
        f32_tile_load(Vs, vb, t * 32, n, p.sn, tid);
        __syncthreads();
        const f32x16 st = f32_scores(Ks, qf, i, g);       // S^T[key][q]
        const f32x16 dpt = f32_scores(Vs, gf, i, g);      // dP^T[key][q]
        f32x16 ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool key_ok = t * 32 + acc_row(r, g) < n;
            const float pr = key_ok ? exp2f(fmaf(st[r], c, -lse2)) : 0.f;
            ds[r] = pr * (dpt[r] - delta) * p.scale;
        }
        f32_accumulate(acc, Ks, ds, i, g);                // dQ^T[d][q] += K[key][d] dS^T[key][q]
    }
    if (ok) f32_store_row(p.dq + b * p.gsb + (int64_t)qrow * p.gsn + h * p.gsh, acc, g);
}

__global__ __launch_bounds__(256) void attn_bwd_f32_dkdv_kernel(AttnF32Bwd p) {
    __shared__ float Qs[32 * kB32St];
    __shared__ float Gs[32 * kB32St];
    __shared__ __attribute__((aligned(16))) float Ls[32];
    __shared__ __attribute__((aligned(16))) float Ds[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, g = lane >> 5;
    const int n = p.n, nkt = (n + 127) / 128;
    const int kt = blockIdx.x % nkt, bh = blockIdx.x / nkt, h = bh % p.H, b = bh / p.H;
    const int krow = kt * 128 + wave * 32 + i;
    const bool ok = krow < n;
    float kf[32], vf[32];
    f32_row_frag(kf, p.k + b * p.sb + (int64_t)krow * p.sn + h * p.sh, ok, g);
    f32_row_frag(vf, p.v + b * p.sb + (int64_t)krow * p.sn + h * p.sh, ok, g);
    const float c = p.scale * kLog2e;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[0][r] = 0.f; dk[1][r] = 0.f; dv[0][r] = 0.f; dv[1][r] = 0.f; }
    const float* qb = p.q + b * p.sb + h * p.sh;
    const float* gb = p.dout + b * p.osb + h * p.osh;
    const int nt = (n + 31) / 32;
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        f32_tile_load(Qs, qb, t * 32, n, p.sn, tid);
        f32_tile_load(Gs, gb, t * 32, n, p.osn, tid);
        if (tid < 32) {
            const int row = t * 32 + tid;
            Ls[tid] = row < n ? p.lse[(int64_t)bh * n + row] * kLog2e : INFINITY;      // rows past the end: P = 0
            Ds[tid] = row < n ? p.delta[(int64_t)bh * n + row] : 0.f;
        }
        __syncthreads();
        const f32x16 sc = f32_scores(Qs, kf, i, g);       // S[q][key]
        const f32x16 dp = f32_scores(Gs, vf, i, g);       // dP[q][key]
        f32x16 pr, ds;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const float4 l4 = *reinterpret_cast<const float4*>(Ls + 8 * r4 + 4 * g);
            const float4 d4 = *reinterpret_cast<const float4*>(Ds + 8 * r4 + 4 * g);
            const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * r4 + e;
                pr[r] = exp2f(fmaf(sc[r], c, -lv[e]));
                ds[r] = pr[r] * (dp[r] - dl[e]) * p.scale;
            }
        }
        f32_accumulate(dv, Gs, pr, i, g);                 // dV^T[d][key] += dO[q][d] P[q][key]
        f32_accumulate(dk, Qs, ds, i, g);                 // dK^T[d][key] += Q[q][d] dS[q][key]
    }
    if (ok) {
        f32_store_row(p.dk + b * p.gsb + (int64_t)krow * p.gsn + h * p.gsh, dk, g);
        f32_store_row(p.dv + b * p.gsb + (int64_t)krow * p.gsn + h * p.gsh, dv, g);
    }
}

// ============================================================================================
// host entry points
// ============================================================================================
// ============================================================================================
// forward with fp8 (e4m3) operands on the block-scaled matrix instruction  -- opt-in, BASELINE configs[4] names it
// ============================================================================================
// v_mfma_scale_f32_32x32x64_f8f6f4 contracts 64 elements per instruction at twice the bf16 rate: one instruction per 32 x 32
// score block (head_dim 64) and one per 32 output dimensions and 64 keys.  Operand lane map (tools/probes/mfma_scale_fp8_map.hip,
// exact integer data): lane (i = l & 31, g = l >> 5) holds elements k = 32 g .. 32 g + 31 of row / column i in its eight
// registers; C/D as for every 32 x 32 form; block scales are E8M0 bytes (127 = 1).
//   * attn_fp8_pack_kernel: k -> e4m3 rows [B, H, nk, 64]; v -> e4m3 TRANSPOSED [B, H, 64, nk] so that the second product's
//     A operand (V^T: 32 consecutive keys of one dimension) is a plain row read, with the keys of each 64-key tile permuted
//     into the order in which a lane holds the scores of the first product: lane half g owns keys acc_row(j, g) of the tile's first
//     32-key block (slots j = 0..15) and 32 + acc_row(j - 16, g) (slots 16..31) -- P goes from the accumulators into the B operand
//     without leaving the lane.  (The queries are converted by the forward kernel itself: each row is used by one wave only.)
//   * attn_fwd_fp8_kernel: same decomposition as attn_fwd_kernel (4 waves x 32 queries, 64-key tiles through a 3-slot LDS ring by
//     LDS-DMA, 8 KiB per slot), 4 matrix instructions per tile instead of 16.  P is stored as e4m3(256 P) with the block scale
//     2^-8 (e4m3 has 3 mantissa bits and a smallest normal of 2^-6: the scale moves the softmax tail into range); the running
//     maximum is not deferred (256 P must stay below 448).  A single last key (n = 64 j + 1) seeds the state exactly as in the
//     bf16 kernel, from the bf16 rows.  Output, LSE and the whole backward stay as they are (the backward recomputes P from the
//     bf16 operands against this LSE, like FlashAttention-3's fp8 forward).
// Needs nk = n or n - 1 to be a multiple of 64; other lengths: NPCD_ERR_UNSUPPORTED (the caller uses the bf16 kernel).
typedef int v8i32 __attribute__((ext_vector_type(8)));
constexpr int kF8Slot = 8192;          // K tile 64 x 64 B, then V^T tile 64 x 64 B
__device__ __forceinline__ int f8_swz(int row) { return (row >> 2) & 3; }      // 16-byte chunk XOR of a 64-byte row: conflict-free b128 row reads
__device__ __forceinline__ int f8_key_of_slot(int p) {                        // position p of a permuted 64-key tile -> key inside the tile
    const int g = p >> 5, j = p & 31;
    return (j < 16 ? 0 : 32) + acc_row(j & 15, g);
}
__device__ __forceinline__ uint32_t f8_pack4(float a, float b, float c, float d) {
    int x = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    x = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, x, true);
    return (uint32_t)x;
}

// one workgroup per (batch, head, 64-key tile; nk is a multiple of 64): the tile's k rows, and its v rows transposed + permuted
__global__ __launch_bounds__(256) void attn_fp8_pack_kernel(AttnParams p, unsigned char* k8, unsigned char* v8t, int nk) {
    __shared__ __attribute__((aligned(16))) unsigned char vt[64 * 80];      // [key][64 B + pad]
    const int ntile = nk >> 6;
    const int tile = blockIdx.x % ntile, bh = blockIdx.x / ntile, h = bh % p.H, b = bh / p.H, tid = threadIdx.x;
    const __bf16* kb = static_cast<const __bf16*>(p.k) + b * p.sb + h * p.sh;
    const __bf16* vb = static_cast<const __bf16*>(p.v) + b * p.sb + h * p.sh;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = tid + 256 * it, row = c >> 3, ch = c & 7, tok = tile * 64 + row;     // 8 elements of one token
        {
            const bf16x8 x = *reinterpret_cast<const bf16x8*>(kb + (int64_t)tok * p.sn + 8 * ch);
            u32x2 o = {f8_pack4((float)x[0], (float)x[1], (float)x[2], (float)x[3]), f8_pack4((float)x[4], (float)x[5], (float)x[6], (float)x[7])};
            *reinterpret_cast<u32x2*>(k8 + ((int64_t)bh * nk + tok) * 64 + 8 * ch) = o;
            const bf16x8 y = *reinterpret_cast<const bf16x8*>(vb + (int64_t)tok * p.sn + 8 * ch);
            u32x2 w = {f8_pack4((float)y[0], (float)y[1], (float)y[2], (float)y[3]), f8_pack4((float)y[4], (float)y[5], (float)y[6], (float)y[7])};
            *reinterpret_cast<u32x2*>(vt + row * 80 + 8 * ch) = w;
        }
    }
    __syncthreads();
    {
        const int d = tid >> 2, pq = tid & 3;
        uint32_t o[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t x = 0;
#pragma unroll
            for (int by = 0; by < 4; ++by) x |= (uint32_t)vt[f8_key_of_slot(16 * pq + 4 * w + by) * 80 + d] << (8 * by);
            o[w] = x;
        }
        *reinterpret_cast<u32x4*>(v8t + ((int64_t)bh * 64 + d) * nk + tile * 64 + 16 * pq) = u32x4{o[0], o[1], o[2], o[3]};
    }
}

// LDS-DMA of tile t into ring slot `slot`: 8 pieces of 16 rows x 64 B (K: 0..3, V^T: 4..7), two per wave
__device__ __forceinline__ void f8_dma_tile(unsigned char* smem, int slot, const unsigned char* k8, const unsigned char* v8t, int nk, int t,
                                            int wave, int lane) {
    const uint32_t dst = lds_addr(smem) + slot * kF8Slot;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = 2 * wave + j, pr = (piece & 3) * 16 + (lane >> 2);                    // row inside the K or V^T tile
        const uint32_t lchunk = (uint32_t)((lane & 3) ^ f8_swz(pr));
        if (piece < 4) dma16_issue(k8 + (int64_t)t * 4096, (uint32_t)(pr * 64) + lchunk * 16, __builtin_amdgcn_readfirstlane(dst + piece * 1024));
        else dma16_issue(v8t + (int64_t)t * 64, (uint32_t)pr * (uint32_t)nk + lchunk * 16, __builtin_amdgcn_readfirstlane(dst + piece * 1024));
    }
}

__global__ __launch_bounds__(256, 2) void attn_fwd_fp8_kernel(AttnParams p, const unsigned char* k8, const unsigned char* v8t, int nk) {
    using E = __bf16;
    using V8 = bf16x8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[3 * kF8Slot + 4 * 4096];      // ring + the four output staging areas
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, g = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = p.n, nt = nk >> 6;
    const bool seeded = nk < n;
    const int nqt = (n + 127) >> 7;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int qt = bid % nqt, bh = bid / nqt, h = bh % p.H, b = bh / p.H;
    const int q0 = qt * 128 + wave * 32, qrow = q0 + r, qclamp = min(qrow, n - 1);
    const float c = p.scale_log2;
    const unsigned char* k8b = k8 + (int64_t)bh * nk * 64;
    const unsigned char* v8b = v8t + (int64_t)bh * 64 * nk;
    // prologue loads ahead of the DMA (see gload16): this lane half's 32 dimensions of the query row (e4m3 operand: elements
    // 32 g .. 32 g + 31), and for the seed the row in the bf16 operand order + the last k / v rows
    const E* qb = static_cast<const E*>(p.q) + b * p.sb + h * p.sh;
    u32x4 q32[4], qraw[4] = {}, keraw[4] = {};
#pragma unroll
    for (int s = 0; s < 4; ++s) q32[s] = gload16(qb + (int64_t)qclamp * p.sn + 32 * g + 8 * s);
    uint32_t vx0 = 0, vx1 = 0;
    if (seeded) {
        const E* kb = static_cast<const E*>(p.k) + b * p.sb + h * p.sh;
        const E* vb = static_cast<const E*>(p.v) + b * p.sb + h * p.sh;
        row_bcast_issue(qb + (int64_t)qclamp * p.sn, g, qraw);
        row_bcast_issue(kb + (int64_t)(n - 1) * p.sn, g, keraw);
        vx0 = gload_u16(vb + (int64_t)(n - 1) * p.sn + r);
        vx1 = gload_u16(vb + (int64_t)(n - 1) * p.sn + 32 + r);
    }
    f8_dma_tile(smem, 0, k8b, v8b, nk, 0, wave, lane);
    if (nt > 1) f8_dma_tile(smem, 1, k8b, v8b, nk, 1, wave, lane);
    if (nt > 1) vm_wait<4>();
    else vm_wait<2>();
    v8i32 qf8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        NPCD_ARRIVED(q32[s]);
        const V8 x = __builtin_bit_cast(V8, q32[s]);
        qf8[2 * s] = (int)f8_pack4((float)x[0], (float)x[1], (float)x[2], (float)x[3]);
        qf8[2 * s + 1] = (int)f8_pack4((float)x[4], (float)x[5], (float)x[6], (float)x[7]);
    }
    f32x16 o0 = {0}, o1 = {0};
    float m = -INFINITY, l = 0.f;           // l is kept times 256 (the scale of the stored P)
    if (seeded) {
        V8 qf[4], ke[4];
        arrived4(qraw, qf);
        arrived4(keraw, ke);
        NPCD_ARRIVED(vx0);
        NPCD_ARRIVED(vx1);
        m = mfma_dot<BF16>(ke, qf) * c;
        l = 0.5f * 256.f;
        outer_seed<BF16>(vx0, vx1, 1.f, lane, o0, o1);
    }
    // per-lane LDS addresses inside a slot: row r, this lane half's two 16-byte chunks
    const uint32_t base = lds_addr(smem) + r * 64;
    const uint32_t ad0 = base + (((2 * g) ^ f8_swz(r)) << 4), ad1 = base + (((2 * g + 1) ^ f8_swz(r)) << 4);
    const int unit = 0x7f7f7f7f, pscale = 0x77777777;            // E8M0: 2^0, 2^-8
    for (int t = 0; t < nt; ++t) {
        const int slot = t % 3;
        // tile t has landed (at most tile t + 1 is still in flight); every wave is done with tile t - 1: its slot takes tile t + 2
        if (t + 1 < nt) NPCD_DMA_WAIT_BARRIER(2);
        else NPCD_DMA_WAIT_BARRIER(0);
        if (t + 2 < nt) f8_dma_tile(smem, (t + 2) % 3, k8b, v8b, nk, t + 2, wave, lane);
        const uint32_t so = slot * kF8Slot;
        u32x4 ka[4], va[4];
        ka[0] = lds_b128_issue<0>(ad0 + so);       ka[1] = lds_b128_issue<0>(ad1 + so);
        ka[2] = lds_b128_issue<2048>(ad0 + so);    ka[3] = lds_b128_issue<2048>(ad1 + so);
        va[0] = lds_b128_issue<4096>(ad0 + so);    va[1] = lds_b128_issue<4096>(ad1 + so);
        va[2] = lds_b128_issue<6144>(ad0 + so);    va[3] = lds_b128_issue<6144>(ad1 + so);
        tr_wait();
        const v8i32 k0 = {(int)ka[0][0], (int)ka[0][1], (int)ka[0][2], (int)ka[0][3], (int)ka[1][0], (int)ka[1][1], (int)ka[1][2], (int)ka[1][3]};
        const v8i32 k1 = {(int)ka[2][0], (int)ka[2][1], (int)ka[2][2], (int)ka[2][3], (int)ka[3][0], (int)ka[3][1], (int)ka[3][2], (int)ka[3][3]};
        const f32x16 z = {0};
        const f32x16 s0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k0, qf8, z, 0, 0, 0, unit, 0, unit);
        const f32x16 s1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(k1, qf8, z, 0, 0, 0, unit, 0, unit);
        float mx = fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s1[0], s1[1]));
#pragma unroll
        for (int i = 2; i < 16; i += 2) mx = fmaxf(fmaxf(fmaxf(mx, s0[i]), s0[i + 1]), fmaxf(s1[i], s1[i + 1]));
        mx = half_max(mx) * c;
        if (__any(mx > m)) {
            const float mn = fmaxf(mx, m), alpha = __builtin_amdgcn_exp2f(m - mn);
            m = mn;
            l *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o0[i] *= alpha;
                o1[i] *= alpha;
            }
        }
        const float off = 8.f - m;
        float rs = 0.f;
        v8i32 pf;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[4 * w], c, off)), a1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[4 * w + 1], c, off));
            const float a2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[4 * w + 2], c, off)), a3 = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[4 * w + 3], c, off));
            const float b0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[4 * w], c, off)), b1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[4 * w + 1], c, off));
            const float b2 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[4 * w + 2], c, off)), b3 = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[4 * w + 3], c, off));
            rs += (a0 + a1) + (a2 + a3) + (b0 + b1) + (b2 + b3);
            pf[w] = (int)f8_pack4(a0, a1, a2, a3);
            pf[4 + w] = (int)f8_pack4(b0, b1, b2, b3);
        }
        l += rs;
        const v8i32 v0 = {(int)va[0][0], (int)va[0][1], (int)va[0][2], (int)va[0][3], (int)va[1][0], (int)va[1][1], (int)va[1][2], (int)va[1][3]};
        const v8i32 v1 = {(int)va[2][0], (int)va[2][1], (int)va[2][2], (int)va[2][3], (int)va[3][0], (int)va[3][1], (int)va[3][2], (int)va[3][3]};
        o0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v0, pf, o0, 0, 0, 0, unit, 0, pscale);
        o1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(v1, pf, o1, 0, 0, 0, unit, 0, pscale);
    }
    l = half_sum(l) * (1.f / 256.f);
    if (q0 < n) {
        E* orow0 = static_cast<E*>(p.o_w) + b * p.osb + (int64_t)q0 * p.osn + h * p.osh;
        store_rows_staged<BF16>(smem + 3 * kF8Slot + wave * 4096, orow0, p.osn, n - q0, o0, o1, 1.f / l, lane);
        if (qrow < n && g == 0) p.lse[(int64_t)(b * p.H + h) * n + qrow] = m * kLn2 + logf(l);
    }
}

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

#ifdef NPCD_TIMELINE
extern "C" int npcd_debug_read(long long* out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(npcd::g_timeline), sizeof(long long) * count);
}
#endif

// which form of the forward a sequence length takes (NPCD_ATTN_FWD=32 / 64 forces one; see attn_fwd_launch)
static bool fwd_rows32(int n) {
    const char* form = getenv("NPCD_ATTN_FWD");
    return (form && form[0] != 'r') ? form[0] == '3' : n < 1024;
}
extern "C" int64_t npcd_attn_fwd_workspace_floats(int B, int n, int H) {
    if (B <= 0 || n <= 0 || H <= 0) return -1;
    static const bool rowx32 = getenv("NPCD_ATTN_ROWX32") != nullptr;     // (opt-in: the 32-row form without its fifth workgroup; slower, see attn_fwd_kernel)
    return (rowx_mode(n) && (rowx32 || !fwd_rows32(n))) ? (int64_t)B * H * ((n - 1) / 64) * kRowxFloats : 0;
}

static int attn_fwd_launch(const void* q, const void* k, const void* v, void* out, float* lse, float* workspace, int B, int n, int H, int d,
                           int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                           float scale, int dtype, void* stream) {
    if (dtype == NPCD_F32) {   // exact fp32 path (lse [B, H, n] written when given: natural log of the row sums of exp(scaled scores))
        if (B <= 0 || n <= 0 || H <= 0 || !q || !k || !v || !out) return NPCD_ERR_ARG;
        if (d == 32 || d == 128) {        // (round 6) the exact vector-ALU form covers the other head dims: forward only (sampling / inference)
            hipStream_t sv = static_cast<hipStream_t>(stream);
            const float *fq = static_cast<const float*>(q), *fk = static_cast<const float*>(k), *fv = static_cast<const float*>(v);
            return d == 32 ? launch_f32_valu<32>(fq, fk, fv, static_cast<float*>(out), lse, B, n, H, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh, scale, sv)
                           : launch_f32_valu<128>(fq, fk, fv, static_cast<float*>(out), lse, B, n, H, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh, scale, sv);
        }
        if (d != 64) return NPCD_ERR_UNSUPPORTED;
        static const bool valu_form = getenv("NPCD_ATTN_F32_VALU") != nullptr;
        const bool vec_ok = !(qkv_sb % 4 || qkv_sn % 4 || qkv_sh % 4 || out_sb % 4 || out_sn % 4 || out_sh % 4) && aligned16(q) &&
                            aligned16(k) && aligned16(v) && aligned16(out);
        if (vec_ok && !valu_form) {        // matrix-instruction form (needs 16-byte rows); the vector-ALU form below takes any layout
            AttnF32Bwd a{};
            a.q = static_cast<const float*>(q); a.k = static_cast<const float*>(k); a.v = static_cast<const float*>(v);
            a.dq = static_cast<float*>(out); a.delta = lse;
            a.B = B; a.n = n; a.H = H; a.sb = qkv_sb; a.sn = qkv_sn; a.sh = qkv_sh; a.gsb = out_sb; a.gsn = out_sn; a.gsh = out_sh;
            a.scale = scale;
            hipLaunchKernelGGL(attn_fwd_f32_mfma_kernel, dim3(B * H * ceil_div(n, 128)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
            NPCD_HIP_CHECK(hipGetLastError());
            return NPCD_OK;
        }
        return launch_f32_valu<64>(static_cast<const float*>(q), static_cast<const float*>(k), static_cast<const float*>(v), static_cast<float*>(out),
                                   lse, B, n, H, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh, scale, static_cast<hipStream_t>(stream));
    }
    // head dims 32 and 128 (the reference's attention works for any width / heads, transformer.py:68-84): the kernels of
    // attention_gen.hip; NPCD_ATTN_GEN=1 sends d = 64 there too (tests: the two kernel families against each other)
    const bool force_gen = getenv("NPCD_ATTN_GEN") != nullptr;      // (read at every call, like NPCD_ATTN_FWD)
    const bool gen = d != 64 || force_gen;
    int rc = check_common(B, n, H, gen && attn_gen_supported(d) ? 64 : d, dtype);
    if (rc != NPCD_OK) return rc;
    if (!q || !k || !v || !out || !lse) return NPCD_ERR_ARG;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out)) return NPCD_ERR_ARG;
    if (!strides_ok(qkv_sb, qkv_sn, qkv_sh) || !strides_ok(out_sb, out_sn, out_sh)) return NPCD_ERR_ARG;
    if (gen) return attn_gen_fwd(q, k, v, out, lse, B, n, H, d, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh, scale, dtype, stream);
    AttnParams p{};
    p.q = q; p.k = k; p.v = v; p.o_w = out; p.lse = lse;
    p.B = B; p.n = n; p.H = H;
    p.sb = qkv_sb; p.sn = qkv_sn; p.sh = qkv_sh;
    p.osb = out_sb; p.osn = out_sn; p.osh = out_sh;
    p.scale = scale; p.scale_log2 = scale * kLog2e;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // Two forms of the forward (measured in one process, tools/probes/gpu_dev_fwd_ab.py): 64 query rows per wave wins
    // on long sequences (n = 2049: 597-614 against 640-653 us), 32 rows per wave on short ones, where a workgroup's start-up and
    // wind-down dominate and its finer grid fills the chip better (n = 513: 113 against 117 us).  NPCD_ATTN_FWD=32 / 64 forces one.
    // n = 513 (the denoiser's sequence), OPT-IN NPCD_ATTN_FWD=res: K / V of a (batch, head) resident in LDS, one workgroup per
    // (batch, head) -- attn_fwd_res_kernel.  Built in round 5 as the "different body" for this shape; parity green, and SLOWER in the
    // step: 141 against 120 us (same box, alternating bench runs): with one workgroup per CU nothing computes while a workgroup's
    // 200 KB arrive, the four rounds of workgroups load and compute in phase, and the eight waves of a workgroup run in lockstep
    // behind the per-tile barrier (docs/experiments.md R5.2).
    const char* fwd_form = getenv("NPCD_ATTN_FWD");
    if (fwd_res_shape(n) && fwd_form && fwd_form[0] == 'r') {
        static DynLds lds_bf, lds_f;
        if (dtype == NPCD_BF16) {
            NPCD_HIP_CHECK(lds_bf.ensure(reinterpret_cast<const void*>(attn_fwd_res_kernel<BF16>), kResLds));
            hipLaunchKernelGGL(attn_fwd_res_kernel<BF16>, dim3(B * H), dim3(512), kResLds, st, p);
        } else {
            NPCD_HIP_CHECK(lds_f.ensure(reinterpret_cast<const void*>(attn_fwd_res_kernel<F16>), kResLds));
            hipLaunchKernelGGL(attn_fwd_res_kernel<F16>, dim3(B * H), dim3(512), kResLds, st, p);
        }
        NPCD_HIP_CHECK(hipGetLastError());
        return NPCD_OK;
    }
    if (fwd_rows32(n)) {
        static const bool rowx32 = getenv("NPCD_ATTN_ROWX32") != nullptr;
        p.rowx = (workspace && rowx32 && rowx_mode(n)) ? workspace : nullptr;
        const int grid = B * H * (p.rowx ? (n - 1) / 128 : ceil_div(n, 128));
        if (p.rowx) {
            if (dtype == NPCD_BF16) hipLaunchKernelGGL((attn_fwd_kernel<BF16, true>), dim3(grid), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((attn_fwd_kernel<F16, true>), dim3(grid), dim3(256), 0, st, p);
        } else {
            if (dtype == NPCD_BF16) hipLaunchKernelGGL((attn_fwd_kernel<BF16, false>), dim3(grid), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((attn_fwd_kernel<F16, false>), dim3(grid), dim3(256), 0, st, p);
        }
        if (p.rowx) {
            if (dtype == NPCD_BF16) hipLaunchKernelGGL(attn_fwd_rowx_merge_kernel<__bf16>, dim3(B * H), dim3(64), 0, st, p, (n - 1) / 64);
            else hipLaunchKernelGGL(attn_fwd_rowx_merge_kernel<_Float16>, dim3(B * H), dim3(64), 0, st, p, (n - 1) / 64);
        }
    } else {
        p.rowx = (workspace && rowx_mode(n)) ? workspace : nullptr;
        const int grid = B * H * (p.rowx ? (n - 1) / 256 : ceil_div(n, 256));
        if (dtype == NPCD_BF16) hipLaunchKernelGGL(attn_fwd64_kernel<BF16>, dim3(grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(attn_fwd64_kernel<F16>, dim3(grid), dim3(256), 0, st, p);
        if (p.rowx) {
            if (dtype == NPCD_BF16) hipLaunchKernelGGL(attn_fwd_rowx_merge_kernel<__bf16>, dim3(B * H), dim3(64), 0, st, p, (n - 1) / 64);
            else hipLaunchKernelGGL(attn_fwd_rowx_merge_kernel<_Float16>, dim3(B * H), dim3(64), 0, st, p, (n - 1) / 64);
        }
    }
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int n, int H, int d,
                             int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                             float scale, int dtype, void* stream) {
    return attn_fwd_launch(q, k, v, out, lse, nullptr, B, n, H, d, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh, scale, dtype, stream);
}
extern "C" int npcd_attn_fwd_ws(const void* q, const void* k, const void* v, void* out, float* lse, float* workspace, int B, int n, int H,
                                int d, int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                                float scale, int dtype, void* stream) {
    return attn_fwd_launch(q, k, v, out, lse, workspace, B, n, H, d, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh, scale, dtype, stream);
}

// fp8 forward (opt-in): workspace = the e4m3 copies of k and v^T
static int fp8_nk(int n) { return (n & 63) == 1 && n > 64 ? n - 1 : n; }
extern "C" int64_t npcd_attn_fwd_fp8_workspace_bytes(int B, int n, int H) {
    if (B <= 0 || n <= 0 || H <= 0) return -1;
    const int nk = fp8_nk(n);
    if (nk % 64 != 0) return 0;              // length not covered: use npcd_attn_fwd
    return (int64_t)B * H * 64 * 2 * (int64_t)nk;
}
extern "C" int npcd_attn_fwd_fp8(const void* q, const void* k, const void* v, void* out, float* lse, void* workspace, int B, int n, int H,
                                 int d, int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                                 float scale, int dtype, void* stream) {
    int rc = npcd::check_common(B, n, H, d, dtype);
    if (rc != NPCD_OK) return rc;
    if (dtype != NPCD_BF16) return NPCD_ERR_UNSUPPORTED;
    const int nk = fp8_nk(n);
    if (nk % 64 != 0) return NPCD_ERR_UNSUPPORTED;
    if (!q || !k || !v || !out || !lse || !workspace) return NPCD_ERR_ARG;
    if (!npcd::aligned16(q) || !npcd::aligned16(k) || !npcd::aligned16(v) || !npcd::aligned16(out) || !npcd::aligned16(workspace)) return NPCD_ERR_ARG;
    if (!npcd::strides_ok(qkv_sb, qkv_sn, qkv_sh) || !npcd::strides_ok(out_sb, out_sn, out_sh)) return NPCD_ERR_ARG;
    npcd::AttnParams p{};
    p.q = q; p.k = k; p.v = v; p.o_w = out; p.lse = lse;
    p.B = B; p.n = n; p.H = H;
    p.sb = qkv_sb; p.sn = qkv_sn; p.sh = qkv_sh;
    p.osb = out_sb; p.osn = out_sn; p.osh = out_sh;
    p.scale = scale; p.scale_log2 = scale * npcd::kLog2e;
    unsigned char* k8 = static_cast<unsigned char*>(workspace);
    unsigned char* v8t = k8 + (int64_t)B * H * nk * 64;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(npcd::attn_fp8_pack_kernel, dim3(B * H * (nk / 64)), dim3(256), 0, st, p, k8, v8t, nk);
    hipLaunchKernelGGL(npcd::attn_fwd_fp8_kernel, dim3(B * H * npcd::ceil_div(n, 128)), dim3(256), 0, st, p, k8, v8t, nk);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

// The three gradient rows of the edge token E = n - 1: partials of the waves of the (n - 1) / 128 workgroups of each pass, added in
// (workgroup, wave) order, plus the one pair neither pass has seen: query E against key E (fp32 from the rows themselves).  One wave per row.
template <class E>
__global__ __launch_bounds__(192) void attn_bwd_edge_kernel(AttnParams p) {
    const int bh = blockIdx.x, h = bh % p.H, b = bh / p.H, mat = threadIdx.x >> 6, d = threadIdx.x & 63;
    const int nb = (p.n >> 7) * 4, last = p.n - 1;          // one partial per wave of the (n - 1) / 128 workgroups
    const float* part = p.delta + 2 * (int64_t)p.B * p.H * (((p.n + 63) >> 6) << 6) + (int64_t)bh * nb * kEdgeFloats + mat * 64 + d;
    float acc = 0.f;
    int j = 0;
    for (; j + 16 <= nb; j += 16) {         // loads first (independent), then the adds in index order
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = part[(int64_t)(j + u) * kEdgeFloats];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
    }
    for (; j < nb; j += 4) {                // nb is a multiple of 4
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = part[(int64_t)(j + u) * kEdgeFloats];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    const int64_t qkv_at = b * p.sb + (int64_t)last * p.sn + h * p.sh + d, o_at = b * p.osb + (int64_t)last * p.osn + h * p.osh + d;
    const float qe = (float)static_cast<const E*>(p.q)[qkv_at], ke = (float)static_cast<const E*>(p.k)[qkv_at];
    const float ve = (float)static_cast<const E*>(p.v)[qkv_at];
    const float oe = (float)static_cast<const E*>(p.out)[o_at], doe = (float)static_cast<const E*>(p.dout)[o_at];
    float s = qe * ke, dp = doe * ve, dl = doe * oe;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s += __shfl_xor(s, m);
        dp += __shfl_xor(dp, m);
        dl += __shfl_xor(dl, m);
    }
    const float pe = __builtin_amdgcn_exp2f(s * p.scale_log2 - p.lse[(int64_t)bh * p.n + last] * kLog2e), ds = pe * (dp - dl);
    acc += mat == 0 ? ds * qe : mat == 1 ? pe * doe : ds * ke;
    E* dst = static_cast<E*>(mat == 0 ? p.dk : mat == 1 ? p.dv : p.dq) + b * p.gsb + (int64_t)last * p.gsn + h * p.gsh + d;
    const E r = (E)(mat == 1 ? acc : acc * p.scale);
    *dst = r;
    if (p.colsum) {      // the three rows as one more partial row per batch element (behind the per-wave rows)
        const int which = mat == 0 ? 1 : mat == 1 ? 2 : 0;
        p.colsum[((int64_t)p.B * (p.n >> 7) * 4 + b) * (3 * 64 * (int64_t)p.H) + h * 192 + which * 64 + d] = (float)r;
    }
}

extern "C" int64_t npcd_attn_bwd_workspace_floats(int B, int n, int H) {
    if (B <= 0 || n <= 0 || H <= 0) return -1;
    return 2 * (int64_t)B * H * ((n + 63) / 64 * 64) + (edge_mode(n) ? (int64_t)B * H * (n >> 7) * 4 * kEdgeFloats : 0);
}

extern "C" int npcd_attn_bwd_colsum_rows(int B, int n, int H) {
    if (B <= 0 || n <= 0 || H <= 0) return -1;
    return npcd::edge_mode(n) ? B * ((n >> 7) * 4 + 1) : B * npcd::ceil_div(n, 128) * 4;
}

static int attn_bwd_launch(int passes, const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                           void* dq, void* dk, void* dv, float* delta, int B, int n, int H, int d,
                           int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                           int64_t g_sb, int64_t g_sn, int64_t g_sh, float scale, int dtype, void* stream, float* colsum = nullptr) {
    if (dtype == NPCD_F32) {
        if (B <= 0 || n <= 0 || H <= 0 || !q || !k || !v || !out || !dout || !lse || !delta || colsum) return NPCD_ERR_ARG;
        if (d != 64) return NPCD_ERR_UNSUPPORTED;
        if (((passes & 1) && !dq) || ((passes & 2) && (!dk || !dv))) return NPCD_ERR_ARG;
        const int64_t strides[] = {qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh, g_sb, g_sn, g_sh};
        for (int64_t x : strides)
            if (x % 4 != 0) return NPCD_ERR_ARG;
        if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || !aligned16(dout) || !aligned16(dq) || !aligned16(dk) ||
            !aligned16(dv))
            return NPCD_ERR_ARG;
        AttnF32Bwd a{static_cast<const float*>(q), static_cast<const float*>(k), static_cast<const float*>(v),
                     static_cast<const float*>(out), static_cast<const float*>(dout), lse,
                     static_cast<float*>(dq), static_cast<float*>(dk), static_cast<float*>(dv), delta, B, n, H,
                     qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh, g_sb, g_sn, g_sh, scale};
        const int grid = B * H * ceil_div(n, 128);
        hipStream_t st32 = static_cast<hipStream_t>(stream);
        if (passes & 1) hipLaunchKernelGGL(attn_bwd_f32_dq_kernel, dim3(grid), dim3(256), 0, st32, a);      // also writes delta
        if (passes & 2) hipLaunchKernelGGL(attn_bwd_f32_dkdv_kernel, dim3(grid), dim3(256), 0, st32, a);
        NPCD_HIP_CHECK(hipGetLastError());
        return NPCD_OK;
    }
    const bool force_gen = getenv("NPCD_ATTN_GEN") != nullptr;      // (read at every call, like NPCD_ATTN_FWD)
    const bool gen = d != 64 || (force_gen && !colsum);
    int rc = check_common(B, n, H, gen && attn_gen_supported(d) ? 64 : d, dtype);
    if (rc != NPCD_OK) return rc;
    if (!q || !k || !v || !out || !dout || !lse || !delta) return NPCD_ERR_ARG;
    if ((passes & 1) && !dq) return NPCD_ERR_ARG;
    if ((passes & 2) && (!dk || !dv)) return NPCD_ERR_ARG;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || !aligned16(dout) || !aligned16(dq) ||
        !aligned16(dk) || !aligned16(dv))
        return NPCD_ERR_ARG;
    if (!strides_ok(qkv_sb, qkv_sn, qkv_sh) || !strides_ok(out_sb, out_sn, out_sh) || !strides_ok(g_sb, g_sn, g_sh)) return NPCD_ERR_ARG;
    if (gen) {
        if (colsum) return NPCD_ERR_UNSUPPORTED;        // the column-sum by-product belongs to the d = 64 kernels (the fused backbone)
        return attn_gen_bwd(passes, q, k, v, out, dout, lse, dq, dk, dv, delta, B, n, H, d, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh,
                            g_sb, g_sn, g_sh, scale, dtype, stream);
    }
    AttnParams p{};
    p.q = q; p.k = k; p.v = v; p.out = out; p.dout = dout; p.lse = const_cast<float*>(lse); p.delta = delta;
    p.dq = dq; p.dk = dk; p.dv = dv;
    p.B = B; p.n = n; p.H = H;
    p.sb = qkv_sb; p.sn = qkv_sn; p.sh = qkv_sh;
    p.osb = out_sb; p.osn = out_sn; p.osh = out_sh;
    p.gsb = g_sb; p.gsn = g_sn; p.gsh = g_sh;
    p.scale = scale; p.scale_log2 = scale * kLog2e;
    if (colsum) {       // only for the packed c_qkv gradient: [.., H, (dq | dk | dv) x 64]
        const size_t es = 2;
        if (g_sh != 192 || (dk && reinterpret_cast<const char*>(dk) != reinterpret_cast<const char*>(dq) + 64 * es) ||
            (dv && reinterpret_cast<const char*>(dv) != reinterpret_cast<const char*>(dq) + 128 * es) || !aligned16(colsum))
            return NPCD_ERR_ARG;
        p.colsum = colsum;
    }
    const bool edge = edge_mode(n);
    const int grid = B * H * (edge ? n >> 7 : ceil_div(n, 128));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int dyn = 3 * kDkdvSlot;
    static DynLds lds_bf16, lds_f16;
    NPCD_HIP_CHECK(lds_bf16.ensure(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<BF16>), dyn));
    NPCD_HIP_CHECK(lds_f16.ensure(reinterpret_cast<const void*>(attn_bwd_dkdv_kernel<F16>), dyn));
    if (dtype == NPCD_BF16) {
        if (passes & 1) hipLaunchKernelGGL(attn_bwd_dq_kernel<BF16>, dim3(grid), dim3(256), 0, st, p);
        if (passes & 2) hipLaunchKernelGGL(attn_bwd_dkdv_kernel<BF16>, dim3(grid), dim3(256), dyn, st, p);
        if ((passes & 2) && edge) hipLaunchKernelGGL(attn_bwd_edge_kernel<BF16::elem>, dim3(B * H), dim3(192), 0, st, p);
    } else {
        if (passes & 1) hipLaunchKernelGGL(attn_bwd_dq_kernel<F16>, dim3(grid), dim3(256), 0, st, p);
        if (passes & 2) hipLaunchKernelGGL(attn_bwd_dkdv_kernel<F16>, dim3(grid), dim3(256), dyn, st, p);
        if ((passes & 2) && edge) hipLaunchKernelGGL(attn_bwd_edge_kernel<F16::elem>, dim3(B * H), dim3(192), 0, st, p);
    }
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int64_t npcd_attn_bwd_fused_slab_floats(int B, int n, int H) {
    if (B <= 0 || n <= 0 || H <= 0) return -1;
    return n <= 256 ? 0 : (int64_t)B * H * ((n + 63) / 64 * 64) * 64;
}

// The single-pass backward (attn_bwd_fused_kernel): same arguments as npcd_attn_bwd + dq_slab, an fp32 scratch of
// npcd_attn_bwd_fused_slab_floats(B, n, H) elements (may be NULL when that is 0).
extern "C" int npcd_attn_bwd_fused(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse,
                                   void* dq, void* dk, void* dv, float* delta, float* dq_slab, int B, int n, int H, int d,
                                   int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                                   int64_t g_sb, int64_t g_sn, int64_t g_sh, float scale, int dtype, void* stream) {
    int rc = check_common(B, n, H, d, dtype);
    if (rc != NPCD_OK) return rc;
    if (!q || !k || !v || !out || !dout || !lse || !delta || !dq || !dk || !dv) return NPCD_ERR_ARG;
    if (n > 256 && !dq_slab) return NPCD_ERR_ARG;
    if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(out) || !aligned16(dout) || !aligned16(dq) || !aligned16(dk) ||
        !aligned16(dv) || (dq_slab && !aligned16(dq_slab)))
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
    hipStream_t st = static_cast<hipStream_t>(stream);
    static DynLds lds_bf16, lds_f16;
    NPCD_HIP_CHECK(lds_bf16.ensure(reinterpret_cast<const void*>(attn_bwd_fused_kernel<BF16>), kFLds));
    NPCD_HIP_CHECK(lds_f16.ensure(reinterpret_cast<const void*>(attn_bwd_fused_kernel<F16>), kFLds));
    if (dtype == NPCD_BF16) hipLaunchKernelGGL(attn_bwd_fused_kernel<BF16>, dim3(B * H), dim3(512), kFLds, st, p, dq_slab);
    else hipLaunchKernelGGL(attn_bwd_fused_kernel<F16>, dim3(B * H), dim3(512), kFLds, st, p, dq_slab);
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

// npcd_attn_bwd / npcd_attn_bwd_pass with the column sums of the packed gradient as a by-product: colsum_part [rows + scratch, 3 H 64]
// fp32, rows = npcd_attn_bwd_colsum_rows(B, n, H) -- finished by npcd_colsum_finalize(colsum_part, rows, 3 H 64, bias_grad, ...).
extern "C" int npcd_attn_bwd_colsum(int passes, const void* q, const void* k, const void* v, const void* out, const void* dout,
                                    const float* lse, void* dq, void* dk, void* dv, float* delta, float* colsum_part, int B, int n, int H, int d,
                                    int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                                    int64_t g_sb, int64_t g_sn, int64_t g_sh, float scale, int dtype, void* stream) {
    if (passes < 1 || passes > 3 || !colsum_part || !dq) return NPCD_ERR_ARG;
    return attn_bwd_launch(passes, q, k, v, out, dout, lse, dq, dk, dv, delta, B, n, H, d, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh,
                           g_sb, g_sn, g_sh, scale, dtype, stream, colsum_part);
}

extern "C" int npcd_attn_bwd_pass(int pass, const void* q, const void* k, const void* v, const void* out, const void* dout,
                                  const float* lse, void* dq, void* dk, void* dv, float* delta, int B, int n, int H, int d,
                                  int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh, int64_t out_sb, int64_t out_sn, int64_t out_sh,
                                  int64_t g_sb, int64_t g_sn, int64_t g_sh, float scale, int dtype, void* stream) {
    if (pass != 1 && pass != 2) return NPCD_ERR_ARG;
    return attn_bwd_launch(pass, q, k, v, out, dout, lse, dq, dk, dv, delta, B, n, H, d, qkv_sb, qkv_sn, qkv_sh, out_sb, out_sn, out_sh,
                           g_sb, g_sn, g_sh, scale, dtype, stream);
}
