// Fused PointNeRF shading for gfx950: neighbour gather -> relative position -> positional encoding
// -> per-pair MLP -> inverse-distance aggregation -> density / colour heads.
//
// Reference op chain replaced (one eager PyTorch kernel per line there):
//   gather + x_rel + weights + posenc   npcd/models/pointnerf/fields/aggregators/mlp.py:62-88,
//                                       aggregator.py:122-156, utils/positional_encoder.py:16-20
//   local_field MLP (95->256x4->256)    aggregators/mlp.py:83-84, utils/model.py:22-36
//   weighted aggregation (index_add_)   aggregators/mlp.py:102-125
//   shape_net + softplus(x-1)           fields/mlp.py:38-51, field.py:30,126-128
//   channel_net + sigmoid               fields/mlp.py:53-72, field.py:139-140
//
// Design.  Activations never leave the CU between layers: a 128-row tile lives in LDS as fp16
// (64 KiB, XOR-swizzled 512-byte rows) and every 256x256 layer is a chain of 32x32x16 f16 MFMAs
// with fp32 accumulation, oriented as  H_out^T[out][row] = W[out][in] . H_in^T[in][row]  so that
//   * the WEIGHTS are the A operand, read straight from global/L2 in a pre-packed fragment order
//     (1 KiB contiguous per wave-instruction, no LDS staging: all workgroups stream the same
//     1.2 MB, which stays L2 resident),
//   * the ACTIVATIONS are the B operand (ds_read_b128 of 8 consecutive input channels of one row),
//   * each lane ends up owning 4 consecutive output channels of one row per accumulator group, so
//     the epilogue (bias, LeakyReLU, fp16 convert) writes 8-byte packed values back in place.
// The last aggregator layer is linear, so it commutes with the (normalised) weighted sum over a
// point's neighbours: kernel A runs the four non-linear layers on the (point, neighbour) pairs and
// aggregates; kernel B applies that last linear layer and both heads on POINTS (6.3x fewer rows).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "shade_common.h"

namespace npcd {

// LDS byte offset of 16-byte chunk `chunk` (0..31) of activation row `row`.  The row pitch is 512 + 16 bytes: consecutive rows
// start 4 banks apart, so the 16 rows a ds_read_b128 lane group touches (same chunk) cover all 64 banks -- conflict-free like
// an XOR swizzle, but every address is "per-lane base + compile-time constant": no vector arithmetic inside the layer loops
// (the kernel is bound by vector-instruction issue, not by the matrix pipe).
__device__ __forceinline__ int act_off(int row, int chunk) { return row * kRowBytes + (chunk << 4); }
// The 16x16x32 form of the pair kernel uses a pitch of 512 + 32 bytes: ds_read_b128 is served in four fixed groups of 16 lanes that mix
// the halves of two k-groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS), i.e. eight rows at chunk c and the OTHER eight rows of
// the 16-row block at chunk c + 1.  At 528 bytes (rows 4 banks apart) rows 12 and 11 of such a group meet in one bank quad -- 36 % of the
// kernel's LDS cycles were conflict cycles; at 544 bytes the rows of a group sit on even and odd quads.
constexpr int kRowBytes16 = 544;
template <int PITCH>
__device__ __forceinline__ int act_off_p(int row, int chunk) { return row * PITCH + (chunk << 4); }

// One layer:  acc[oi][cb] (+)= W[(2*wave+oi)*32.., :] . H^T[:, cb*32..]   for oi in {0,1}, cb in 0..NB-1.
// NB (the number of 32-row blocks that hold packed rows) is a COMPILE-TIME parameter: as a run-time bound inside the k-loop it
// split the loop body into one basic block per row block -- LDS read, wait, two matrix instructions, branch -- so that no read
// was ever in flight behind a matrix instruction (round 2: the kernel ran at the latency of its LDS reads).
template <int KSTEPS, int UNROLL = 4, int NB = 4, int PF = 1>
__device__ __forceinline__ void layer_mfma(const unsigned char* H, const unsigned char* wfrag, const float* bias, int wave, int lane,
                                           f32x16 (&acc)[2][4]) {
    const int r = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int oi = 0; oi < 2; ++oi) {
        const float* bp = bias + (2 * wave + oi) * 32;
        f32x16 init;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(bp + acc_group_off(g, hh));
#pragma unroll
            for (int b = 0; b < 4; ++b) init[4 * g + b] = b4[b];
        }
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) acc[oi][cb] = init;
    }
    const f16x8* w0 = reinterpret_cast<const f16x8*>(wfrag + (int64_t)(2 * wave) * KSTEPS * kFragBytes) + lane;
    const f16x8* w1 = reinterpret_cast<const f16x8*>(wfrag + (int64_t)(2 * wave + 1) * KSTEPS * kFragBytes) + lane;
    const unsigned char* hb = H + r * kRowBytes + hh * 16;      // B operand: row (cb*32 + r), chunk 2s + hh -> hb + const
    if constexpr (PF > 1) {
        // weight fragments PF k-steps ahead through a ring of 4 register pairs (the one-step form below leaves the loop at the
        // latency of an L2 hit per k-step when few row blocks share a fragment)
        static_assert(PF <= 3, "ring of four");
        f16x8 ra[4], rb[4];
#pragma unroll
        for (int d = 0; d < PF; ++d) { ra[d] = w0[d * 64]; rb[d] = w1[d * 64]; }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            if (s + PF < KSTEPS) {
                ra[(s + PF) & 3] = w0[(s + PF) * 64];
                rb[(s + PF) & 3] = w1[(s + PF) * 64];
            }
            f16x8 b[NB];
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) b[cb] = *reinterpret_cast<const f16x8*>(hb + cb * 32 * kRowBytes + s * 32);
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                acc[0][cb] = F16::mfma32(ra[s & 3], b[cb], acc[0][cb]);
                acc[1][cb] = F16::mfma32(rb[s & 3], b[cb], acc[1][cb]);
            }
        }
        return;
    }
    f16x8 a0 = w0[0], a1 = w1[0];
#pragma unroll UNROLL
    for (int s = 0; s < KSTEPS; ++s) {
        f16x8 n0 = a0, n1 = a1;
#ifndef NPCD_DIAG_NO_WLOAD          // DIAGNOSTIC builds only (wrong results, timing): the layer loops without their weight loads
        if (s + 1 < KSTEPS) {
            n0 = w0[(s + 1) * 64];
            n1 = w1[(s + 1) * 64];
        }
#endif
        f16x8 b[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) b[cb] = *reinterpret_cast<const f16x8*>(hb + cb * 32 * kRowBytes + s * 32);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {            // 32-row blocks past the tile's packed rows are skipped (NB < 4)
            acc[0][cb] = F16::mfma32(a0, b[cb], acc[0][cb]);
            acc[1][cb] = F16::mfma32(a1, b[cb], acc[1][cb]);
        }
        a0 = n0;
        a1 = n1;
    }
}

// ---- the same layer as a PINNED software pipeline (pair kernel) --------------------------------------------------------------
// In the form above the machine scheduler sinks every weight load down to its first use (register pressure), whatever the source
// order says: the listing shows `global_load ... s_waitcnt vmcnt(1) ... v_mfma`, i.e. one L2 round trip per k-step that only the
// second wave of the SIMD can hide.  Here the order is fixed with scheduling fences: per k-step  [weight fragments of step s + PF |
// activation fragments of step s + 1] fence [the step's 2 NB matrix instructions] fence.  The bias values (the C operand of the
// first step) and the first PF weight fragments of a layer are requested by layer_prefetch(), which the caller issues BEFORE the
// previous layer's epilogue, so that they travel during the barrier and the LDS write-back.
// Weights and biases come through a raw BUFFER resource over the pack: address = scalar offset (layer, wave, group of four k-steps)
// + per-lane constant (lane * 16) + 12-bit immediate -- no vector address arithmetic at all (as `global_load` the compiler
// re-associated base + lane offset into 64-bit vector addresses and two v_add_co per 4-KB window).
struct WRing { f16x8 a[4], b[4]; };
typedef __amdgpu_buffer_rsrc_t wrsrc_t;
__device__ __forceinline__ wrsrc_t pack_rsrc(const unsigned char* wpack, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wpack), 0, (int)bytes, 0x00020000);
}
template <int KSTEPS>
__device__ __forceinline__ f16x8 wfrag_load(wrsrc_t rs, int w_off, int wave, int oi, int lane, int step) {
    const int so = w_off + (2 * wave + oi) * (KSTEPS * kFragBytes) + (step >> 2) * (4 * kFragBytes);
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + (step & 3) * kFragBytes, so, 0));
}
template <int KSTEPS, int PF>
__device__ __forceinline__ void layer_prefetch(wrsrc_t rs, int w_off, int b_off, int wave, int lane, WRing& ring, f32x16 (&init)[2]) {
    const int hh = lane >> 5;
#pragma unroll
    for (int oi = 0; oi < 2; ++oi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {       // bias of out channel (2 wave + oi) * 32 + acc_group_off(g, hh) + b
            const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, hh * 32 + (g >> 1) * 64 + (g & 1) * 16, b_off + (2 * wave + oi) * 128, 0));
#pragma unroll
            for (int b = 0; b < 4; ++b) init[oi][4 * g + b] = b4[b];
        }
#pragma unroll
    for (int d = 0; d < PF && d < KSTEPS; ++d) {
        ring.a[d] = wfrag_load<KSTEPS>(rs, w_off, wave, 0, lane, d);
        ring.b[d] = wfrag_load<KSTEPS>(rs, w_off, wave, 1, lane, d);
    }
}
template <int KSTEPS, int NB, int PF>
__device__ __forceinline__ void layer_mfma_pipe(const unsigned char* H, wrsrc_t rs, int w_off, int wave, int lane, WRing& ring,
                                                const f32x16 (&init)[2], f32x16 (&acc)[2][4]) {
    static_assert(PF >= 1 && PF <= 3, "ring of four");
    const int r = lane & 31, hh = lane >> 5;
    const unsigned char* hb = H + r * kRowBytes + hh * 16;
    f16x8 bc[NB], bn[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) bc[cb] = *reinterpret_cast<const f16x8*>(hb + cb * 32 * kRowBytes);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        if (s + PF < KSTEPS) {
            ring.a[(s + PF) & 3] = wfrag_load<KSTEPS>(rs, w_off, wave, 0, lane, s + PF);
            ring.b[(s + PF) & 3] = wfrag_load<KSTEPS>(rs, w_off, wave, 1, lane, s + PF);
        }
        if (s + 1 < KSTEPS) {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) bn[cb] = *reinterpret_cast<const f16x8*>(hb + cb * 32 * kRowBytes + (s + 1) * 32);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            acc[0][cb] = F16::mfma32(ring.a[s & 3], bc[cb], s == 0 ? init[0] : acc[0][cb]);
            acc[1][cb] = F16::mfma32(ring.b[s & 3], bc[cb], s == 0 ? init[1] : acc[1][cb]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) bc[cb] = bn[cb];
    }
}

// epilogue: optional LeakyReLU, convert to fp16, write back in place.  With the rows of the matrices permuted in the pack
// (acc_channel, shade_common.h) the register groups 2 q, 2 q + 1 of a lane are 8 consecutive channels of its row: every lane writes
// whole 16-byte chunks (ds_write_b128: 16 rows per pass, 4 banks apart = all 64 banks; as 8-byte writes rows r and r + 16 of a
// 32-lane pass met in the same banks -- the 2-way conflicts of profiles/r3_shade_sq_pmc.json).
template <bool ACT, int NB = 4>
__device__ __forceinline__ void layer_store(unsigned char* H, int wave, int lane, const f32x16 (&acc)[2][4]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const int r = lane & 31, hh = lane >> 5;
    unsigned char* sb = H + r * kRowBytes + hh * 16 + wave * 128;     // row (cb*32 + r), channels (2 wave + oi)*32 + 16 gp + 8 hh .. + 7
#pragma unroll
    for (int oi = 0; oi < 2; ++oi)
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint32_t v[2][2];        // [g - 2 gp][dword]: channels 16 gp + 8 hh + 4 (g - 2 gp) + {0,1 | 2,3}
#pragma unroll
                for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int e = 4 * (2 * gp + gg) + 2 * b;
                        float x0 = acc[oi][cb][e], x1 = acc[oi][cb][e + 1];
#ifdef NPCD_SHADE_LEAKY_F32
                        if (ACT) {                       // LeakyReLU(0.01): max(x, 0.01 x)
                            x0 = fmaxf(x0, kLeaky * x0);
                            x1 = fmaxf(x1, kLeaky * x1);
                        }
                        const f32x2 f = {x0, x1};
                        v[gg][b] = __builtin_bit_cast(uint32_t, __builtin_convertvector(f, f16x2));
#else
                        // LeakyReLU on the converted pair: max(h, 0.01 h) in packed fp16 -- 1.5 vector instructions per value instead
                        // of 2.5 (the negative side is rounded twice: to fp16, and after the scaling)
                        const f32x2 f = {x0, x1};
                        f16x2 h = __builtin_convertvector(f, f16x2);
                        if (ACT) {
                            const f16x2 sc = {(_Float16)kLeaky, (_Float16)kLeaky};
                            h = __builtin_elementwise_max(h, h * sc);
                        }
                        v[gg][b] = __builtin_bit_cast(uint32_t, h);
#endif
                    }
                *reinterpret_cast<u32x4*>(sb + cb * 32 * kRowBytes + oi * 64 + gp * 32) = u32x4{v[0][0], v[0][1], v[1][0], v[1][1]};
            }
}

// ============================================================================================
// kernel A: (point, neighbour) pairs.  tile = 16 points x 8 neighbour slots = 128 rows
// ============================================================================================
#ifdef NPCD_SHADE_TL
__device__ long long g_shade_tl[64];
__device__ long long g_shade_span[512 * 4];          // per workgroup: begin / end (100-MHz real time), tiles done, clocks
#define NPCD_STS(i) do { if (tl_on) tl[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NPCD_STS(i) do { } while (0)
#endif
#ifdef NPCD_SHADE_TL
#define NPCD_TL_PARAMS , long long* tl, bool tl_hit
#define NPCD_TL_ARGS , tl, tl_on
#define NPCD_STL(i) do { if (tl_hit) tl[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NPCD_TL_PARAMS
#define NPCD_TL_ARGS
#define NPCD_STL(i) do { } while (0)
#endif
#ifndef NPCD_PAIRS_PF
#define NPCD_PAIRS_PF 2          // weight fragments in flight ahead of the matrix instructions (k-steps); 3 measured the same
#endif
struct PairPack { wrsrc_t rs; int w1, b0; };          // buffer resource of the weight pack, offsets of A1 and of the first bias
// the four non-linear aggregator layers of one tile, activations in place in LDS (two barriers per layer)
template <int FEAT, int NB>
__device__ __forceinline__ void pair_layers(unsigned char* H, const PairPack& pk, int wave, int lane NPCD_TL_PARAMS) {
    constexpr int K0 = FEAT + kEncBlock, PF = NPCD_PAIRS_PF;
    constexpr int kLayerBytes = 8 * (kHidden / 16) * kFragBytes;
    f32x16 acc[2][4];
    WRing ring;
    f32x16 init[2];
    layer_prefetch<K0 / 16, PF>(pk.rs, 0, pk.b0, wave, lane, ring, init);
    layer_mfma_pipe<K0 / 16, NB, PF>(H, pk.rs, 0, wave, lane, ring, init, acc);
    layer_prefetch<kHidden / 16, PF>(pk.rs, pk.w1, pk.b0 + kHidden * 4, wave, lane, ring, init);
    NPCD_STL(3);
    __syncthreads();
    NPCD_STL(4);
    layer_store<true, NB>(H, wave, lane, acc);
    NPCD_STL(5);
    __syncthreads();
    NPCD_STL(6);
#pragma unroll 1
    for (int l = 1; l < 4; ++l) {
        // (offsets in closed form: indexing the layout's tables with the loop counter put the whole struct into scratch memory)
        const int w_off = pk.w1 + (l - 1) * kLayerBytes, b_off = pk.b0 + l * (kHidden * 4);
        layer_mfma_pipe<kHidden / 16, NB, PF>(H, pk.rs, w_off, wave, lane, ring, init, acc);
        // the next layer's bias and first fragments travel during the barrier and the write-back (after the last layer: not needed)
        if (l < 3) layer_prefetch<kHidden / 16, PF>(pk.rs, w_off + kLayerBytes, b_off + kHidden * 4, wave, lane, ring, init);
        if (l == 1) NPCD_STL(7);
        __syncthreads();
        if (l == 1) NPCD_STL(8);
        layer_store<true, NB>(H, wave, lane, acc);
        if (l == 1) NPCD_STL(9);
        __syncthreads();
        if (l == 1) NPCD_STL(10);
    }
}

// ---- the four layers on v_mfma_f32_16x16x32_f16 (round 5, experiments R5.13) -------------------------------------------------------
// The shading kernels are POWER-bound on this pool, and the clock the chip holds depends on the instruction: the bare layer loop
// (tools/probes/mfma_energy_probe.hip, random operands) runs 32x32x16 at 1.44-1.53 GHz and 16x16x32 at 1.66-1.78 GHz -- the same
// FLOPs, the same operand bytes, half the accumulator traffic per FLOP.  Same wave tile (64 output channels x <= 128 rows, 128
// accumulator registers as 4 channel blocks x 8 row blocks of 16 x 16), same LDS bytes: per 32 input channels 4 weight fragments
// (one 4-KiB run of the pack) and 8 activation fragments (lane = (row l & 15, 16-byte chunk l >> 4)), in two half-steps of 4 row
// blocks so that the fragment registers stay what they were.  Output channel of accumulator row 4 g + b of block mb:
// 64 wave + 32 (mb >> 1) + 8 g + 4 (mb & 1) + b (the pack's row order): blocks 2 p, 2 p + 1 of a lane are 8 CONSECUTIVE channels, a
// whole 16-byte chunk per store.  Row granularity 16: a tile computes ceil(V / 16) row blocks (the 32x32 form: 2 ceil(V / 32)).
typedef float f32x4a __attribute__((ext_vector_type(4)));
struct WRing16 { f16x8 a[2][4]; };
__device__ __forceinline__ f32x4a mfma16(f16x8 a, f16x8 b, f32x4a c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
template <int KS32>
__device__ __forceinline__ f16x8 wfrag16(wrsrc_t rs, int w_off, int wave, int lane, int s, int mb) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + mb * kFragBytes, w_off + (wave * KS32 + s) * (4 * kFragBytes), 0));
}
// (the first weight fragments of a layer are requested BEFORE the previous layer's write-back, like in the 32x32 form.  The bias --
//  the C operand of the first step -- is read from an LDS copy of the four layers' biases instead: prefetched through registers it was
//  spilled across the write-back, with a blocking wait on the load it had just issued)
template <int KS32>
__device__ __forceinline__ void layer_prefetch16(wrsrc_t rs, int w_off, int wave, int lane, WRing16& ring) {
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) ring.a[0][mb] = wfrag16<KS32>(rs, w_off, wave, lane, 0, mb);
}
template <int KS32, int NB16, int PITCH = kRowBytes>
__device__ __forceinline__ void layer_mfma16(const unsigned char* H, const unsigned char* bias_l, wrsrc_t rs, int w_off, int wave, int lane,
                                             WRing16& ring, f32x4a (&acc)[4][8]) {
    constexpr int NH = NB16 > 4 ? 2 : 1;                     // half-steps per 32 input channels
    const unsigned char* hb = H + (lane & 15) * PITCH + (lane >> 4) * 16;
    f32x4a init[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)      // bias16: [wave][g][mb][4]
        init[mb] = *reinterpret_cast<const f32x4a*>(bias_l + wave * 256 + (lane >> 4) * 64 + mb * 16);
    f16x8 bc[4], bn[4];
#pragma unroll
    for (int r = 0; r < (NB16 < 4 ? NB16 : 4); ++r) bc[r] = *reinterpret_cast<const f16x8*>(hb + r * 16 * PITCH);
#pragma unroll
    for (int s = 0; s < KS32; ++s) {
        if (s + 1 < KS32) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) ring.a[(s + 1) & 1][mb] = wfrag16<KS32>(rs, w_off, wave, lane, s + 1, mb);
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int r0 = h * 4, nr = (NB16 - r0) < 4 ? (NB16 - r0) : 4;                       // this half-step's row blocks
            const int ns = (h + 1 < NH) ? s : s + 1, nr0 = (h + 1 < NH) ? 4 : 0;                 // the next half-step
            const int nnr = (NB16 - nr0) < 4 ? (NB16 - nr0) : 4;
            if (ns < KS32) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < nnr) bn[r] = *reinterpret_cast<const f16x8*>(hb + (nr0 + r) * 16 * PITCH + ns * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nr) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) acc[mb][r0 + r] = mfma16(ring.a[s & 1][mb], bc[r], s == 0 ? init[mb] : acc[mb][r0 + r]);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) bc[r] = bn[r];
        }
    }
}
template <bool ACT, int NB16, int PITCH = kRowBytes>
__device__ __forceinline__ void layer_store16(unsigned char* H, int wave, int lane, const f32x4a (&acc)[4][8]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    unsigned char* sb = H + (lane & 15) * PITCH + (lane >> 4) * 16 + wave * 128;     // row rb*16 + (l & 15), chunk 8 wave + 4 p + g
#pragma unroll
    for (int rb = 0; rb < NB16; ++rb)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
            uint32_t v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {       // channels 2 q, 2 q + 1 of the chunk: block 2 p + (q >> 1), registers 2 (q & 1), + 1
                const f32x2 f = {acc[2 * p2 + (q >> 1)][rb][2 * (q & 1)], acc[2 * p2 + (q >> 1)][rb][2 * (q & 1) + 1]};
                f16x2 h = __builtin_convertvector(f, f16x2);
                if (ACT) {
                    const f16x2 sc = {(_Float16)kLeaky, (_Float16)kLeaky};
                    h = __builtin_elementwise_max(h, h * sc);
                }
                v[q] = __builtin_bit_cast(uint32_t, h);
            }
            *reinterpret_cast<u32x4*>(sb + rb * 16 * PITCH + p2 * 64) = u32x4{v[0], v[1], v[2], v[3]};
        }
}
struct PairPack16 { wrsrc_t rs; int w0, w1; const unsigned char* bias; };      // offsets of A0 / A1 in the pack, the LDS copy of the biases
template <int FEAT, int NB16>
__device__ __forceinline__ void pair_layers16(unsigned char* H, const PairPack16& pk, int wave, int lane) {
    constexpr int K0 = FEAT + kEncBlock;
    f32x4a acc[4][8];
    WRing16 ring;
    layer_prefetch16<K0 / 32>(pk.rs, pk.w0, wave, lane, ring);
    layer_mfma16<K0 / 32, NB16, kRowBytes16>(H, pk.bias, pk.rs, pk.w0, wave, lane, ring, acc);
    layer_prefetch16<kHidden / 32>(pk.rs, pk.w1, wave, lane, ring);
    __syncthreads();
    layer_store16<true, NB16, kRowBytes16>(H, wave, lane, acc);
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l < 4; ++l) {
        // (A1..A3 are equally spaced: offsets in closed form, no indexed struct access -- see pair_layers)
        const int w_off = pk.w1 + (l - 1) * (kHidden * kHidden * 2);
        layer_mfma16<kHidden / 32, NB16, kRowBytes16>(H, pk.bias + l * (kHidden * 4), pk.rs, w_off, wave, lane, ring, acc);
        if (l < 3) layer_prefetch16<kHidden / 32>(pk.rs, w_off + kHidden * kHidden * 2, wave, lane, ring);
        __syncthreads();
        layer_store16<true, NB16, kRowBytes16>(H, wave, lane, acc);
        __syncthreads();
    }
}

// What a thread needs to build its input row of a tile: requested one tile AHEAD (the neighbour indices while the previous tile's
// layers run, the gathered data while its aggregation runs), so that the prologue starts from registers instead of two dependent
// trips to memory.
// (the thread index made opaque per phase: otherwise every lane-dependent address of the prologue, the gathers and the aggregation is
// hoisted out of the tile loop and held in registers across the layers, which have none to spare)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
template <int FEAT>
struct PairInputs {
    int gi, gi_other;                  // neighbour index of this thread's candidate, and of the candidate 64 rows away (for the counts)
    float pt[3], kp[3];                // shading point, neighbour position
    f32x4 feat[FEAT / 8];              // this half's FEAT / 2 neighbour features
};
template <int FEAT>
__device__ __forceinline__ void pair_load_indices(const ShadeArgs& a, int tile, int P, int tid, PairInputs<FEAT>& in) {
    tid = opaque(tid);
    const int row = tid & 127, slot = row & 7;
    const int p = tile * 16 + (row >> 3), po = tile * 16 + ((row ^ 64) >> 3);
    in.gi = -1;
    in.gi_other = -1;
    if (p < P && slot < a.k) in.gi = a.nb_idx[(int64_t)p * a.k + slot];
    if (po < P && slot < a.k) in.gi_other = a.nb_idx[(int64_t)po * a.k + slot];
}
template <int FEAT>
__device__ __forceinline__ void pair_load_data(const ShadeArgs& a, int tile, int tid, PairInputs<FEAT>& in) {
    // (unconditional, with clamped indices: loads under `if (gi >= 0)` would keep the previous tile's values alive across the layers)
    const int row = opaque(tid) & 127, half = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int p = min(tile * 16 + (row >> 3), a.max_points - 1), g = max(in.gi, 0);
#pragma unroll
    for (int c = 0; c < 3; ++c) { in.pt[c] = a.pts[(int64_t)p * 3 + c]; in.kp[c] = a.kp_pos[(int64_t)g * 3 + c]; }
    const float* fp = a.kp_feat + (int64_t)g * FEAT + half * (FEAT / 2);
#pragma unroll
    for (int c4 = 0; c4 < FEAT / 8; ++c4) in.feat[c4] = *reinterpret_cast<const f32x4*>(fp + c4 * 4);
}

template <int FEAT, int FORM = 0>          // FORM 0: v_mfma_f32_32x32x16_f16 layers, 1: the 16x16x32 layers
__global__ __launch_bounds__(256, 2) void shade_pairs_kernel(ShadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    unsigned char* H = dsmem;
    constexpr int RB = FORM == 1 ? kRowBytes16 : kRowBytes;              // row pitch of the activation tile
    float* wrow = reinterpret_cast<float*>(dsmem + kRows * RB);         // [128] inverse distances of the packed rows
    int* pstart = reinterpret_cast<int*>(wrow + kRows);                 // [16] first packed row of each point of the tile
    int* pcount = pstart + 16;                                          // [16] its number of valid neighbours
    // (the wave number as a SCALAR: the weight / bias addresses of the layer loops are scalar offset + per-lane constant)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ShadeLayout L = shade_layout(FEAT);
    const PairPack pk = {pack_rsrc(a.wpack, L.total), (int)L.w[1], (int)L.bias[0]};
    // (FORM 1: the four layers' biases, 4 KiB in the order of their C operands, copied to LDS once per workgroup; first read behind the
    //  first tile's prologue barrier)
    unsigned char* bias_lds = reinterpret_cast<unsigned char*>(pcount + 16 + 4);
    const PairPack16 pk16 = {pk.rs, (int)L.w16[0], (int)L.w16[1], bias_lds};
    if constexpr (FORM == 1) {
        for (int i = tid; i < 4 * kHidden / 4; i += 256)
            reinterpret_cast<f32x4*>(bias_lds)[i] = *reinterpret_cast<const f32x4*>(a.wpack + L.bias16[0] + (int64_t)i * 16);
    }
    // the count comes from device memory (the compact query's counter); that counter keeps counting past the capacity of the
    // lists when they overflow (the host then retries with larger buffers), so it is clamped to the rows that exist
    const int P = min(*a.n_points, a.max_points);
    const int ntiles = (P + 15) / 16;

#ifdef NPCD_SHADE_TL
    // stamps of EVERY tile of workgroup NPCD_SHADE_TL (wave 0), summed per interval over its tiles with four occupied row blocks:
    // g_shade_tl[0..13] = interval sums, [14] = such tiles, [15] = all tiles, [16] = the workgroup's whole time, [17] = sum of nblk
    const bool tl_on = blockIdx.x == NPCD_SHADE_TL && wave == 0;
    long long tl[16], tl_sum[16];
    for (int i = 0; i < 16; ++i) { tl[i] = 0; tl_sum[i] = 0; }
    long long tl_n4 = 0, tl_n = 0, tl_nb = 0;
    const long long tl_begin = __builtin_amdgcn_s_memtime();
    const long long rt_begin = __builtin_amdgcn_s_memrealtime();
    int span_tiles = 0;
#endif
    // Tiles differ in cost (1-4 occupied row blocks) and a 128^2 view has ~9.1 tiles per resident workgroup: handed out through
    // a counter the kernel ends when the LAST tile ends, not when the unluckiest stride does.
    int* next_tile = pcount + 16;
    int tile = blockIdx.x;
    PairInputs<FEAT> in;
    if (tile < ntiles) {
        pair_load_indices<FEAT>(a, tile, P, tid, in);
        pair_load_data<FEAT>(a, tile, tid, in);
    }
    while (tile < ntiles) {
        if (tid == 0) *next_tile = a.tile_counter ? (int)gridDim.x + atomicAdd(a.tile_counter, 1) : tile + (int)gridDim.x;
        NPCD_STS(0);
        // ---- prologue: build the layer-0 input rows -------------------------------------------
        // The tile's 16 x 8 (point, slot) candidates are PACKED: a valid pair takes the row number "valid pairs before
        // it", so the rows of a point stay consecutive and the unused rows collect at the end of the tile, where whole
        // 32-row MFMA blocks are skipped (typically one in four: ~29 % of the slots are empty).
        int nblk;
        {
            const int row = opaque(tid) & 127, half = __builtin_amdgcn_readfirstlane(tid >> 7);  // half is wave-uniform
            const int slot = row & 7;
            const int gi = in.gi;
            const unsigned long long mine = __ballot(gi >= 0), other = __ballot(in.gi_other >= 0);
            const int n_mine = __popcll(mine), n_other = __popcll(other);
            const int before = __popcll(mine & ((1ull << lane) - 1ull)) + ((row & 64) ? n_other : 0);
            const int V = n_mine + n_other;
            nblk = FORM == 1 ? (V + 15) >> 4 : (V + 31) >> 5;               // occupied row blocks (of 16 / of 32 rows)
            const int prow = before;                                        // packed row of this candidate (if valid)
            if (half == 0 && slot == 0) {
                pstart[row >> 3] = before;
                pcount[row >> 3] = __popcll((mine >> (lane & ~7)) & 0xffull);
            }
            float rel[3] = {0.f, 0.f, 0.f};
            if (gi >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rel[c] = in.pt[c] - in.kp[c];
            }
            if (half == 0 && gi >= 0) wrow[prow] = 1.f / (sqrtf(__builtin_fmaf(rel[0], rel[0], __builtin_fmaf(rel[1], rel[1], rel[2] * rel[2]))) + 1e-5f);
            constexpr int FH = FEAT / 2;
            if (gi >= 0) {
                // features: this half converts FEAT/2 channels
#pragma unroll
                for (int c8 = 0; c8 < FH / 8; ++c8) {
                    f16x8 v;
                    const f32x4 x0 = in.feat[2 * c8], x1 = in.feat[2 * c8 + 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = (_Float16)x0[j]; v[4 + j] = (_Float16)x1[j]; }
                    *reinterpret_cast<f16x8*>(H + act_off_p<RB>(prow, half * (FH / 8) + c8)) = v;
                }
                // positional encoding: this half fills 32 of the 64 columns (the select on `half` is wave-uniform)
#pragma unroll
                for (int c8 = 0; c8 < 4; ++c8) {
                    f16x8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (_Float16)(half == 0 ? enc_value(c8 * 8 + j, rel) : enc_value(32 + c8 * 8 + j, rel));
                    *reinterpret_cast<f16x8*>(H + act_off_p<RB>(prow, FEAT / 8 + half * 4 + c8)) = v;
                }
            }
            // rows V .. 32 nblk - 1 are computed (whole MFMA blocks) but never aggregated: give them defined inputs
            if (row >= V && row < (FORM == 1 ? 16 : 32) * nblk) {
                f16x8 z;
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
#pragma unroll
                for (int c8 = 0; c8 < FH / 8; ++c8) *reinterpret_cast<f16x8*>(H + act_off_p<RB>(row, half * (FH / 8) + c8)) = z;
#pragma unroll
                for (int c8 = 0; c8 < 4; ++c8) *reinterpret_cast<f16x8*>(H + act_off_p<RB>(row, FEAT / 8 + half * 4 + c8)) = z;
            }
        }
        NPCD_STS(1);
        __syncthreads();
        NPCD_STS(2);
        // the next tile (written at the top of this iteration) and its neighbour indices: they arrive while the layers run
        const int ntile = *next_tile;
        if (ntile < ntiles) pair_load_indices<FEAT>(a, ntile, P, tid, in);
        // ---- four non-linear layers (one instantiation per number of occupied 32-row blocks; nblk is workgroup-uniform) ----
        if constexpr (FORM == 1) {
            switch (nblk) {
                case 8: pair_layers16<FEAT, 8>(H, pk16, wave, lane); break;
                case 7: pair_layers16<FEAT, 7>(H, pk16, wave, lane); break;
                case 6: pair_layers16<FEAT, 6>(H, pk16, wave, lane); break;
                case 5: pair_layers16<FEAT, 5>(H, pk16, wave, lane); break;
                case 4: pair_layers16<FEAT, 4>(H, pk16, wave, lane); break;
                case 3: pair_layers16<FEAT, 3>(H, pk16, wave, lane); break;
                case 2: pair_layers16<FEAT, 2>(H, pk16, wave, lane); break;
                case 1: pair_layers16<FEAT, 1>(H, pk16, wave, lane); break;
                default: break;
            }
        } else {
        switch (nblk) {
            case 4: pair_layers<FEAT, 4>(H, pk, wave, lane NPCD_TL_ARGS); break;
            case 3: pair_layers<FEAT, 3>(H, pk, wave, lane NPCD_TL_ARGS); break;
            case 2: pair_layers<FEAT, 2>(H, pk, wave, lane NPCD_TL_ARGS); break;
            case 1: pair_layers<FEAT, 1>(H, pk, wave, lane NPCD_TL_ARGS); break;
            default: break;                                    // no valid pair in the tile
        }
        }
        NPCD_STS(11);
        // ... and its gathered points / positions / features while the aggregation runs
        // (unconditional, tile clamped: under `if (ntile < ntiles)` the previous values would have to survive the layers)
        pair_load_data<FEAT>(a, min(ntile, ntiles - 1), tid, in);
        __builtin_amdgcn_sched_barrier(0);
        // ---- inverse-distance aggregation over the 8 neighbour slots ------------------------
        {
            // point, and its channels 8 cc .. + 7 and 128 + 8 cc .. + 7: the 16 lanes of a point read 16 different 16-byte chunks of
            // a row per instruction (as chunks 2 cc, 2 cc + 1 they met pairwise in the same banks)
            const int ta = opaque(tid), pl = ta >> 4, cc = ta & 15;
            const int p = tile * 16 + pl;
            const int r0 = pstart[pl], cnt = pcount[pl];             // the point's packed rows
            // All eight slots at once: slot s2 >= cnt re-reads the point's last row with weight 0 (adds +0: the sums are the same
            // bits as a loop over cnt rows), so that the 8 + 16 LDS reads are independent and in flight together -- as a loop
            // with a run-time trip count every iteration waited for its own reads.
            const int last = max(cnt - 1, 0);
            float w8[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) w8[s2] = wrow[r0 + min(s2, last)];
            f16x8 v0[8], v1[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                const int row = r0 + min(s2, last);
                v0[s2] = *reinterpret_cast<const f16x8*>(H + act_off_p<RB>(row, cc));
                v1[s2] = *reinterpret_cast<const f16x8*>(H + act_off_p<RB>(row, cc + 16));
            }
            float wsum = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                w8[s2] = s2 < cnt ? w8[s2] : 0.f;
                wsum += w8[s2];
            }
            const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
            float out[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) out[j] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                const float ws = w8[s2] * inv;
#pragma unroll
                for (int j = 0; j < 8; ++j) {      // (explicit fused multiply-adds: the eight-wave form must round the same way)
                    out[j] = __builtin_fmaf(ws, (float)v0[s2][j], out[j]);
                    out[8 + j] = __builtin_fmaf(ws, (float)v1[s2][j], out[8 + j]);
                }
            }
            if (cnt == 0) {          // (the re-read row of a point without neighbours holds whatever the tile left there)
#pragma unroll
                for (int j = 0; j < 16; ++j) out[j] = 0.f;
            }
            if (p < P) {
                f16x8 o0, o1;
#pragma unroll
                for (int j = 0; j < 8; ++j) { o0[j] = (_Float16)out[j]; o1[j] = (_Float16)out[8 + j]; }
                f16x8* gp = reinterpret_cast<f16x8*>(a.G + (int64_t)p * kHidden + cc * 8);
                gp[0] = o0;
                gp[16] = o1;
                // range guard: an fp16 overflow in any of the four layers makes every channel of the pair's row inf / NaN in the next
                // one and from there the aggregated row (a slot >= cnt re-reads the point's own last row with weight 0: 0 x inf is
                // NaN, still not finite).  One sum of magnitudes per thread says whether all 16 values are finite.
                float mag = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j) mag += __builtin_fabsf(out[j]);
                if (a.status && not_finite_bits(mag)) atomicOr(a.status, kShadeNonfinitePairs);
            }
        }
        NPCD_STS(12);
        __syncthreads();            // the tile's rows, weights and counts are free for the next prologue
        NPCD_STS(13);
#ifdef NPCD_SHADE_TL
        span_tiles += 1;
        if (tl_on) {
            tl_n += 1; tl_nb += nblk;
            if (nblk == 4) {
                tl_n4 += 1;
                for (int i = 1; i < 14; ++i) tl_sum[i] += tl[i] - tl[i - 1];
            }
        }
#endif
        tile = ntile;
    }
#ifdef NPCD_SHADE_TL
    if (tl_on && lane == 0) {
        for (int i = 0; i < 14; ++i) g_shade_tl[i] = tl_sum[i];
        g_shade_tl[14] = tl_n4; g_shade_tl[15] = tl_n; g_shade_tl[16] = __builtin_amdgcn_s_memtime() - tl_begin; g_shade_tl[17] = tl_nb;
    }
    if (tid == 0 && blockIdx.x < 512) {
        long long* sp = g_shade_span + 4 * blockIdx.x;
        // (tiles | HW_ID << 16 | XCC_ID << 48: which CU / XCD the workgroup ran on)
        const long long hw = (long long)(unsigned)__builtin_amdgcn_s_getreg(63492), xcc = (long long)(__builtin_amdgcn_s_getreg(6164) & 15);
        sp[0] = rt_begin; sp[1] = __builtin_amdgcn_s_memrealtime(); sp[2] = span_tiles | (hw << 16) | (xcc << 48); sp[3] = __builtin_amdgcn_s_memtime() - tl_begin;
    }
#endif
}
#ifdef NPCD_SHADE_TL
}
namespace npcd { int rows_debug_read(long long* out, int count); }
extern "C" int npcd_shade_span_read(long long* out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(npcd::g_shade_span), sizeof(long long) * count);
}
extern "C" int npcd_shade_debug_read(long long* out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(npcd::g_shade_tl), sizeof(long long) * count);
}
extern "C" int npcd_shade_rows_debug_read(long long* out, int count) { return npcd::rows_debug_read(out, count); }
namespace npcd {
#endif

// ============================================================================================
// kernel A, eight-wave form: the same tile on 512 threads
// ============================================================================================
// Why (docs/experiments.md R5.12): with two workgroups of four waves a SIMD holds TWO waves, each alternating between a matrix phase
// and a phase without matrix instructions (write-back, barriers, prologue, aggregation, the start of a layer's weight stream).  The
// counters say a wave is issuing 27 % of its cycles and no unit is saturated (matrix pipe 0.51, LDS ~0.3, vector L1 ~0.3): the tile is
// a dependency chain, and two chains per SIMD leave the matrix pipe idle whenever both are outside their matrix phases.  Here a wave
// owns 32 output channels instead of 64 (accumulators 64 registers instead of 128, <= 128 registers per wave): two workgroups of
// EIGHT waves put four chains on a SIMD.  The price: every wave reads all activation fragments of its tile, so the LDS read volume of
// the layers doubles (~1.8 MB per tile = ~14 k clocks of the CU's LDS pipe against ~12 k of matrix pipe).  Same instructions per
// output element in the same order: results are bit-identical to the four-wave form.
struct WRing8 { f16x8 a[4]; };
template <int KSTEPS>
__device__ __forceinline__ f16x8 wfrag_load8(wrsrc_t rs, int w_off, int wave, int lane, int step) {
    const int so = w_off + wave * (KSTEPS * kFragBytes) + (step >> 2) * (4 * kFragBytes);
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + (step & 3) * kFragBytes, so, 0));
}
template <int KSTEPS, int PF>
__device__ __forceinline__ void layer_prefetch8(wrsrc_t rs, int w_off, int b_off, int wave, int lane, WRing8& ring, f32x16& init) {
    const int hh = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, hh * 32 + (g >> 1) * 64 + (g & 1) * 16, b_off + wave * 128, 0));
#pragma unroll
        for (int b = 0; b < 4; ++b) init[4 * g + b] = b4[b];
    }
#pragma unroll
    for (int d = 0; d < PF && d < KSTEPS; ++d) ring.a[d] = wfrag_load8<KSTEPS>(rs, w_off, wave, lane, d);
}
// per k-step: [weight fragment of step s + PF] fence [NB matrix instructions] fence [activation fragments of step s + 1, into the
// registers the matrix instructions just read: one set -- the other three waves of the SIMD cover the LDS latency]
template <int KSTEPS, int NB, int PF>
__device__ __forceinline__ void layer_mfma8(const unsigned char* H, wrsrc_t rs, int w_off, int wave, int lane, WRing8& ring, const f32x16& init,
                                            f32x16 (&acc)[4]) {
    static_assert(PF >= 1 && PF <= 3, "ring of four");
    const int r = lane & 31, hh = lane >> 5;
    const unsigned char* hb = H + r * kRowBytes + hh * 16;
    f16x8 bc[NB];
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) bc[cb] = *reinterpret_cast<const f16x8*>(hb + cb * 32 * kRowBytes);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        if (s + PF < KSTEPS) ring.a[(s + PF) & 3] = wfrag_load8<KSTEPS>(rs, w_off, wave, lane, s + PF);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) acc[cb] = F16::mfma32(ring.a[s & 3], bc[cb], s == 0 ? init : acc[cb]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < KSTEPS) {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) bc[cb] = *reinterpret_cast<const f16x8*>(hb + cb * 32 * kRowBytes + (s + 1) * 32);
        }
    }
}
template <bool ACT, int NB>
__device__ __forceinline__ void layer_store8(unsigned char* H, int wave, int lane, const f32x16 (&acc)[4]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const int r = lane & 31, hh = lane >> 5;
    unsigned char* sb = H + r * kRowBytes + hh * 16 + wave * 64;      // row (cb*32 + r), channels 32 wave + 16 gp + 8 hh .. + 7
#pragma unroll
    for (int cb = 0; cb < NB; ++cb)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            uint32_t v[2][2];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int e = 4 * (2 * gp + gg) + 2 * b;
                    const f32x2 f = {acc[cb][e], acc[cb][e + 1]};
                    f16x2 h = __builtin_convertvector(f, f16x2);
                    if (ACT) {
                        const f16x2 sc = {(_Float16)kLeaky, (_Float16)kLeaky};
                        h = __builtin_elementwise_max(h, h * sc);
                    }
                    v[gg][b] = __builtin_bit_cast(uint32_t, h);
                }
            *reinterpret_cast<u32x4*>(sb + cb * 32 * kRowBytes + gp * 32) = u32x4{v[0][0], v[0][1], v[1][0], v[1][1]};
        }
}
#ifndef NPCD_PAIRS8_PF
#define NPCD_PAIRS8_PF 2
#endif
template <int FEAT, int NB>
__device__ __forceinline__ void pair_layers8(unsigned char* H, const PairPack& pk, int wave, int lane) {
    constexpr int K0 = FEAT + kEncBlock, PF = NPCD_PAIRS8_PF;
    constexpr int kLayerBytes = 8 * (kHidden / 16) * kFragBytes;
    f32x16 acc[4];
    WRing8 ring;
    f32x16 init;
    layer_prefetch8<K0 / 16, PF>(pk.rs, 0, pk.b0, wave, lane, ring, init);
    layer_mfma8<K0 / 16, NB, PF>(H, pk.rs, 0, wave, lane, ring, init, acc);
    layer_prefetch8<kHidden / 16, PF>(pk.rs, pk.w1, pk.b0 + kHidden * 4, wave, lane, ring, init);
    __syncthreads();
    layer_store8<true, NB>(H, wave, lane, acc);
    __syncthreads();
#pragma unroll 1
    for (int l = 1; l < 4; ++l) {
        const int w_off = pk.w1 + (l - 1) * kLayerBytes, b_off = pk.b0 + l * (kHidden * 4);
        layer_mfma8<kHidden / 16, NB, PF>(H, pk.rs, w_off, wave, lane, ring, init, acc);
        if (l < 3) layer_prefetch8<kHidden / 16, PF>(pk.rs, w_off + kLayerBytes, b_off + kHidden * 4, wave, lane, ring, init);
        __syncthreads();
        layer_store8<true, NB>(H, wave, lane, acc);
        __syncthreads();
    }
}
template <int FEAT>
struct PairInputs8 {
    int gi, gi_other;
    float pt[3], kp[3];
    f32x4 feat[FEAT / 16];             // this quarter's FEAT / 4 neighbour features
};
template <int FEAT>
__device__ __forceinline__ void pair_load_indices8(const ShadeArgs& a, int tile, int P, int tid, PairInputs8<FEAT>& in) {
    tid = opaque(tid);
    const int row = tid & 127, slot = row & 7;
    const int p = tile * 16 + (row >> 3), po = tile * 16 + ((row ^ 64) >> 3);
    in.gi = -1;
    in.gi_other = -1;
    if (p < P && slot < a.k) in.gi = a.nb_idx[(int64_t)p * a.k + slot];
    if (po < P && slot < a.k) in.gi_other = a.nb_idx[(int64_t)po * a.k + slot];
}
template <int FEAT>
__device__ __forceinline__ void pair_load_data8(const ShadeArgs& a, int tile, int tid, PairInputs8<FEAT>& in) {
    const int row = opaque(tid) & 127, part = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int p = min(tile * 16 + (row >> 3), a.max_points - 1), g = max(in.gi, 0);
#pragma unroll
    for (int c = 0; c < 3; ++c) { in.pt[c] = a.pts[(int64_t)p * 3 + c]; in.kp[c] = a.kp_pos[(int64_t)g * 3 + c]; }
    const float* fp = a.kp_feat + (int64_t)g * FEAT + part * (FEAT / 4);
#pragma unroll
    for (int c4 = 0; c4 < FEAT / 16; ++c4) in.feat[c4] = *reinterpret_cast<const f32x4*>(fp + c4 * 4);
}
// the 16 positional-encoding columns 16 PART .. 16 PART + 15 of a row (two 16-byte chunks)
template <int FEAT, int PART>
__device__ __forceinline__ void enc_put16(unsigned char* H, int prow, const float (&rel)[3]) {
#pragma unroll
    for (int c8 = 0; c8 < 2; ++c8) {
        f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)enc_value(PART * 16 + c8 * 8 + j, rel);
        *reinterpret_cast<f16x8*>(H + act_off(prow, FEAT / 8 + PART * 2 + c8)) = v;
    }
}

template <int FEAT>
__global__ __launch_bounds__(512, 4) void shade_pairs8_kernel(ShadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    unsigned char* H = dsmem;
    float* wrow = reinterpret_cast<float*>(dsmem + kRows * kRowBytes);
    int* pstart = reinterpret_cast<int*>(wrow + kRows);
    int* pcount = pstart + 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const ShadeLayout L = shade_layout(FEAT);
    const PairPack pk = {pack_rsrc(a.wpack, L.total), (int)L.w[1], (int)L.bias[0]};
    const int P = min(*a.n_points, a.max_points);
    const int ntiles = (P + 15) / 16;
    int* next_tile = pcount + 16;
    int tile = blockIdx.x;
    PairInputs8<FEAT> in;
    if (tile < ntiles) {
        pair_load_indices8<FEAT>(a, tile, P, tid, in);
        pair_load_data8<FEAT>(a, tile, tid, in);
    }
    while (tile < ntiles) {
        if (tid == 0) *next_tile = a.tile_counter ? (int)gridDim.x + atomicAdd(a.tile_counter, 1) : tile + (int)gridDim.x;
        int nblk;
        {
            // a wave = 64 candidates of one QUARTER of the input columns: rows (wave & 1) * 64 .., quarter wave >> 1
            const int row = opaque(tid) & 127, part = __builtin_amdgcn_readfirstlane(tid >> 7);
            const int slot = row & 7;
            const int gi = in.gi;
            const unsigned long long mine = __ballot(gi >= 0), other = __ballot(in.gi_other >= 0);
            const int n_mine = __popcll(mine), n_other = __popcll(other);
            const int before = __popcll(mine & ((1ull << lane) - 1ull)) + ((row & 64) ? n_other : 0);
            const int V = n_mine + n_other;
            nblk = (V + 31) >> 5;
            const int prow = before;
            if (part == 0 && slot == 0) {
                pstart[row >> 3] = before;
                pcount[row >> 3] = __popcll((mine >> (lane & ~7)) & 0xffull);
            }
            float rel[3] = {0.f, 0.f, 0.f};
            if (gi >= 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) rel[c] = in.pt[c] - in.kp[c];
            }
            if (part == 0 && gi >= 0) wrow[prow] = 1.f / (sqrtf(__builtin_fmaf(rel[0], rel[0], __builtin_fmaf(rel[1], rel[1], rel[2] * rel[2]))) + 1e-5f);
            constexpr int FQ = FEAT / 4;
            if (gi >= 0) {
#pragma unroll
                for (int c8 = 0; c8 < FQ / 8; ++c8) {
                    f16x8 v;
                    const f32x4 x0 = in.feat[2 * c8], x1 = in.feat[2 * c8 + 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = (_Float16)x0[j]; v[4 + j] = (_Float16)x1[j]; }
                    *reinterpret_cast<f16x8*>(H + act_off(prow, part * (FQ / 8) + c8)) = v;
                }
                switch (part) {        // (wave-uniform)
                    case 0: enc_put16<FEAT, 0>(H, prow, rel); break;
                    case 1: enc_put16<FEAT, 1>(H, prow, rel); break;
                    case 2: enc_put16<FEAT, 2>(H, prow, rel); break;
                    default: enc_put16<FEAT, 3>(H, prow, rel); break;
                }
            }
            if (row >= V && row < 32 * nblk) {
                f16x8 z;
#pragma unroll
                for (int j = 0; j < 8; ++j) z[j] = (_Float16)0.f;
#pragma unroll
                for (int c8 = 0; c8 < FQ / 8; ++c8) *reinterpret_cast<f16x8*>(H + act_off(row, part * (FQ / 8) + c8)) = z;
#pragma unroll
                for (int c8 = 0; c8 < 2; ++c8) *reinterpret_cast<f16x8*>(H + act_off(row, FEAT / 8 + part * 2 + c8)) = z;
            }
        }
        __syncthreads();
        const int ntile = *next_tile;
        if (ntile < ntiles) pair_load_indices8<FEAT>(a, ntile, P, tid, in);
        switch (nblk) {
            case 4: pair_layers8<FEAT, 4>(H, pk, wave, lane); break;
            case 3: pair_layers8<FEAT, 3>(H, pk, wave, lane); break;
            case 2: pair_layers8<FEAT, 2>(H, pk, wave, lane); break;
            case 1: pair_layers8<FEAT, 1>(H, pk, wave, lane); break;
            default: break;
        }
        pair_load_data8<FEAT>(a, min(ntile, ntiles - 1), tid, in);
        __builtin_amdgcn_sched_barrier(0);
        {
            // point, and its channels 8 cc .. + 7: the 32 lanes of a point read the 32 chunks of a row
            const int ta = opaque(tid), pl = ta >> 5, cc = ta & 31;
            const int p = tile * 16 + pl;
            const int r0 = pstart[pl], cnt = pcount[pl];
            const int last = max(cnt - 1, 0);
            float w8[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) w8[s2] = wrow[r0 + min(s2, last)];
            f16x8 v0[8];
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) v0[s2] = *reinterpret_cast<const f16x8*>(H + act_off(r0 + min(s2, last), cc));
            float wsum = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                w8[s2] = s2 < cnt ? w8[s2] : 0.f;
                wsum += w8[s2];
            }
            const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
            float out[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) {
                const float ws = w8[s2] * inv;
#pragma unroll
                for (int j = 0; j < 8; ++j) out[j] = __builtin_fmaf(ws, (float)v0[s2][j], out[j]);
            }
            if (cnt == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) out[j] = 0.f;
            }
            if (p < P) {
                f16x8 o0;
#pragma unroll
                for (int j = 0; j < 8; ++j) o0[j] = (_Float16)out[j];
                *reinterpret_cast<f16x8*>(a.G + (int64_t)p * kHidden + cc * 8) = o0;
                float mag = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) mag += __builtin_fabsf(out[j]);
                if (a.status && not_finite_bits(mag)) atomicOr(a.status, kShadeNonfinitePairs);
            }
        }
        __syncthreads();
        tile = ntile;
    }
}

// ============================================================================================
// kernel B: points.  tile = 128 points
// ============================================================================================
__device__ __forceinline__ float softplus_m1(float x) {
    x -= 1.f;
    return x > 20.f ? x : log1pf(expf(x));  // F.softplus(beta=1, threshold=20)
}

#ifndef NPCD_POINTS_PF
#define NPCD_POINTS_PF 2
#endif
#ifdef NPCD_POINTS_TL          // DIAGNOSTIC: s_memtime stamps of the first pass of workgroup NPCD_POINTS_TL, wave 0
__device__ long long g_points_tl[32];
#define NPCD_PTS(i) do { if (blockIdx.x == NPCD_POINTS_TL && tid == 0 && g_points_tl[31] == 0) g_points_tl[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define NPCD_PTS(i) do { } while (0)
#endif
// A point-level layer: the pinned pipeline of the pair kernel (published configuration), or -- with view directions, whose kernel has
// no registers to spare for the ring and the prefetched bias -- the plain form with the weights three k-steps ahead.
template <bool PIPE, int PF>
__device__ __forceinline__ void point_prefetch(wrsrc_t rs, int w_off, int b_off, int wave, int lane, WRing& ring, f32x16 (&init)[2]) {
    if constexpr (PIPE) layer_prefetch<16, PF>(rs, w_off, b_off, wave, lane, ring, init);
}
template <bool PIPE, int NB, int PF>
__device__ __forceinline__ void point_layer(const unsigned char* H, const unsigned char* wpack, wrsrc_t rs, int w_off, int b_off, int wave, int lane,
                                            WRing& ring, f32x16 (&init)[2], f32x16 (&acc)[2][4]) {
    if constexpr (PIPE) layer_mfma_pipe<16, NB, PF>(H, rs, w_off, wave, lane, ring, init, acc);
    else layer_mfma<16, 4, NB, 3>(H, wpack + w_off, reinterpret_cast<const float*>(wpack + b_off), wave, lane, acc);
}
// One pass of the point-level layers over NB 32-row blocks (rows row0 .. row0 + 32 NB of the compact lists).
struct PointsArgs {          // what a pass needs of ShadeArgs, passed BY VALUE (a reference to the kernel argument would be a scratch copy)
    const unsigned char* wpack;
    const _Float16* G;
    float *sigma, *rgb;
    const float* dir_bias;
    const int32_t* point_ray;
    int32_t* status;
    int feat_dim;
    int red_rows;    // rows per wave plane of the reduction buffer (32 x the kernel's largest pass)
};
// (inlined four times into the range loop.  The thread index is made opaque per pass: otherwise the lane-dependent LDS / weight
// addresses of all four variants are hoisted out of the loop and held in registers across it -- 64-92 spilled registers; as a
// real function call instead, the LDS pointers become generic and every LDS access a flat instruction)
template <bool DIR, int NB>
__device__ __forceinline__ void points_pass(PointsArgs a, unsigned char* H, float* red, int row0, int P, int tid) {
    asm volatile("" : "+v"(tid));
    const ShadeLayout L = shade_layout(a.feat_dim);
    const int lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const float* s1 = reinterpret_cast<const float*>(a.wpack + L.s1);
    const float* c4 = reinterpret_cast<const float*>(a.wpack + L.c4);
    // the six matrices A4, S0, C0..C3 and their biases lie back to back in the pack; every layer's bias and first weight fragments
    // are requested before the previous layer's epilogue (layer_prefetch / layer_mfma_pipe, as in the pair kernel)
    constexpr bool PIPE = !DIR;
    constexpr int PF = NPCD_POINTS_PF, kLayerBytes = 8 * (kHidden / 16) * kFragBytes;
    const wrsrc_t rs = pack_rsrc(a.wpack, L.total);
    const int wv = __builtin_amdgcn_readfirstlane(wave), w4 = (int)L.w[4], b4 = (int)L.bias[4];
    WRing ring;
    f32x16 init[2];
    point_prefetch<PIPE, PF>(rs, w4, b4, wv, lane, ring, init);
    NPCD_PTS(0);
    // ---- load the aggregated features of 32 NB points ---------------------------------------
#pragma unroll
    for (int it = 0; it < 4 * NB; ++it) {
        const int cidx = it * 256 + tid, row = cidx >> 5, chunk = cidx & 31;
        const int p = row0 + row;
        u32x4 v = {0, 0, 0, 0};
        if (p < P) v = *reinterpret_cast<const u32x4*>(a.G + (int64_t)p * kHidden + chunk * 8);
        *reinterpret_cast<u32x4*>(H + act_off(row, chunk)) = v;
    }
    __syncthreads();
    NPCD_PTS(1);
    f32x16 acc[2][4];
    // ---- last aggregator layer (linear): feat ------------------------------------------
    point_layer<PIPE, NB, PF>(H, a.wpack, rs, w4, b4, wv, lane, ring, init, acc);
    point_prefetch<PIPE, PF>(rs, w4 + kLayerBytes, b4 + kHidden * 4, wv, lane, ring, init);
    NPCD_PTS(2);
    __syncthreads();
    NPCD_PTS(3);
    layer_store<false, NB>(H, wave, lane, acc);
    NPCD_PTS(4);
    __syncthreads();
    NPCD_PTS(5);
    // ---- density head: Linear(256,256) + LeakyReLU + Linear(256,1), softplus(x - 1) ------
    point_layer<PIPE, NB, PF>(H, a.wpack, rs, w4 + kLayerBytes, b4 + kHidden * 4, wv, lane, ring, init, acc);
    point_prefetch<PIPE, PF>(rs, w4 + 2 * kLayerBytes, b4 + 2 * kHidden * 4, wv, lane, ring, init);
    {
        float part[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) part[cb] = 0.f;
#pragma unroll
        for (int oi = 0; oi < 2; ++oi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(s1 + (2 * wave + oi) * 32 + acc_group_off(g, hh));
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb) {
                        float x = acc[oi][cb][4 * g + b];
                        x = x > 0.f ? x : kLeaky * x;
                        part[cb] += x * w4[b];
                    }
            }
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const float t = part[cb] + swap_half(part[cb]);
            if (hh == 0) red[(wave * a.red_rows + cb * 32 + r) * 4 + 3] = t;
        }
    }
    NPCD_PTS(6);
    // (H still holds feat: the density pass did not write activations)
    // ---- colour head: 4 x [Linear(256,256) + LeakyReLU] + Linear(256,3), sigmoid ----------
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const int w_off = w4 + (2 + l) * kLayerBytes, b_off = b4 + (2 + l) * (kHidden * 4);
        point_layer<PIPE, NB, PF>(H, a.wpack, rs, w_off, b_off, wv, lane, ring, init, acc);
        if (l < 3) point_prefetch<PIPE, PF>(rs, w_off + kLayerBytes, b_off + kHidden * 4, wv, lane, ring, init);
        if (DIR && l == 0) {       // + the view-direction part of the first colour layer (per ray, fp32)
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
                const int p = row0 + cb * 32 + r;
                const float* db = a.dir_bias + (int64_t)(p < P ? a.point_ray[p] : 0) * kHidden;
#pragma unroll
                for (int oi = 0; oi < 2; ++oi)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(db + (2 * wave + oi) * 32 + acc_group_off(g, hh));
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[oi][cb][4 * g + b] += v[b];
                    }
            }
        }
        NPCD_PTS(7 + 4 * l);
        if (l < 3) {
            __syncthreads();
            NPCD_PTS(8 + 4 * l);
            layer_store<true, NB>(H, wave, lane, acc);
            NPCD_PTS(9 + 4 * l);
            __syncthreads();
            NPCD_PTS(10 + 4 * l);
        }
    }
    {
        // (a channel's three projection weights are loaded once and applied to all NB row blocks)
        float pc[NB][3];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) pc[cb][0] = pc[cb][1] = pc[cb][2] = 0.f;
#pragma unroll
        for (int oi = 0; oi < 2; ++oi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float* cp = c4 + (2 * wave + oi) * 32 + acc_group_off(g, hh);
                const f32x4 wr = *reinterpret_cast<const f32x4*>(cp), wg = *reinterpret_cast<const f32x4*>(cp + kHidden),
                            wb = *reinterpret_cast<const f32x4*>(cp + 2 * kHidden);
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int cb = 0; cb < NB; ++cb) {
                        float x = acc[oi][cb][4 * g + b];
                        x = x > 0.f ? x : kLeaky * x;
                        pc[cb][0] += x * wr[b];
                        pc[cb][1] += x * wg[b];
                        pc[cb][2] += x * wb[b];
                    }
            }
#pragma unroll
        for (int cb = 0; cb < NB; ++cb) {
            const float pr = pc[cb][0] + swap_half(pc[cb][0]), pg = pc[cb][1] + swap_half(pc[cb][1]), pb = pc[cb][2] + swap_half(pc[cb][2]);
            if (hh == 0) {
                float* q = red + (wave * a.red_rows + cb * 32 + r) * 4;
                q[0] = pr; q[1] = pg; q[2] = pb;
            }
        }
    }
    NPCD_PTS(21);
    __syncthreads();
    if (tid < 32 * NB) {
        const int p = row0 + tid;
        if (p < P) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(red + (w * a.red_rows + tid) * 4);
                t += v;
            }
            // range guard: the heads' hidden activations are fp16 between layers; an overflow there arrives here as inf / NaN in the
            // final pre-activation -- where sigmoid(+inf) = 1 would hide it
            if (a.status && not_finite_bits(__builtin_fabsf(t[0]) + __builtin_fabsf(t[1]) + __builtin_fabsf(t[2]) + __builtin_fabsf(t[3])))
                atomicOr(a.status, kShadeNonfiniteHeads);
            a.sigma[p] = softplus_m1(t[3] + s1[kHidden]);
#pragma unroll
            for (int c = 0; c < 3; ++c) a.rgb[(int64_t)p * 3 + c] = 1.f / (1.f + expf(-(t[c] + c4[3 * kHidden + c])));
        }
    }
    __syncthreads();
    NPCD_PTS(20);
    NPCD_PTS(31);
}

// The same pass on v_mfma_f32_16x16x32_f16 (round 5, R5.13; the layer functions of the pair kernel's 16x16x32 form): a lane holds row
// rb * 16 + (l & 15) and, per channel block mb, the four channels 64 wave + 32 (mb >> 1) + 8 g + 4 (mb & 1) + b (g = l >> 4) -- the heads'
// dot products run over those and are summed over the four lane groups g with two lane swaps, then over the waves through `red`.
// One form for both configurations: the view-direction variant has the registers for the pipelined layers here.
__device__ __forceinline__ float sum_lane_groups(float x) {       // x summed over lanes l, l ^ 16, l ^ 32, l ^ 48
    const auto r16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(r16[0]) + __uint_as_float(r16[1]);
    const auto r32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r32[0]) + __uint_as_float(r32[1]);
}
template <bool DIR, int NB>
__device__ __forceinline__ void points_pass16(PointsArgs a, unsigned char* H, float* red, const unsigned char* bias_lds, int row0, int P, int tid) {
    asm volatile("" : "+v"(tid));
    constexpr int NB16 = 2 * NB;
    const ShadeLayout L = shade_layout(a.feat_dim);
    const int lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const float* s1 = reinterpret_cast<const float*>(a.wpack + L.s1);
    const float* c4 = reinterpret_cast<const float*>(a.wpack + L.c4);
    constexpr int kLayerBytes = kHidden * kHidden * 2;
    const wrsrc_t rs = pack_rsrc(a.wpack, L.total);
    const int wv = __builtin_amdgcn_readfirstlane(wave), w4 = (int)L.w16[4];
    WRing16 ring;
    layer_prefetch16<8>(rs, w4, wv, lane, ring);
#pragma unroll
    for (int it = 0; it < 4 * NB; ++it) {
        const int cidx = it * 256 + tid, row = cidx >> 5, chunk = cidx & 31;
        const int p = row0 + row;
        u32x4 v = {0, 0, 0, 0};
        if (p < P) v = *reinterpret_cast<const u32x4*>(a.G + (int64_t)p * kHidden + chunk * 8);
        *reinterpret_cast<u32x4*>(H + act_off(row, chunk)) = v;
    }
    __syncthreads();
    f32x4a acc[4][8];
    // ---- last aggregator layer (linear): feat ------------------------------------------
    layer_mfma16<8, NB16>(H, bias_lds, rs, w4, wv, lane, ring, acc);
    layer_prefetch16<8>(rs, w4 + kLayerBytes, wv, lane, ring);
    __syncthreads();
    layer_store16<false, NB16>(H, wave, lane, acc);
    __syncthreads();
    // ---- density head: Linear(256,256) + LeakyReLU + Linear(256,1), softplus(x - 1) ------
    layer_mfma16<8, NB16>(H, bias_lds + kHidden * 4, rs, w4 + kLayerBytes, wv, lane, ring, acc);
    layer_prefetch16<8>(rs, w4 + 2 * kLayerBytes, wv, lane, ring);
    {
        float part[NB16];
#pragma unroll
        for (int rb = 0; rb < NB16; ++rb) part[rb] = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const f32x4 wv4 = *reinterpret_cast<const f32x4*>(s1 + 64 * wave + 32 * (mb >> 1) + 8 * g + 4 * (mb & 1));
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int rb = 0; rb < NB16; ++rb) {
                    float x = acc[mb][rb][b];
                    x = x > 0.f ? x : kLeaky * x;
                    part[rb] += x * wv4[b];
                }
        }
#pragma unroll
        for (int rb = 0; rb < NB16; ++rb) {
            const float t = sum_lane_groups(part[rb]);
            if (g == 0) red[(wave * a.red_rows + rb * 16 + n) * 4 + 3] = t;
        }
    }
    // (H still holds feat: the density pass did not write activations)
    // ---- colour head: 4 x [Linear(256,256) + LeakyReLU] + Linear(256,3), sigmoid ----------
#pragma unroll 1
    for (int l = 0; l < 4; ++l) {
        const int w_off = w4 + (2 + l) * kLayerBytes;
        layer_mfma16<8, NB16>(H, bias_lds + (2 + l) * (kHidden * 4), rs, w_off, wv, lane, ring, acc);
        if (l < 3) layer_prefetch16<8>(rs, w_off + kLayerBytes, wv, lane, ring);
        if (DIR && l == 0) {       // + the view-direction part of the first colour layer (per ray, fp32)
#pragma unroll
            for (int rb = 0; rb < NB16; ++rb) {
                const int p = row0 + rb * 16 + n;
                const float* db = a.dir_bias + (int64_t)(p < P ? a.point_ray[p] : 0) * kHidden;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(db + 64 * wave + 32 * (mb >> 1) + 8 * g + 4 * (mb & 1));
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[mb][rb][b] += v[b];
                }
            }
        }
        if (l < 3) {
            __syncthreads();
            layer_store16<true, NB16>(H, wave, lane, acc);
            __syncthreads();
        }
    }
    {
        float pc[NB16][3];
#pragma unroll
        for (int rb = 0; rb < NB16; ++rb) pc[rb][0] = pc[rb][1] = pc[rb][2] = 0.f;
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const float* cp = c4 + 64 * wave + 32 * (mb >> 1) + 8 * g + 4 * (mb & 1);
            const f32x4 wr = *reinterpret_cast<const f32x4*>(cp), wg = *reinterpret_cast<const f32x4*>(cp + kHidden),
                        wb = *reinterpret_cast<const f32x4*>(cp + 2 * kHidden);
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int rb = 0; rb < NB16; ++rb) {
                    float x = acc[mb][rb][b];
                    x = x > 0.f ? x : kLeaky * x;
                    pc[rb][0] += x * wr[b];
                    pc[rb][1] += x * wg[b];
                    pc[rb][2] += x * wb[b];
                }
        }
#pragma unroll
        for (int rb = 0; rb < NB16; ++rb) {
            const float pr = sum_lane_groups(pc[rb][0]), pg = sum_lane_groups(pc[rb][1]), pb = sum_lane_groups(pc[rb][2]);
            if (g == 0) {
                float* q = red + (wave * a.red_rows + rb * 16 + n) * 4;
                q[0] = pr; q[1] = pg; q[2] = pb;
            }
        }
    }
    __syncthreads();
    if (tid < 32 * NB) {
        const int p = row0 + tid;
        if (p < P) {
            f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(red + (w * a.red_rows + tid) * 4);
                t += v;
            }
            if (a.status && not_finite_bits(__builtin_fabsf(t[0]) + __builtin_fabsf(t[1]) + __builtin_fabsf(t[2]) + __builtin_fabsf(t[3])))
                atomicOr(a.status, kShadeNonfiniteHeads);
            a.sigma[p] = softplus_m1(t[3] + s1[kHidden]);
#pragma unroll
            for (int c = 0; c < 3; ++c) a.rgb[(int64_t)p * 3 + c] = 1.f / (1.f + expf(-(t[c] + c4[3 * kHidden + c])));
        }
    }
    __syncthreads();
}

// kernel B: the point-level layers.  The compact points are cut into 32-row blocks and the blocks are dealt out EVENLY:
// workgroup i takes the contiguous range [i q + min(i, rem), ...) of q or q + 1 blocks and walks it in passes of at most four blocks
// (128 rows), the passes of a range as equal as possible (5 blocks = 3 + 2).  (Until round 3 the grid strode over 128-row tiles:
// 584 tiles of a 128^2 view on 512 resident workgroups are two rounds for 1.14 rounds of work.)
template <bool DIR, int FORM = 0>          // FORM 0: 32x32x16 layers, 1: 16x16x32 layers (points_pass16)
__global__ __launch_bounds__(256, 2) void shade_points_kernel(ShadeArgs a) {
    constexpr int MAXNB = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    unsigned char* H = dsmem;
    float* red = reinterpret_cast<float*>(dsmem + MAXNB * 32 * kRowBytes);  // [4 waves][128 rows][4]
    const int tid = threadIdx.x;
    unsigned char* bias_lds = reinterpret_cast<unsigned char*>(red + 4 * 32 * MAXNB * 4);   // FORM 1: the six layers' biases (6 KiB)
    if constexpr (FORM == 1) {
        const ShadeLayout L = shade_layout(a.feat_dim);
        for (int i = tid; i < 6 * kHidden / 4; i += 256)
            reinterpret_cast<f32x4*>(bias_lds)[i] = *reinterpret_cast<const f32x4*>(a.wpack + L.bias16[4] + (int64_t)i * 16);
        __syncthreads();
    }
    if (blockIdx.x == 0 && tid == 0 && a.tile_counter) *a.tile_counter = 0;      // the pair kernel of this call is done: its ticket word is free again
    const PointsArgs pa{a.wpack, a.G, a.sigma, a.rgb, a.dir_bias, a.point_ray, a.status, a.feat_dim, 32 * MAXNB};
    const int P = min(*a.n_points, a.max_points);
    const int nblk = (P + 31) >> 5, q = nblk / (int)gridDim.x, rem = nblk % (int)gridDim.x;
    int b = (int)blockIdx.x * q + min((int)blockIdx.x, rem);
    int left = q + ((int)blockIdx.x < rem ? 1 : 0);
    while (left > 0) {
        const int passes = (left + MAXNB - 1) / MAXNB, nb = (left + passes - 1) / passes;
        if constexpr (FORM == 1) {
            if (nb >= 4) points_pass16<DIR, 4>(pa, H, red, bias_lds, b * 32, P, tid);
            else if (nb == 3) points_pass16<DIR, 3>(pa, H, red, bias_lds, b * 32, P, tid);
            else if (nb == 2) points_pass16<DIR, 2>(pa, H, red, bias_lds, b * 32, P, tid);
            else points_pass16<DIR, 1>(pa, H, red, bias_lds, b * 32, P, tid);
        } else {
        if (nb >= 4) points_pass<DIR, 4>(pa, H, red, b * 32, P, tid);
        else if (nb == 3) points_pass<DIR, 3>(pa, H, red, b * 32, P, tid);
        else if (nb == 2) points_pass<DIR, 2>(pa, H, red, b * 32, P, tid);
        else points_pass<DIR, 1>(pa, H, red, b * 32, P, tid);
        }
        b += nb;
        left -= nb;
    }
}

}  // namespace npcd

using namespace npcd;

#ifdef NPCD_POINTS_TL
extern "C" int npcd_points_debug_read(long long* out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(npcd::g_points_tl), sizeof(long long) * count);
}
#endif

static int shade_check(int feat_dim, int n_freqs, int hidden) {
    if (hidden != kHidden || n_freqs != kNFreqs) return NPCD_ERR_UNSUPPORTED;
    if (feat_dim != 32 && feat_dim != 128) return NPCD_ERR_UNSUPPORTED;
    return NPCD_OK;
}

extern "C" int64_t npcd_shade_wpack_bytes(int feat_dim, int n_freqs, int hidden) {
    if (shade_check(feat_dim, n_freqs, hidden) != NPCD_OK) return -1;
    return shade_layout(feat_dim).total;
}

extern "C" int64_t npcd_shade_workspace_bytes(int max_points, int hidden) {
    if (hidden != kHidden || max_points < 0) return -1;
    return (int64_t)(max_points + kRows) * kHidden * 2 + shade_rows_workspace_bytes(max_points) + 16;      // + the tile counter
}

// fragment order: [out block ob][k-step s][lane][8]  with  element = W[ob*32 + acc_channel(lane&31)][16 s + 8 (lane>>5) + j]
static void pack_matrix(const float* W, int out_dim, int in_dim, int k_padded, unsigned char* dst) {
    _Float16* d = reinterpret_cast<_Float16*>(dst);
    const int ksteps = k_padded / 16;
    for (int ob = 0; ob < out_dim / 32; ++ob)
        for (int s = 0; s < ksteps; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int o = ob * 32 + acc_channel(lane & 31), c = 16 * s + 8 * (lane >> 5) + j;
                    const float v = c < in_dim ? W[(int64_t)o * in_dim + c] : 0.f;
                    d[(((int64_t)ob * ksteps + s) * 64 + lane) * 8 + j] = (_Float16)v;
                }
}

// the 16x16x32 form's order (pair_layers16): [wave][32-column step s][block mb][lane][8] with
//   element = W[64 wave + 32 (mb >> 1) + 8 g + 4 (mb & 1) + b][32 s + 8 (lane >> 4) + j],  4 g + b = lane & 15
static void pack_matrix16(const float* W, int in_dim, int k_padded, unsigned char* dst) {
    _Float16* d = reinterpret_cast<_Float16*>(dst);
    const int ks = k_padded / 32;
    for (int w = 0; w < 4; ++w)
        for (int s = 0; s < ks; ++s)
            for (int mb = 0; mb < 4; ++mb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int m = lane & 15, o = 64 * w + 32 * (mb >> 1) + 8 * (m >> 2) + 4 * (mb & 1) + (m & 3), c = 32 * s + 8 * (lane >> 4) + j;
                        const float v = c < in_dim ? W[(int64_t)o * in_dim + c] : 0.f;
                        d[((((int64_t)w * ks + s) * 4 + mb) * 64 + lane) * 8 + j] = (_Float16)v;
                    }
}
static void pack_bias16(const float* bias, unsigned char* dst) {        // [wave][g][mb][b]
    float* d = reinterpret_cast<float*>(dst);
    for (int w = 0; w < 4; ++w)
        for (int g = 0; g < 4; ++g)
            for (int mb = 0; mb < 4; ++mb)
                for (int b = 0; b < 4; ++b) d[((w * 4 + g) * 4 + mb) * 4 + b] = bias[64 * w + 32 * (mb >> 1) + 8 * g + 4 * (mb & 1) + b];
}

// The slab stream of the rows kernel (shade_rows.hip): layers A0..A3, each as 4 quarters (output blocks 2 q, 2 q + 1) x steps of
// two k-steps; a slab = fragments (k-step, block) = (0,0) (0,1) (1,0) (1,1).  Input column of k-slot (s, g = lane >> 5, e):
//   hidden layers: 32 (s/2) + 8 (2 (s%2) + e/4) + 4 g + e%4   (an accumulator tile's values 8 jp .. 8 jp + 7 ARE fragment 2 mb + jp)
//   layer 0: the feature channels in order, then per pair t = 8 s2 + e < 30 of (coordinate, frequency) the sine on g = 0 and the
//   cosine on g = 1, then x, y (g = 0) and z, zero (g = 1)
static void pack_rows_stream(const float* const* W, int feat_dim, int k0, unsigned char* dst) {
    _Float16* d = reinterpret_cast<_Float16*>(dst);
    const int in0 = feat_dim + 3 + 6 * kNFreqs;
    int64_t frag = 0;
    for (int l = 0; l < 4; ++l) {
        const int ks = (l == 0 ? k0 : kHidden) / 16, in_dim = l == 0 ? in0 : kHidden;
        for (int q = 0; q < 4; ++q)
            for (int i = 0; i < ks / 2; ++i)
                for (int sl = 0; sl < 2; ++sl)
                    for (int m = 0; m < 2; ++m, ++frag)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int o = 32 * (2 * q + m) + (lane & 31), g = lane >> 5, s = 2 * i + sl;
                                int c;
                                if (l > 0) c = 32 * (s / 2) + 8 * (2 * (s % 2) + e / 4) + 4 * g + e % 4;
                                else if (s < feat_dim / 16) c = 16 * s + 8 * g + e;
                                else {
                                    const int t = 8 * (s - feat_dim / 16) + e;
                                    const int qq = t < 30 ? 3 + 20 * (t / 10) + (g ? 10 : 0) + t % 10 : (t == 30 ? (g ? 2 : 0) : (g ? 63 : 1));
                                    c = feat_dim + qq;
                                }
                                const float v = c < in_dim ? W[l][(int64_t)o * in_dim + c] : 0.f;
                                d[(frag * 64 + lane) * 8 + e] = (_Float16)v;
                            }
    }
}

// weights_host / biases_host: 12 pointers in the order
//   aggregator.local_field.{0,2,4,6,8}, shape_net.{0,2}, channel_net.{0,2,4,6,8}   
extern "C" int npcd_shade_pack_weights(const float* const* weights_host, const float* const* biases_host, int feat_dim, int n_freqs,
                                       int hidden, void* wpack_host) {
    int rc = shade_check(feat_dim, n_freqs, hidden);
    if (rc != NPCD_OK) return rc;
    if (!weights_host || !biases_host || !wpack_host) return NPCD_ERR_ARG;
    for (int i = 0; i < 12; ++i)
        if (!weights_host[i] || !biases_host[i]) return NPCD_ERR_ARG;
    const ShadeLayout L = shade_layout(feat_dim);
    unsigned char* out = static_cast<unsigned char*>(wpack_host);
    memset(out, 0, L.total);
    const int in0 = feat_dim + 3 + 6 * kNFreqs;
    // packed slot -> source index: A0..A3 = 0..3, A4 = 4, S0 = 5, C0..C3 = 7..10
    const int src[10] = {0, 1, 2, 3, 4, 5, 7, 8, 9, 10};
    for (int i = 0; i < 10; ++i) {
        pack_matrix(weights_host[src[i]], kHidden, i == 0 ? in0 : kHidden, i == 0 ? L.k0 : kHidden, out + L.w[i]);
        memcpy(out + L.bias[i], biases_host[src[i]], kHidden * 4);
    }
    pack_rows_stream(weights_host, feat_dim, L.k0, out + L.rows);
    for (int i = 0; i < 10; ++i) {
        pack_matrix16(weights_host[src[i]], i == 0 ? in0 : kHidden, i == 0 ? L.k0 : kHidden, out + L.w16[i]);
        pack_bias16(biases_host[src[i]], out + L.bias16[i]);
    }
    float* s1 = reinterpret_cast<float*>(out + L.s1);
    memcpy(s1, weights_host[6], kHidden * 4);
    s1[kHidden] = biases_host[6][0];
    float* c4 = reinterpret_cast<float*>(out + L.c4);
    memcpy(c4, weights_host[11], 3 * kHidden * 4);
    memcpy(c4 + 3 * kHidden, biases_host[11], 3 * 4);
    return NPCD_OK;
}

// Tile tickets of the pair kernel: library-owned words (zero at load), one per call in flight, handed out round-robin; the point
// kernel of the same call -- behind the pair kernel in stream order -- sets its word back to zero.  (Rounds 1-3: a word of the
// caller's workspace, zeroed by a fill launch per call: 4.5 us of a 390-us view.)  64 calls may be in flight on different streams.
namespace npcd { __device__ int32_t g_tile_tickets[64]; }
static int32_t* tile_ticket_slot() {
    static std::atomic<unsigned> next{0};
    static std::atomic<int32_t*> base[DynLds::kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DynLds::kMaxDevices) return nullptr;
    int32_t* b = base[dev].load(std::memory_order_acquire);
    if (!b) {
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&b), HIP_SYMBOL(npcd::g_tile_tickets)) != hipSuccess) return nullptr;
        base[dev].store(b, std::memory_order_release);
    }
    return b + (next.fetch_add(1) & 63u);
}

static int shade_points_launch(const void* wpack, int feat_dim, int n_freqs, int hidden, const int32_t* nb_idx, const float* pts,
                               const float* kp_pos, const float* kp_feat, const int32_t* n_points_dev, int max_points, int k,
                               float* sigma, float* rgb, void* workspace, const float* dir_bias, const int32_t* point_ray, int32_t* status,
                               void* stream) {
    int rc = shade_check(feat_dim, n_freqs, hidden);
    if (rc != NPCD_OK) return rc;
    if (!wpack || !nb_idx || !pts || !kp_pos || !kp_feat || !n_points_dev || !sigma || !rgb || !workspace) return NPCD_ERR_ARG;
    if ((dir_bias != nullptr) != (point_ray != nullptr) || (reinterpret_cast<uintptr_t>(dir_bias) & 15)) return NPCD_ERR_ARG;
    if (k <= 0 || k > 8) return NPCD_ERR_UNSUPPORTED;
    if (max_points <= 0) return NPCD_OK;
    ShadeArgs a{};
    a.wpack = static_cast<const unsigned char*>(wpack);
    a.feat_dim = feat_dim; a.k = k;
    a.nb_idx = nb_idx; a.pts = pts; a.kp_pos = kp_pos; a.kp_feat = kp_feat;
    a.n_points = n_points_dev;
    a.max_points = max_points;
    a.G = static_cast<_Float16*>(workspace);
    a.sigma = sigma; a.rgb = rgb;
    a.dir_bias = dir_bias; a.point_ray = point_ray;
    a.status = status;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const bool static_tiles = getenv("NPCD_SHADE_STATIC_TILES") != nullptr;
    if (!static_tiles) a.tile_counter = tile_ticket_slot();      // (nullptr: tiles strided over the grid)
    const int ldsA = kRows * kRowBytes + kRows * 4 + 32 * 4 + 16;   // activations, row weights, per-point packed-row ranges, next tile
    const int ldsA16 = ldsA + kRows * (kRowBytes16 - kRowBytes) + 4 * kHidden * 4;      // 16x16x32 form: wider pitch + the four layers' biases
    const int ldsB = kRows * (kRowBytes + 4 * 4 * 4);
    static DynLds lds_a32, lds_a128, lds_b, lds_bd;
    NPCD_HIP_CHECK(lds_a32.ensure(reinterpret_cast<const void*>(shade_pairs_kernel<32>), ldsA));
    NPCD_HIP_CHECK(lds_a128.ensure(reinterpret_cast<const void*>(shade_pairs_kernel<128>), ldsA));
    static DynLds lds_a8_32, lds_a8_128;
    NPCD_HIP_CHECK(lds_a8_32.ensure(reinterpret_cast<const void*>(shade_pairs8_kernel<32>), ldsA));
    NPCD_HIP_CHECK(lds_a8_128.ensure(reinterpret_cast<const void*>(shade_pairs8_kernel<128>), ldsA));
    static DynLds lds_a16_32, lds_a16_128;
    NPCD_HIP_CHECK(lds_a16_32.ensure(reinterpret_cast<const void*>(shade_pairs_kernel<32, 1>), ldsA16));
    NPCD_HIP_CHECK(lds_a16_128.ensure(reinterpret_cast<const void*>(shade_pairs_kernel<128, 1>), ldsA16));
    const int ldsB16 = ldsB + 6 * kHidden * 4;                       // + the six point-level layers' biases: 80 KiB, two workgroups = the CU's LDS
    static DynLds lds_b16, lds_bd16;
    NPCD_HIP_CHECK(lds_b16.ensure(reinterpret_cast<const void*>(shade_points_kernel<false, 1>), ldsB16));
    NPCD_HIP_CHECK(lds_bd16.ensure(reinterpret_cast<const void*>(shade_points_kernel<true, 1>), ldsB16));
    const char* points16_env = getenv("NPCD_SHADE_POINTS16");             // read per call (A/B in one process)
    const bool points16 = !(points16_env && points16_env[0] == '0');      // default since round 5; 0 = the 32x32x16 layers
    const char* pairs16_env = getenv("NPCD_SHADE_PAIRS16");               // read per call (A/B in one process)
    const bool pairs16 = !(pairs16_env && pairs16_env[0] == '0');         // the default since round 5 (R5.13); 0 = the 32x32x16 layers
    const char* pairs8_env = getenv("NPCD_SHADE_PAIRS8");                 // read per call (A/B in one process)
    const bool pairs8 = pairs8_env && pairs8_env[0] == '1';
    NPCD_HIP_CHECK(lds_b.ensure(reinterpret_cast<const void*>(shade_points_kernel<false>), ldsB));
    NPCD_HIP_CHECK(lds_bd.ensure(reinterpret_cast<const void*>(shade_points_kernel<true>), ldsB));
    // persistent-style grids: 2 workgroups per CU, tiles strided over the grid; the tile count is
    // read from device memory so that no host round trip is needed after the neighbour query
    const int tilesA = (max_points + 15) / 16, tilesB = (max_points + 31) / 32;       // B: 32-row blocks, dealt out evenly
    const int gridA = tilesA < 512 ? tilesA : 512, gridB = tilesB < 512 ? tilesB : 512;
    // kernel A: LDS tiles of 16 points (below), or with NPCD_SHADE_ROWS=1 the rows form (shade_rows.hip: activations in registers;
    // opt-in: as fast, but its matrix-product aggregation adds a point's rows in an order that depends on where the point sits in
    // the compact lists, so two renders agree to fp16 rounding instead of bit for bit -- DESIGN.md 5.3).  Read per call.
    const char* rows_env = getenv("NPCD_SHADE_ROWS");
    if (rows_env && rows_env[0] == '1') {
        rc = shade_rows_launch(a, static_cast<unsigned char*>(workspace) + (int64_t)(max_points + kRows) * kHidden * 2, st);
        if (rc != NPCD_OK) return rc;
    } else if (pairs8) {
        if (feat_dim == 32) hipLaunchKernelGGL(shade_pairs8_kernel<32>, dim3(gridA), dim3(512), ldsA, st, a);
        else hipLaunchKernelGGL(shade_pairs8_kernel<128>, dim3(gridA), dim3(512), ldsA, st, a);
    } else if (pairs16) {
        if (feat_dim == 32) hipLaunchKernelGGL((shade_pairs_kernel<32, 1>), dim3(gridA), dim3(256), ldsA16, st, a);
        else hipLaunchKernelGGL((shade_pairs_kernel<128, 1>), dim3(gridA), dim3(256), ldsA16, st, a);
    } else if (feat_dim == 32) hipLaunchKernelGGL(shade_pairs_kernel<32>, dim3(gridA), dim3(256), ldsA, st, a);
    else hipLaunchKernelGGL(shade_pairs_kernel<128>, dim3(gridA), dim3(256), ldsA, st, a);
    if (points16) {
        if (dir_bias) hipLaunchKernelGGL((shade_points_kernel<true, 1>), dim3(gridB), dim3(256), ldsB16, st, a);
        else hipLaunchKernelGGL((shade_points_kernel<false, 1>), dim3(gridB), dim3(256), ldsB16, st, a);
    } else if (dir_bias) hipLaunchKernelGGL(shade_points_kernel<true>, dim3(gridB), dim3(256), ldsB, st, a);
    else hipLaunchKernelGGL(shade_points_kernel<false>, dim3(gridB), dim3(256), ldsB, st, a);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_shade_points(const void* wpack, int feat_dim, int n_freqs, int hidden, const int32_t* nb_idx, const float* pts,
                                 const float* kp_pos, const float* kp_feat, const int32_t* n_points_dev, int max_points, int k,
                                 float* sigma, float* rgb, void* workspace, int32_t* status, void* stream) {
    return shade_points_launch(wpack, feat_dim, n_freqs, hidden, nb_idx, pts, kp_pos, kp_feat, n_points_dev, max_points, k, sigma, rgb,
                               workspace, nullptr, nullptr, status, stream);
}
extern "C" int npcd_shade_points_dir(const void* wpack, int feat_dim, int n_freqs, int hidden, const int32_t* nb_idx, const float* pts,
                                     const float* kp_pos, const float* kp_feat, const int32_t* n_points_dev, int max_points, int k,
                                     float* sigma, float* rgb, void* workspace, const float* dir_bias, const int32_t* point_ray,
                                     int32_t* status, void* stream) {
    if (!dir_bias || !point_ray) return NPCD_ERR_ARG;
    return shade_points_launch(wpack, feat_dim, n_freqs, hidden, nb_idx, pts, kp_pos, kp_feat, n_points_dev, max_points, k, sigma, rgb,
                               workspace, dir_bias, point_ray, status, stream);
}
