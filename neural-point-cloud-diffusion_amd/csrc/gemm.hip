// Weight-gradient GEMM of the denoiser's Linear layers for gfx950:  dW[N, K] (fp32) = dy[T, N]^T x[T, K],  16-bit operands.
//
// Reference: the nn.Linear weight gradients of transformer.py:67-72 (c_qkv), :107-115 (attn.c_proj), :118-137 (mlp.c_fc / c_proj),
// produced by autograd in the reference; here called from the hand-written backward of the fused backbone
// (npcd/models/diffusion/fused.py).  24 layers x 4 Linears x 2 T N K FLOP = 19.8 TFLOP of the 64.5 TFLOP step.
//
// Why an own kernel: the product reduces over the TOKEN dimension (T = 32,832 at cfg-D) into a small output (1-4 M elements =
// 16-64 tiles of 256 x 256 for 256 CUs), both operands have the reduction index as their SLOW index, and the result must be fp32.
// The library runs this at 0.74-0.98 PF/s (row-split batched GEMM + a sum pass; its fp32-output kernels are outside TunableOp).
//
// What bounds it (measured, tools/probes/gpu_dev_wgrad_own.py + diagnostic builds -DNPCD_WGRAD_DIAG=1..4, c_fc shape, T = 32,832):
// a 256 x 256 tile streams 128 FLOP per byte through LDS whatever the split, i.e. 2.15 GB per call; the LDS-DMA stream ALONE (no
// fragment reads, no matrix instructions) takes 170-190 us = 49 GB/s per CU, which is what 63 % L2 hits (70 GB/s per CU) and 37 %
// misses to the Infinity Cache / HBM (30 GB/s) give (rocprofv3: TCP_TCC_READ_REQ 2.15 GB, FETCH_SIZE x 2 = 788 MB -- the minimum for
// eight private L2s: every XCD needs its share of dy once and all of x).  The matrix instructions alone take ~190 us; the two overlap
// to 255-275 us -- which is also where the library is (MT256x256x32 / 256x192x64 kernels + a sum pass: 0.92-1.0 PF/s).  The kernel is
// at parity with the library, not ahead; it is an opt-in (NPCD_OWN_WGRAD=1) that makes the weight gradients bitwise reproducible
// without a library dependency.
//
// Structure:
//   * workgroup = 8 waves = one 256 x 256 output tile over ONE slice of the token range (split-T: tiles x slices ~ 256 workgroups,
//     one per CU); every slice writes its own fp32 slab, npcd_wgrad_reduce adds the slabs in slice order (bitwise reproducible, no
//     atomics);
//   * both operands stream as [32 tokens][64 columns] sub-tiles (128-byte rows, the XOR-swizzled image of the attention kernels)
//     through a four-stage LDS ring by LDS-DMA (one stage read, three in flight, counted vmcnt); BOTH MFMA operands are transposed reads (ds_read_b64_tr_b16) of those row-major
//     images -- the contraction order is the same permutation of the 16 token rows for A and B, so it drops out;
//   * wave = 128 (N) x 64 (K) of the tile: 8 accumulators of 32 x 32, v_mfma_f32_32x32x16; the fragment reads of a stage's first
//     half are issued right behind the barrier that says it has landed and hide behind the previous stage's second half;
//   * workgroup -> (tile, slice) mapping keeps the workgroups that read the same dy panel on one XCD (its L2 then serves the
//     panel to all of them; x is small enough for the Infinity Cache).
#include <type_traits>

#include "common.h"

namespace npcd {

typedef __attribute__((address_space(3))) void lptr_g;

__device__ __forceinline__ uint32_t g_lds_addr(const unsigned char* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char*)p;
}
__device__ __forceinline__ const void* g_uniform_ptr(const void* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
}
// one LDS-DMA wave-instruction: wave-uniform 64-bit base + one 32-bit byte offset per lane -> LDS (M0 = wave-uniform destination)
__device__ __forceinline__ void g_dma16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    sbase = g_uniform_ptr(sbase);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int OFF>
__device__ __forceinline__ u32x2 g_tr(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
struct GFrag {
    u32x2 lo, hi;
};
template <class TR>
__device__ __forceinline__ typename TR::vec8 g_vec(const GFrag& t) {
    const u32x4 x = {t.lo[0], t.lo[1], t.hi[0], t.hi[1]};
    return __builtin_bit_cast(typename TR::vec8, x);
}

struct WgradParams {
    const void* dy;   // [T, N]
    const void* x;    // [T, K]
    float* out;       // [S][N][K] slabs (S == 1: the result itself)
    int T, N, K, S;
    int tiles_n, tiles_k;
};
#ifndef NPCD_WGRAD_DIAG
#define NPCD_WGRAD_DIAG 0      // DIAGNOSTIC builds (wrong results, timing only): 1 no DMA inside the loop, 2 no matrix instructions, 3 no slab stores,
                               // 4 neither matrix instructions nor fragment reads (the DMA stream + its waits and barriers alone)
#endif

constexpr int kStage = 32768;          // one ring stage = 32 tokens: 4 dy sub-tiles + 4 x sub-tiles of [32 tokens][64 columns] = 4 KiB each
constexpr int kSub = 4096;
constexpr int kRing = 4;               // stages: one being read, up to three in flight

// the 4 DMA pieces (8 rows x 128 B each) of one [32 token][64 column] sub-tile; rows past T come from a page of zeros
template <class E>
__device__ __forceinline__ void dma_subtile(uint32_t lds_dst, const E* base, int64_t ld, int t0, int T, int col0, int lane, const E* zeros) {
    const int rowin = lane >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int prow = i * 8 + rowin;
        const int chunk = (lane & 7) ^ tile_swz(prow);
        if (t0 + i * 8 + 8 <= T) {                                  // (wave-uniform) all eight rows exist
            const char* sb = reinterpret_cast<const char*>(base + (int64_t)(t0 + i * 8) * ld + col0);
            g_dma16(sb, (uint32_t)((rowin * ld + chunk * 8) * (int64_t)sizeof(E)), __builtin_amdgcn_readfirstlane(lds_dst + i * 1024));
        } else {
            // the ragged end of the token range: per-lane choice between the row and the zero page (64-bit per-lane address form)
            const E* src = (t0 + prow < T) ? base + (int64_t)(t0 + prow) * ld + col0 + chunk * 8 : zeros + (lane & 7) * 8;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_dst + i * 1024)) : "memory");
        }
    }
}

__device__ unsigned char g_zero_page[128];      // (zero-initialised device memory)

// (Tried: touching the 32 cache lines of a sub-tile a few stages ahead of its LDS-DMA with one plain load per wave and stage, so that
//  the DMA finds them in the XCD's L2.  c_fc 269 -> 360 us at every distance tried (4, 6, 10 stages): vmcnt retires in order, and the
//  far-ahead loads that miss the L2 hold the counted waits of the ring behind them.  Removed.)
// one 256 x 256 output tile at (n0, k0) over the ring stages [s_lo, s_hi) of the token range, written to `out` (row pitch p.K)
template <class TR>
__device__ __forceinline__ void wgrad_tile(const WgradParams& p, float* out, int n0, int k0, int s_lo, int s_hi) {
    using E = typename TR::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // kRing stages x 32 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                                   // wave tile: rows [128 wm, +128) x columns [64 wn, +64)
    const E* dy = static_cast<const E*>(p.dy);
    const E* xx = static_cast<const E*>(p.x);
    const E* zeros = reinterpret_cast<const E*>(g_zero_page);
    const uint32_t lds0 = g_lds_addr(smem);
    // this wave's share of a stage's DMA: sub-tile `wave` (waves 0..3: dy columns n0 + 64 wave; waves 4..7: x columns k0 + 64 (wave - 4))
    const bool isx = wave >= 4;
    const E* dbase = isx ? xx : dy;
    const int64_t dld = isx ? p.K : p.N;
    const int dcol = isx ? k0 + 64 * (wave - 4) : n0 + 64 * wave;
    const uint32_t ddst = lds0 + wave * kSub;

    // fragment addresses inside a sub-tile (stage / sub-tile / 16-row step go into immediates or one add)
    uint32_t tr[2][2];
    {
        const int grp = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3, h = grp >> 1;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            const int col = db * 32 + 16 * (grp & 1) + 4 * pp;
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) tr[db][hi] = lds0 + tile_off(4 * h + q + 8 * hi, col >> 3) + (col & 7) * 2;
        }
    }
    const uint32_t a_off = (2 * wm) * kSub, b_off = (4 + wn) * kSub;

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = f32x16{0};

    // ring protocol: stage st is read while st + 1 .. st + 3 are in flight; the slot of st + 3 is the one of st - 1, free after the
    // barrier that ended iteration st - 1.  A wave issues 4 DMA instructions per stage: "stage st + 1 has landed" = at most the 8
    // instructions of the two younger stages outstanding.
#define NPCD_G_VMWAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
    if (s_lo < s_hi) {
        dma_subtile<E>(ddst, dbase, dld, s_lo * 32, p.T, dcol, lane, zeros);
        if (s_lo + 1 < s_hi) dma_subtile<E>(ddst + kStage, dbase, dld, (s_lo + 1) * 32, p.T, dcol, lane, zeros);
        if (s_lo + 2 < s_hi) dma_subtile<E>(ddst + 2 * kStage, dbase, dld, (s_lo + 2) * 32, p.T, dcol, lane, zeros);
        if (s_lo + 2 < s_hi) NPCD_G_VMWAIT(8);
        else if (s_lo + 1 < s_hi) NPCD_G_VMWAIT(4);
        else NPCD_G_VMWAIT(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // Software pipeline over the stages: the fragments of a stage's FIRST 16 tokens are read at the end of the previous
    // iteration, right behind the barrier that says the stage has landed, and their latency hides behind the matrix instructions
    // of the previous stage's second half; the second half's fragments are read behind the first half's matrix instructions.
    GFrag fa0[4] = {}, fb0[2] = {}, fa1[4] = {}, fb1[2] = {};
#define NPCD_G_ISSUE(FA, FB, SO, G16)                                                                     \
    do {                                                                                                  \
        if (NPCD_WGRAD_DIAG == 4) break;                                                                  \
        FA[0].lo = g_tr<(G16) * 2048>(tr[0][0] + (SO) + a_off);        FA[0].hi = g_tr<(G16) * 2048>(tr[0][1] + (SO) + a_off);        \
        FA[1].lo = g_tr<(G16) * 2048>(tr[1][0] + (SO) + a_off);        FA[1].hi = g_tr<(G16) * 2048>(tr[1][1] + (SO) + a_off);        \
        FA[2].lo = g_tr<kSub + (G16) * 2048>(tr[0][0] + (SO) + a_off); FA[2].hi = g_tr<kSub + (G16) * 2048>(tr[0][1] + (SO) + a_off); \
        FA[3].lo = g_tr<kSub + (G16) * 2048>(tr[1][0] + (SO) + a_off); FA[3].hi = g_tr<kSub + (G16) * 2048>(tr[1][1] + (SO) + a_off); \
        FB[0].lo = g_tr<(G16) * 2048>(tr[0][0] + (SO) + b_off);        FB[0].hi = g_tr<(G16) * 2048>(tr[0][1] + (SO) + b_off);        \
        FB[1].lo = g_tr<(G16) * 2048>(tr[1][0] + (SO) + b_off);        FB[1].hi = g_tr<(G16) * 2048>(tr[1][1] + (SO) + b_off);        \
    } while (0)
#define NPCD_G_MMA(FA, FB)                                                                                 \
    do {                                                                                                  \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                  \
            _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                              \
                if (NPCD_WGRAD_DIAG != 2 && NPCD_WGRAD_DIAG != 4) acc[mi][ni] = TR::mfma32(g_vec<TR>(FA[mi]), g_vec<TR>(FB[ni]), acc[mi][ni]);   \
                else if (NPCD_WGRAD_DIAG == 2) acc[mi][ni][0] += __uint_as_float(FA[mi].lo[0] ^ FB[ni].hi[1]);          \
    } while (0)
#define NPCD_G_WAIT()                                       \
    do {                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                  \
    } while (0)
    if (s_lo < s_hi) NPCD_G_ISSUE(fa0, fb0, 0u, 0);
    for (int st = s_lo; st < s_hi; ++st) {
        const uint32_t so = ((st - s_lo) & (kRing - 1)) * kStage, so1 = ((st - s_lo + 1) & (kRing - 1)) * kStage;
        if (st + 3 < s_hi && NPCD_WGRAD_DIAG != 1)
            dma_subtile<E>(ddst + (((st - s_lo + 3) & (kRing - 1)) * kStage), dbase, dld, (st + 3) * 32, p.T, dcol, lane, zeros);
        NPCD_G_WAIT();                                   // first-half fragments of stage st
        NPCD_G_ISSUE(fa1, fb1, so, 1);
        NPCD_G_MMA(fa0, fb0);
        NPCD_G_WAIT();                                   // second-half fragments: this wave has read everything it needs of stage st
        // stage st + 1 must have landed (the younger ones stay in flight); every wave is done reading stage st
        if (st + 3 < s_hi) NPCD_G_VMWAIT(8);
        else if (st + 2 < s_hi) NPCD_G_VMWAIT(4);
        else NPCD_G_VMWAIT(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 1 < s_hi) NPCD_G_ISSUE(fa0, fb0, so1, 0);
        NPCD_G_MMA(fa1, fb1);
    }
#undef NPCD_G_ISSUE
#undef NPCD_G_MMA
#undef NPCD_G_WAIT
#undef NPCD_G_VMWAIT
    // ---- row n0 + 128 wm + 32 mi + acc_row(i, hh), column k0 + 64 wn + 32 ni + (lane & 31)
    const int r = lane & 31, hh = lane >> 5;
    if (NPCD_WGRAD_DIAG == 3 && acc[0][0][0] != 12345.678f) return;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            float* o = out + (int64_t)(n0 + 128 * wm + 32 * mi) * p.K + k0 + 64 * wn + 32 * ni + r;
#pragma unroll
            for (int i = 0; i < 16; ++i) o[(int64_t)acc_row(i, hh) * p.K] = acc[mi][ni][i];
        }
}

template <class TR>
__global__ __launch_bounds__(512, 2) void wgrad_kernel(WgradParams p) {
    // workgroup -> (n tile, slice, k tile): k fastest, so that the workgroups of one XCD chunk share dy panels
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    const int tk = w % p.tiles_k, s = (w / p.tiles_k) % p.S, tn = w / (p.tiles_k * p.S);
    const int steps = (p.T + 31) >> 5;                                         // ring stages of 32 tokens
    const int s_lo = (int)((int64_t)steps * s / p.S), s_hi = (int)((int64_t)steps * (s + 1) / p.S);
    wgrad_tile<TR>(p, p.out + (int64_t)s * p.N * p.K, tn * 256, tk * 256, s_lo, s_hi);      // the slab of this slice
}

// Several weight gradients over the SAME token range in one launch (the four Linear layers of a residual block at the token count of
// one rank of the 8-GPU job): every 256 x 256 output tile of every product is ONE workgroup over the whole token range -- no slices,
// no slabs, no second kernel, one fixed summation order per element.  4,104 tokens x (3072 + 1024 + 4096 + 1024) x 1024: 192 tiles,
// i.e. one round on 192 of the 256 CUs; the launch runs on a side stream beside the backward's critical path, which takes the rest.
constexpr int kWgradGroupMax = 8;
struct WgradGroup {
    const void* dy[kWgradGroupMax];
    const void* x[kWgradGroupMax];
    float* out[kWgradGroupMax];
    int N[kWgradGroupMax], K[kWgradGroupMax];
    int first[kWgradGroupMax + 1];      // first tile of product g in the launch's tile order
    int T, count;
};

template <class TR>
__global__ __launch_bounds__(512, 2) void wgrad_group_kernel(WgradGroup gp) {
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    int g = 0;
#pragma unroll
    for (int i = 1; i < kWgradGroupMax; ++i)
        if (i < gp.count && w >= gp.first[i]) g = i;
    WgradParams p;
    p.dy = gp.dy[g]; p.x = gp.x[g]; p.out = gp.out[g];
    p.T = gp.T; p.N = gp.N[g]; p.K = gp.K[g]; p.S = 1;
    p.tiles_n = p.N / 256; p.tiles_k = p.K / 256;
    const int t = w - gp.first[g];
    wgrad_tile<TR>(p, p.out, (t / p.tiles_k) * 256, (t % p.tiles_k) * 256, 0, (gp.T + 31) >> 5);
}

// out[i] = slab[0][i] + slab[1][i] + ... in slice order (fp32, 16 bytes per thread and trip)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const f32x4* __restrict__ slabs, f32x4* __restrict__ out, int S, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 a = __builtin_nontemporal_load(slabs + i);
        for (int s = 1; s < S; ++s) a += __builtin_nontemporal_load(slabs + s * n4 + i);
        out[i] = a;
    }
}

}  // namespace npcd

using namespace npcd;

static int wgrad_slices(int T, int N, int K) {
    const int tiles = (N / 256) * (K / 256);
    static const int forced = [] { const char* e = getenv("NPCD_WGRAD_SLICES"); return e ? atoi(e) : 0; }();     // A/B probes only
    int S = forced > 0 ? forced : 256 / tiles;
    const int steps = (T + 31) / 32;
    if (S > steps / 16) S = steps / 16;        // at least 16 ring stages (512 tokens) per slice
    if (S > 16) S = 16;
    return S < 1 ? 1 : S;
}

extern "C" int npcd_wgrad_slices(int T, int N, int K) {
    if (T <= 0 || N <= 0 || K <= 0 || N % 256 || K % 256) return -1;
    return wgrad_slices(T, N, K);
}

// dW [N, K] fp32 = dy[T, N]^T x[T, K].  workspace: npcd_wgrad_slices(T, N, K) slabs of N K floats (unused when that is 1).
extern "C" int npcd_wgrad(const void* dy, const void* x, float* out, float* workspace, int T, int N, int K, int dtype, void* stream) {
    if (!dy || !x || !out || T <= 0 || N <= 0 || K <= 0) return NPCD_ERR_ARG;
    if (N % 256 || K % 256 || (dtype != NPCD_BF16 && dtype != NPCD_F16)) return NPCD_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dy) & 15) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return NPCD_ERR_ARG;
    const int S = wgrad_slices(T, N, K);
    if (S > 1 && (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15))) return NPCD_ERR_ARG;
    WgradParams p{};
    p.dy = dy; p.x = x; p.out = S > 1 ? workspace : out;
    p.T = T; p.N = N; p.K = K; p.S = S;
    p.tiles_n = N / 256; p.tiles_k = K / 256;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int grid = p.tiles_n * p.tiles_k * S;
    static DynLds lds_b, lds_h;
    if (dtype == NPCD_BF16) {
        NPCD_HIP_CHECK(lds_b.ensure(reinterpret_cast<const void*>(wgrad_kernel<BF16>), kRing * kStage));
        hipLaunchKernelGGL(wgrad_kernel<BF16>, dim3(grid), dim3(512), kRing * kStage, st, p);
    } else {
        NPCD_HIP_CHECK(lds_h.ensure(reinterpret_cast<const void*>(wgrad_kernel<F16>), kRing * kStage));
        hipLaunchKernelGGL(wgrad_kernel<F16>, dim3(grid), dim3(512), kRing * kStage, st, p);
    }
    if (S > 1) {
        const int64_t n4 = (int64_t)N * K / 4;
        const int rgrid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), dim3(256), 0, st, reinterpret_cast<const f32x4*>(workspace), reinterpret_cast<f32x4*>(out), S, n4);
    }
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

// count (<= 8) weight gradients dW_g [N_g, K_g] fp32 = dy_g[T, N_g]^T x_g[T, K_g] over one token range in ONE launch: one workgroup per
// 256 x 256 output tile over all T tokens (no slices, no workspace); meant for token counts of a few thousand, where the tiles of all
// products together are about one round of the chip.  Same shape rules as npcd_wgrad per product.
extern "C" int npcd_wgrad_group(int count, const void* const* dy, const void* const* x, float* const* out, const int* N, const int* K, int T,
                                int dtype, void* stream) {
    if (count < 1 || count > kWgradGroupMax || !dy || !x || !out || !N || !K || T <= 0) return NPCD_ERR_ARG;
    if (dtype != NPCD_BF16 && dtype != NPCD_F16) return NPCD_ERR_UNSUPPORTED;
    WgradGroup gp{};
    gp.T = T; gp.count = count;
    int tiles = 0;
    for (int g = 0; g < count; ++g) {
        if (!dy[g] || !x[g] || !out[g] || N[g] <= 0 || K[g] <= 0) return NPCD_ERR_ARG;
        if (N[g] % 256 || K[g] % 256) return NPCD_ERR_UNSUPPORTED;
        if ((reinterpret_cast<uintptr_t>(dy[g]) & 15) || (reinterpret_cast<uintptr_t>(x[g]) & 15) || (reinterpret_cast<uintptr_t>(out[g]) & 15)) return NPCD_ERR_ARG;
        gp.dy[g] = dy[g]; gp.x[g] = x[g]; gp.out[g] = out[g]; gp.N[g] = N[g]; gp.K[g] = K[g];
        gp.first[g] = tiles;
        tiles += (N[g] / 256) * (K[g] / 256);
    }
    gp.first[count] = tiles;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static DynLds lds_b, lds_h;
    if (dtype == NPCD_BF16) {
        NPCD_HIP_CHECK(lds_b.ensure(reinterpret_cast<const void*>(wgrad_group_kernel<BF16>), kRing * kStage));
        hipLaunchKernelGGL(wgrad_group_kernel<BF16>, dim3(tiles), dim3(512), kRing * kStage, st, gp);
    } else {
        NPCD_HIP_CHECK(lds_h.ensure(reinterpret_cast<const void*>(wgrad_group_kernel<F16>), kRing * kStage));
        hipLaunchKernelGGL(wgrad_group_kernel<F16>, dim3(tiles), dim3(512), kRing * kStage, st, gp);
    }
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
