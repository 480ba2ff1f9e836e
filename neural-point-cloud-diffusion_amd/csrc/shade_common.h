// Definitions shared by the shading kernels (shade.hip: tiles of 16 points in LDS, the point-level heads; shade_rows.hip: the
// per-pair layers with the activations in registers).
#pragma once
#include "common.h"

namespace npcd {

constexpr int kHidden = 256;
constexpr int kNFreqs = 10;
constexpr int kEncBlock = 64;          // 3 + 60 positional-encoding columns + 1 zero pad
constexpr int kRows = 128;             // rows per tile
constexpr int kRowBytes = kHidden * 2 + 16; // 528: rows padded by one 16-byte chunk instead of an XOR swizzle (below)
constexpr int kFragBytes = 1024;       // one 32(out) x 16(in) fp16 weight fragment
constexpr float kLeaky = 0.01f;

// Output channel (within a 32-row block) of accumulator row m = 8 g + 4 hh + b of a 32 x 32 tile (g = register group, hh = lane half,
// b = register of the group).  The matrices' rows are PERMUTED in the pack so that the register groups g = 2 q and 2 q + 1 of a lane
// are 8 CONSECUTIVE channels -- a whole 16-byte chunk of the activation row, written back without cross-lane traffic (in the natural
// order m = channel, lanes l and l + 32 held the halves of a chunk: one v_permlane32_swap per dword, 32 per wave and layer).
__host__ __device__ inline int acc_channel(int m) {
    const int g = m >> 3, hh = (m >> 2) & 1, b = m & 3;
    return 16 * (g >> 1) + 8 * hh + 4 * (g & 1) + b;
}
// float offset of register group g of lane half hh inside a 32-channel block of a per-channel vector (bias, head weights)
__host__ __device__ inline int acc_group_off(int g, int hh) { return 16 * (g >> 1) + 8 * hh + 4 * (g & 1); }

struct ShadeLayout {
    int k0;          // padded input width of layer 0
    int64_t w[10];   // byte offsets of the packed matrices: A0..A3, A4, S0, C0..C3
    int64_t bias[10];
    int64_t s1, c4;  // fp32 vectors: s1 = [256 w | 1 b | pad], c4 = [3*256 w | 3 b | pad]
    int64_t rows;    // A0..A3 once more, as the slab stream of the rows kernel (shade_rows.hip)
    int rows_slabs;  // its length in 4-KiB slabs: 4 quarters x (k0 / 32 + 3 * 8)
    // all ten matrices and their biases once more, in the fragment order of the 16x16x32 forms of the pair and point kernels
    // (shade.hip, pack_matrix16 / pack_bias16; the bias blocks lie back to back: copied to LDS in one run)
    int64_t w16[10], bias16[10];
    int64_t total;
};
__host__ __device__ inline ShadeLayout shade_layout(int feat_dim) {
    ShadeLayout L;
    L.k0 = feat_dim + kEncBlock;
    int64_t off = 0;
    for (int i = 0; i < 10; ++i) {
        L.w[i] = off;
        const int ksteps = (i == 0 ? L.k0 : kHidden) / 16;
        off += (int64_t)8 * ksteps * kFragBytes;
    }
    for (int i = 0; i < 10; ++i) { L.bias[i] = off; off += kHidden * 4; }
    L.s1 = off; off += 264 * 4;
    L.c4 = off; off += 776 * 4;
    off = (off + 4095) / 4096 * 4096;
    L.rows = off;
    L.rows_slabs = 4 * (L.k0 / 32 + 3 * (kHidden / 32));
    off += (int64_t)L.rows_slabs * 4096;
    for (int i = 0; i < 10; ++i) { L.w16[i] = off; off += (int64_t)kHidden * (i == 0 ? L.k0 : kHidden) * 2; }
    for (int i = 0; i < 10; ++i) { L.bias16[i] = off; off += kHidden * 4; }
    L.total = off;
    return L;
}

struct ShadeArgs {
    const unsigned char* wpack;
    int feat_dim, k;
    const int32_t* nb_idx;
    const float *pts, *kp_pos, *kp_feat;
    const int32_t* n_points;
    int max_points;  // rows allocated in nb_idx / pts / G / sigma / rgb: the device-side count is clamped to it
    _Float16* G;  // [max_points][256] aggregated hidden features (workspace)
    float *sigma, *rgb;
    // use_view_dir (fields/mlp.py:67-70): the first colour layer sees [feat | enc(ray direction)]; the direction part of its
    // pre-activation is the same for every point of a ray, so the caller hands it over per RAY (dir_bias [n_rays][256] fp32 =
    // enc(d) . W[:, 256:]^T) together with the ray of every compact point; nullptr = published configuration (no view direction)
    const float* dir_bias;
    const int32_t* point_ray;
    // pair kernel: next tile to hand out (zeroed before the launch; nullptr = tiles strided over the grid).  A workgroup takes
    // tile blockIdx.x first and then whatever the counter says: which workgroup computes a tile changes nothing about its result
    int32_t* tile_counter;
    // range guard (ABI 8): the activations travel between layers as fp16; a trained field whose pre-activations leave the fp16 range
    // (|x| >= 65,520 -> inf, then inf - inf -> NaN) would otherwise render NaN pixels without a word.  Every non-finite value reaches
    // the end of its kernel (an inf in one layer makes the whole row non-finite in the next), so the check sits where the values
    // leave: the pair kernel ORs NPCD_SHADE_NONFINITE_PAIRS into *status when an aggregated feature row is not finite, the point
    // kernel NPCD_SHADE_NONFINITE_HEADS when a head's final pre-activation is not.  Never cleared by the kernels; nullptr = no guard.
    int32_t* status;
};
// (the bits of *status; include/npcd_hip.h repeats them for callers)
constexpr int kShadeNonfinitePairs = 1, kShadeNonfiniteHeads = 2;
// inf or NaN, decided on the bits of an OPAQUE integer: the shading sources are compiled with -fno-honor-nans, under which
// `!(x < big)` becomes `x >= big` and even `(bits(x) & 0x7f800000) == 0x7f800000` is recognised as a floating-point class test and
// narrowed to `|x| == inf` (seen in the listing: v_cmp_eq_f32 |v|, 0x7f800000 -- a NaN row passed).  The empty asm hides where
// the integer came from.
__device__ __forceinline__ bool not_finite_bits(float x) {
    uint32_t u = __float_as_uint(x);
    asm volatile("" : "+v"(u));
    return (u & 0x7f800000u) == 0x7f800000u;
}

// ---- positional-encoding column q (0..63) of the [x_rel(3) | per coord: sin f0..9, cos f0..9 | 0] block
__device__ __forceinline__ float enc_value(int q, const float rel[3]) {
    if (q < 3) return rel[q];
    if (q >= 63) return 0.f;
    const int c = (q - 3) / 20, rem = (q - 3) % 20, i = rem % 10;
    // sin(x * 2^i * pi) = sin(2 pi u), u = x * 2^(i-1) (exact scaling); v_sin/v_cos take revolutions
    const float u = rel[c] * (0.5f * (float)(1 << i));
    const float f = __builtin_amdgcn_fractf(u);
    return rem < 10 ? __builtin_amdgcn_sinf(f) : __builtin_amdgcn_cosf(f);
}


// shade_rows.hip: the per-pair layers + aggregation (kernel A) with register-resident activations; `rows_ws` is the part of the
// workspace behind the aggregated features (npcd_shade_workspace_bytes)
int64_t shade_rows_workspace_bytes(int max_points);
int shade_rows_launch(const ShadeArgs& a, void* rows_ws, hipStream_t st);

}  // namespace npcd
