/*
 * npcd_hip.h -- C ABI of libnpcd_hip.so, the MI355X (gfx950) implementation of the NPCD hot path.
 *
 * These entry points are what the reference's two native plug points bind to
 * (paths relative to lmb-freiburg/neural-point-cloud-diffusion @ 2024_10_08):
 *
 *   flash_attn.flash_attn_func       called at npcd/models/diffusion/denoisers/transformer.py:75
 *                                    (import :9-12)                     -> npcd_attn_fwd / npcd_attn_bwd
 *   torch_knnquery.VoxelGrid         ctor  npcd/models/pointnerf/pointnerf.py:20,147-153
 *     .set_pointset                  pointnerf.py:67-75,116-124        -> npcd_grid_build
 *     .query                         fields/aggregators/aggregator.py:63 -> npcd_grid_query
 *
 * plus the fused replacements of the PyTorch op chains around them on the render path
 * (ray generation, neighbour gather + positional encoding + MLP shading, ray marching) and the
 * elementwise chains of the denoiser training step (LayerNorm, bias+GELU, q_sample/MSE, AdamW+EMA).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all buffers
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it and never synchronise
 *   - return value: 0 on success, a negative NPCD_ERR_* otherwise (no exceptions cross the ABI)
 *   - strides are in ELEMENTS, the innermost (head_dim / channel) dimension is contiguous
 *   - dtype codes: NPCD_BF16 = 0, NPCD_F16 = 1, NPCD_F32 = 2
 */
#ifndef NPCD_HIP_H
#define NPCD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPCD_OK 0
#define NPCD_ERR_ARG (-1)          /* bad argument (null pointer, misaligned, negative size) */
#define NPCD_ERR_UNSUPPORTED (-2)  /* shape / dtype not implemented by the kernels           */
#define NPCD_ERR_HIP (-3)          /* a HIP runtime call failed (see npcd_last_hip_error)     */

#define NPCD_BF16 0
#define NPCD_F16 1
#define NPCD_F32 2

int npcd_abi_version(void);
const char* npcd_error_string(int code);
const char* npcd_last_hip_error(void);

/* ------------------------------------------------------------------------------------------
 * Attention over points: out = softmax(q k^T * scale) v, non-causal, no mask, no dropout.
 * Replaces flash_attn_func(q, k, v, causal=False, dropout_p=0) (transformer.py:75) and its
 * autograd backward.  q/k/v are [B, n, H, d] views (typically of one interleaved [B,n,H,3d]
 * buffer, transformer.py:71-72) sharing the stride triple (sb, sn, sh).  Head dims: d = 64 runs the specialised kernels of
 * csrc/attention.hip; d = 32 and d = 128 (ABI 9; the reference's attention takes any width / heads, transformer.py:68-84) run
 * csrc/attention_gen.hip in the 16-bit types, forward and backward (for those the first B H n floats of `delta` hold
 * rowsum(dout * out); no edge scratch is used) and the vector-ALU form of the exact fp32 FORWARD; anything else, and the fp32
 * backward at d != 64, is NPCD_ERR_UNSUPPORTED.
 * lse [B, H, n] fp32 receives log(sum_j exp(scale * <q_i, k_j>)).
 * dtype NPCD_F32 selects the exact-fp32 kernels on the fp32 matrix instruction (the reference's fp32 sampling path,
 * diffusion_model.py:108-133, and `--dtype float32` training through its einsum attention, transformer.py:76-81): nothing is rounded
 * to 16 bits; lse is written when given (may be NULL for inference); strides multiples of 4 elements and 16-byte aligned pointers
 * take the matrix-instruction form, any other layout a vector-ALU form of the forward.  npcd_attn_bwd accepts NPCD_F32 as well
 * (two kernels, dq and dk/dv; `delta` then needs B*H*n floats and is written by the dq kernel).
 * ------------------------------------------------------------------------------------------ */
int npcd_attn_fwd(const void* q, const void* k, const void* v, void* out, float* lse,
                  int B, int n, int H, int d,
                  int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh,
                  int64_t out_sb, int64_t out_sn, int64_t out_sh,
                  float scale, int dtype, void* stream);

/* The same with caller-provided scratch (uninitialised): npcd_attn_fwd_workspace_floats(B, n, H) floats, 0 when the shape needs none.
 * With it, a sequence of 256 j + 1 tokens (the denoiser's 512 points + timestep token) runs without a workgroup for its last query
 * row: the waves of each (batch, head) split that row's keys and a small second kernel merges their partial softmax states.
 * workspace == NULL is npcd_attn_fwd. */
int64_t npcd_attn_fwd_workspace_floats(int B, int n, int H);
int npcd_attn_fwd_ws(const void* q, const void* k, const void* v, void* out, float* lse, float* workspace,
                     int B, int n, int H, int d,
                     int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh,
                     int64_t out_sb, int64_t out_sn, int64_t out_sh,
                     float scale, int dtype, void* stream);

/* The same forward with fp8 (e4m3) operands on the block-scaled matrix instruction (v_mfma_scale_f32_32x32x64_f8f6f4; BASELINE
 * configs[4] names fp8 attention; opt-in, csrc/attention.hip attn_fwd_fp8_kernel): k is quantised to e4m3 and v to e4m3
 * transposed by a packing pass into `workspace`, q and P (with a 2^-8 block scale) inside the kernel (npcd_attn_fwd_fp8_workspace_bytes(B, n, H)
 * bytes; 0 = this length is not covered, use npcd_attn_fwd: n or n - 1 must be a multiple of 64).  bf16 only; out / lse as above;
 * the backward is npcd_attn_bwd on the bf16 operands.  Measured slower than the bf16 kernel at head_dim 64. */
int64_t npcd_attn_fwd_fp8_workspace_bytes(int B, int n, int H);
int npcd_attn_fwd_fp8(const void* q, const void* k, const void* v, void* out, float* lse, void* workspace,
                      int B, int n, int H, int d,
                      int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh,
                      int64_t out_sb, int64_t out_sn, int64_t out_sh,
                      float scale, int dtype, void* stream);

/* Backward of the above.  dq/dk/dv share (g_sb, g_sn, g_sh); out/dout share the out strides.
 * delta: fp32 scratch of npcd_attn_bwd_workspace_floats(B, n, H) elements, written by the call.  First [2, B, H, npad]
 * (npad = n rounded up to a multiple of 64): the per-row constants the dK/dV pass starts its accumulators from, plane 0 =
 * -lse / scale, plane 1 = -rowsum(dout * out), pad rows -inf / 0.  Behind them, for sequences of 128 j + 1 tokens (the
 * denoiser's 512 points + timestep token): 192 floats per (batch, head, 32-row block), the partial sums of the last token's
 * dK / dV / dQ rows (that token gets no workgroup of its own; csrc/attention.hip, 'the edge token').  Bitwise reproducible. */
int64_t npcd_attn_bwd_workspace_floats(int B, int n, int H);
int npcd_attn_bwd(const void* q, const void* k, const void* v, const void* out, const void* dout,
                  const float* lse, void* dq, void* dk, void* dv, float* delta,
                  int B, int n, int H, int d,
                  int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh,
                  int64_t out_sb, int64_t out_sn, int64_t out_sh,
                  int64_t g_sb, int64_t g_sn, int64_t g_sh,
                  float scale, int dtype, void* stream);

/* The backward with the COLUMN SUMS of the packed gradient as a by-product (the bias gradient of c_qkv; replaces the separate
 * npcd_colsum_bf16 pass over dqkv in the fused backbone; reference: nn.Linear bias gradient of c_qkv, transformer.py:67-72).
 * Only for the packed layout (dk = dq + 64 elements, dv = dq + 128, g_sh = 192).  passes: 1 = dq pass, 2 = dk/dv pass, 3 = both.
 * colsum_part: fp32 [npcd_attn_bwd_colsum_rows(B, n, H) + npcd_colsum_scratch_rows(), 3 H 64]; every wave writes the sums of the
 * ROUNDED rows it stores (fixed order); finish with npcd_colsum_finalize(colsum_part, rows, 3 H 64, out, ...).  Bitwise reproducible. */
int npcd_attn_bwd_colsum_rows(int B, int n, int H);
int npcd_attn_bwd_colsum(int passes, const void* q, const void* k, const void* v, const void* out, const void* dout,
                         const float* lse, void* dq, void* dk, void* dv, float* delta, float* colsum_part,
                         int B, int n, int H, int d,
                         int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh,
                         int64_t out_sb, int64_t out_sn, int64_t out_sh,
                         int64_t g_sb, int64_t g_sn, int64_t g_sh,
                         float scale, int dtype, void* stream);

/* One pass of the backward on its own: pass 1 = dq (also writes delta), pass 2 = dk/dv (reads delta,
 * so pass 1 must have run).  Same arguments as npcd_attn_bwd.  Lets a caller time / schedule the two
 * kernels separately. */
int npcd_attn_bwd_pass(int pass, const void* q, const void* k, const void* v, const void* out, const void* dout,
                       const float* lse, void* dq, void* dk, void* dv, float* delta,
                       int B, int n, int H, int d,
                       int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh,
                       int64_t out_sb, int64_t out_sn, int64_t out_sh,
                       int64_t g_sb, int64_t g_sn, int64_t g_sh,
                       float scale, int dtype, void* stream);

/* Single-pass backward (csrc/attention.hip, attn_bwd_fused_kernel): the same gradients as npcd_attn_bwd from FIVE matrix products per
 * (query, key) tile instead of seven -- S and dP are formed once, dV / dK are accumulated from them with the keys on the lanes,
 * dS crosses LDS once and dQ is summed over the keys inside the matrix instruction; one workgroup per (batch, head), 256 keys per
 * pass.  dq_slab: fp32 scratch of npcd_attn_bwd_fused_slab_floats(B, n, H) elements (0 for n <= 256: may be NULL), the running
 * dQ between passes.  delta: the [2, B, H, npad] planes above (written by the kernel's prologue).  Bitwise reproducible. */
int64_t npcd_attn_bwd_fused_slab_floats(int B, int n, int H);
int npcd_attn_bwd_fused(const void* q, const void* k, const void* v, const void* out, const void* dout,
                        const float* lse, void* dq, void* dk, void* dv, float* delta, float* dq_slab,
                        int B, int n, int H, int d,
                        int64_t qkv_sb, int64_t qkv_sn, int64_t qkv_sh,
                        int64_t out_sb, int64_t out_sn, int64_t out_sh,
                        int64_t g_sb, int64_t g_sn, int64_t g_sh,
                        float scale, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------
 * Voxel grid (torch_knnquery.VoxelGrid).  The grid description is passed by value.
 * ------------------------------------------------------------------------------------------ */
typedef struct npcd_grid_params {
    float voxel_size[3];     /* fine voxel edge lengths            (pointnerf.py:148) */
    int32_t voxel_scale[3];  /* coarse cell = fine // voxel_scale  (:149)             */
    int32_t kernel_size[3];  /* odd; dilation + candidate window   (:150)             */
    int32_t max_points_per_voxel;        /* (:151) */
    int32_t max_occ_voxels_per_example;  /* (:152) */
    float range_min[3];      /* (:153) */
    float range_max[3];
    int32_t dims[3];         /* fine grid dimensions, computed by the host */
    int32_t cdims[3];        /* coarse grid dimensions                      */
    /* Which grid the point lists, the point cap and the candidate window live on (the source of torch_knnquery is not
     * available: DESIGN.md section 3 states both readings and the evidence for each):
     *   NPCD_GRID_FINE   (0): <= max_points_per_voxel points per FINE voxel (voxel_size), candidates = the kernel_size window of
     *                         fine voxels around the sample; occupancy on the coarse grid (fine // voxel_scale), dilated by kernel_size.
     *   NPCD_GRID_SCALED (1): everything on ONE grid of edge voxel_size * voxel_scale with ceil(dims / voxel_scale) cells per
     *                         axis: point lists (<= max_points_per_voxel per cell), the max_occ_voxels cap, occupancy dilated by
     *                         kernel_size, and the kernel_size candidate window.
     * The query radius is r * max(voxel_size) -- the UNSCALED edge (aggregator.py:20) -- in both. */
    int32_t grid_level;
} npcd_grid_params;
#define NPCD_GRID_FINE 0
#define NPCD_GRID_SCALED 1

/* bytes of device workspace npcd_grid_build needs for (B, N): per point {ix,iy,iz,kept} + the
 * coarse occupancy bitmaps */
int64_t npcd_grid_workspace_bytes(const npcd_grid_params* g, int B, int N);

/* set_pointset: points [B,N,3] fp32, counts [B] int32 (points >= counts[b] are ignored). */
int npcd_grid_build(const npcd_grid_params* g, const float* points, const int32_t* counts,
                    int B, int N, void* workspace, void* stream);

/* query, dense form: x is given implicitly as ray samples x = o + depth_s * d with
 * depth_s = t0 + (s/(S-1)) * (t1 - t0) (renderer.py:49-77, volume_renderer.py:63-70), or
 * explicitly through `x` [B,R,S,3] when x != NULL (then rays_o/rays_d/t0/t1 may be NULL).
 *   mode 0 = voxel-grid semantics (DESIGN.md "VoxelGrid spec"), radius = r * max(voxel_size)
 *   mode 1 = the reference's voxel_grid=None branch (aggregator.py:42-58), radius = r
 * Outputs (dense, per ray): sample_idx [B,R,M,k] int32 (global index b*N+i sorted by
 * (dist^2, i), -1 pad), sample_loc [B,R,M,3] fp32, slot_sample [B,R,M] int32 (depth-sample
 * index of each slot, -1 = empty), nsel [B,R] int32 (selected slots per ray).            */
int npcd_grid_query(const npcd_grid_params* g, const void* workspace, const float* points,
                    int B, int N, int R, int S, int M, int k, float r, int mode,
                    const float* x, const float* rays_o, const float* rays_d,
                    const float* t0, const float* t1,
                    int32_t* sample_idx, float* sample_loc, int32_t* slot_sample, int32_t* nsel,
                    void* stream);

/* query, fused-render form (voxel-grid semantics only): COMPACT shading-point lists instead of the
 * dense [ray, slot] arrays, no host round trip.  Per ray: ray_base (row of its first valid slot in the
 * compact lists), ray_nsel, ray_bits (bit j = slot j has >= 1 neighbour).  Compact rows (slot order
 * within a ray, ray order unspecified): nb_idx [capacity,k] int32, pts [capacity,3] fp32.
 * counter [4] int32 (ABI 8; two words before): counter[0] = number of compact rows P, counter[1] != 0 if capacity was too small,
 * counter[2] = 0 (the range-guard word the caller may pass to npcd_shade_points as `status`: zeroed by this call so that the
 * guard costs no launch of its own), counter[3] = 0 (reserved). */
int npcd_grid_query_compact(const npcd_grid_params* g, const void* workspace, const float* points,
                            int B, int N, int R, int S, int M, int k, float r,
                            const float* rays_o, const float* rays_d, const float* t0, const float* t1,
                            int32_t* counter, int32_t capacity, int32_t* ray_base, int32_t* ray_nsel,
                            uint64_t* ray_bits, int32_t* nb_idx, float* pts, void* stream);

/* The same query with the compact rows laid out in RAY ORDER: the query kernel leaves each ray's rows and count in `order_ws`
 * (npcd_grid_query_order_ws_bytes(B, R, M, k) bytes of device scratch, 16-byte aligned, uninitialised), a second launch turns the
 * counts into ray_base by a prefix sum and moves the rows.  No atomics: ray_base / nb_idx / pts are bit-identical from run to run
 * (torch_knnquery.VoxelGrid.query returns its rows in ray order as well: aggregator.py:63-73 indexes them by the ray mask).
 * Rows past `capacity` are not written and counter[1] is raised; counter [4] needs no initialisation.  More than 32,768 rays per
 * call: one more small launch adds the counts up per 1,024 rays first (same bits; keeps the prefix sums linear in the rays). */
int64_t npcd_grid_query_order_ws_bytes(int B, int R, int M, int k);
int npcd_grid_query_compact_ordered(const npcd_grid_params* g, const void* workspace, const float* points,
                                    int B, int N, int R, int S, int M, int k, float r,
                                    const float* rays_o, const float* rays_d, const float* t0, const float* t1,
                                    int32_t* counter, int32_t capacity, int32_t* ray_base, int32_t* ray_nsel,
                                    uint64_t* ray_bits, int32_t* nb_idx, float* pts, void* order_ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ray generation + box limits (ray_sampler.py:10-49, math_utils.py:46-97, renderer.py:36-47).
 * extr [V,4,4] world2cam fp32, intr [V,3,3] fp32 -> rays_o/rays_d [V,res*res,3], t0/t1 [V,res*res].
 * limits_ws: npcd_ray_gen_ws_floats(V, res, n_ids) floats of device scratch (n_ids = 0 for all res^2 pixels), uninitialised: the
 * call leaves the global min start / max end / hit flag in its first words, behind them the per-workgroup limits (combined without
 * atomics by the fix-up kernel).
 * ------------------------------------------------------------------------------------------ */
int64_t npcd_ray_gen_ws_floats(int V, int res, int n_ids);
int npcd_ray_gen(const float* extr, const float* intr, int V, int res, float box,
                 float* rays_o, float* rays_d, float* t0, float* t1, float* limits_ws, void* stream);
/* Same for a subset of the pixels: pixel_ids [n_ids] int32 row-major pixel numbers (i * res + j), the same for every view
 * (the training path renders ~100 random pixels per view: renderer.py:232-238) -> rays_o/rays_d [V,n_ids,3], t0/t1 [V,n_ids]. */
int npcd_ray_gen_subset(const float* extr, const float* intr, int V, int res, float box, const int32_t* pixel_ids, int n_ids,
                        float* rays_o, float* rays_d, float* t0, float* t1, float* limits_ws, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused shading of compact shading points (aggregators/mlp.py:36-125, fields/mlp.py:38-72,
 * field.py:113-141): gather -> rel. position -> positional encoding -> 5-layer MLP ->
 * inverse-distance aggregation -> density head (softplus(x-1)) and colour head (sigmoid).
 * Weights are packed once by npcd_shade_pack_weights into `wpack`.
 *   nb_idx [P,k] int32 (-1 pad), pts [P,3] fp32, kp_pos [B*N,3] fp32, kp_feat [B*N,F] fp32
 *   -> sigma [P] fp32, rgb [P,3] fp32.  n_points_dev: device int32 holding P (so that P may be
 *   produced on the device without a host round trip); max_points = rows allocated in every
 *   per-point array: it bounds the launch and the kernels clamp the device-side count to it.
 * Range guard (ABI 8).  The kernels carry activations between layers as fp16 (fp32 accumulation); the reference runs these MLPs in
 * fp32 (eval_pointnerf.py has no autocast).  `status` (device int32, may be NULL) is OR-ed with NPCD_SHADE_NONFINITE_PAIRS when an
 * aggregated feature row, and with NPCD_SHADE_NONFINITE_HEADS when a head's final pre-activation, is inf / NaN -- which is where
 * every fp16 overflow (|x| >= 65,520) of an earlier layer ends up.  The library never clears the word: zero it, render, read it
 * with the point count.  A set bit means the pixels of this call are not the reference's; use the fp32-class path instead.
 * ------------------------------------------------------------------------------------------ */
#define NPCD_SHADE_NONFINITE_PAIRS 1
#define NPCD_SHADE_NONFINITE_HEADS 2
int64_t npcd_shade_wpack_bytes(int feat_dim, int n_freqs, int hidden);
int64_t npcd_shade_workspace_bytes(int max_points, int hidden);
int npcd_shade_pack_weights(const float* const* weights_host, const float* const* biases_host,
                            int feat_dim, int n_freqs, int hidden, void* wpack_host);
int npcd_shade_points(const void* wpack, int feat_dim, int n_freqs, int hidden,
                      const int32_t* nb_idx, const float* pts, const float* kp_pos, const float* kp_feat,
                      const int32_t* n_points_dev, int max_points, int k,
                      float* sigma, float* rgb, void* workspace, int32_t* status, void* stream);

/* The same with the reference's use_view_dir option (models/npcd.py:8 -> fields/mlp.py:30-36,67-70): the first colour layer sees
 * [feat | enc(ray direction)].  Its direction part is the same for every shading point of a ray, so the caller passes it per RAY:
 * dir_bias [n_rays, hidden] fp32 = enc(d) . W[:, hidden:]^T (16-byte aligned), point_ray [max_points] int32 = the ray of every compact
 * row; wpack holds that layer packed from its first `hidden` columns.  Added to the fp32 accumulators of the layer. */
int npcd_shade_points_dir(const void* wpack, int feat_dim, int n_freqs, int hidden,
                          const int32_t* nb_idx, const float* pts, const float* kp_pos, const float* kp_feat,
                          const int32_t* n_points_dev, int max_points, int k,
                          float* sigma, float* rgb, void* workspace, const float* dir_bias, const int32_t* point_ray,
                          int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------
 * The POINT-level layers in the reference's fp32 numerics class (round 5, ABI 8): last aggregator layer (linear; aggregators/mlp.py:83-84,
 * local_field[8]), shape_net + softplus(x - 1) (fields/mlp.py:38-51, field.py:30,126-128), channel_net + sigmoid (fields/mlp.py:53-72,
 * field.py:139-140) on `feat` [max_points, 256] fp32 = the aggregated per-point features (the output of npcd_pair_mlp_fwd with
 * NPCD_PAIR_MLP_X2 + the weighted mean).  Every operand is two bf16 halves, every product three matrix instructions in fp32
 * accumulators (the numerics of NPCD_PAIR_MLP_X2: ~4e-6 relative per layer, fp32's exponent range); csrc/points_x2.hip.
 *   npcd_points_x2_pack: the twelve HOST pointers of npcd_shade_pack_weights (entries 4..11 are read); c0_in_dim = input columns of
 *   channel_net.0 (256, or 256 + the direction encoding: its first 256 columns are packed).
 *   npcd_points_x2: n_points_dev may be NULL (= max_points); dir_bias [rows, 256] fp32 + point_ray [max_points] int32 add row
 *   point_ray[p] to the first colour layer's pre-activation of point p (use_view_dir, fields/mlp.py:67-70), or both NULL.
 *   -> sigma [max_points], rgb [max_points, 3] (activations applied).
 * ------------------------------------------------------------------------------------------ */
/* The four non-linear PER-PAIR layers + the inverse-distance mean in the same numerics, forward only (aggregators/mlp.py:62-125,
 * aggregator.py:122-156, positional_encoder.py:16-20): nb_idx [max_points, k] int32 global neighbour indices (-1 = none, anywhere in a
 * row: the compact query's lists as they are), pts [max_points, 3], kp_pos [B N, 3], kp_feat [B N, feat_dim] -> G [max_points, 256] fp32,
 * the input of npcd_points_x2.  feat_dim in {32, 128}, k <= 8; n_points_dev may be NULL.  npcd_pairs_x2_pack reads entries 0..3 of the
 * twelve host pointers.  (npcd_pair_mlp_fwd with NPCD_PAIR_MLP_X2 computes the same and can save its activations for training.) */
int64_t npcd_pairs_x2_wpack_bytes(int feat_dim);
int npcd_pairs_x2_pack(const float* const* weights_host, const float* const* biases_host, int feat_dim, void* wpack_host);
int npcd_pairs_x2(const void* wpack, int feat_dim, const int32_t* nb_idx, const float* pts, const float* kp_pos, const float* kp_feat,
                  const int32_t* n_points_dev, int max_points, int k, float* G, void* stream);
int64_t npcd_points_x2_wpack_bytes(void);
int npcd_points_x2_pack(const float* const* weights_host, const float* const* biases_host, int c0_in_dim, void* wpack_host);
int npcd_points_x2(const void* wpack, const float* feat, const int32_t* n_points_dev, int max_points, float* sigma, float* rgb,
                   const float* dir_bias, const int32_t* point_ray, void* stream);
/* The same layers as the stage-1 TRAINING forward (published configuration, no view directions): save [6][max_points][256] fp32 = feat, s0, c0,
 * c1, c2, c3 (every hidden activation, after its LeakyReLU where the layer has one: what the backward of the eight Linear layers
 * needs), pre [max_points][4] = the heads' pre-activations (r, g, b, sigma) with their biases -- softplus(x - 1) / sigmoid are the
 * caller's (autograd's). */
int npcd_points_x2_train(const void* wpack, const float* feat, int max_points, float* save, float* pre, void* stream);
/* npcd_points_x2_pack from DEVICE tensors (training packs once per optimizer step): weights_dev / biases_dev = a host array of the eight
 * device pointers of local_field.8, shape_net.{0,2}, channel_net.{0,2,4,6,8}. */
int npcd_points_x2_pack_dev(const float* const* weights_dev, const float* const* biases_dev, int c0_in_dim, void* wpack_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Ray marching (renderer.py:96-110,120-185, volume_renderer.py:23-39) on the dense slot layout:
 * sigma/rgb are COMPACT per valid slot (row-major over [ray, slot]); slot_valid [Nr,M] uint8,
 * slot_loc [Nr,M,3], point_base [Nr] int32 = index of the ray's first compact point.
 * -> mask [Nr], depth [Nr], channels [Nr,3].  depth_ws: npcd_ray_march_ws_floats(Nr) floats of scratch, uninitialised: the
 * call leaves the global depth limits in its first two words (npcd_ray_march_bwd reads them), the rest holds the per-workgroup
 * limits of the march (one wave per ray for M <= 64; combined without atomics by the clamp kernel).
 * ------------------------------------------------------------------------------------------ */
int64_t npcd_ray_march_ws_floats(int Nr);
int npcd_ray_march(const float* sigma, const float* rgb, const uint8_t* slot_valid, const float* slot_loc,
                   const int32_t* point_base, const float* rays_o, const float* rays_d, const float* t1,
                   int Nr, int M, int white_back, float* mask, float* depth, float* channels,
                   float* depth_ws, void* stream);

/* Backward of npcd_ray_march w.r.t. sigma and rgb (stage-1 training: positions and rays are constants).  depth_ws is the
 * scratch the forward call left (global depth limits); g_mask / g_depth [Nr], g_chan [Nr,3] are the upstream gradients;
 * dsigma [P], drgb [P,3] are written for every compact point.  M <= 64. */
int npcd_ray_march_bwd(const float* sigma, const float* rgb, const uint8_t* slot_valid, const float* slot_loc,
                       const int32_t* point_base, const float* rays_o, const float* rays_d, const float* t1,
                       int Nr, int M, int white_back, const float* depth_ws, const float* g_mask, const float* g_depth,
                       const float* g_chan, float* dsigma, float* drgb, void* stream);

/* ------------------------------------------------------------------------------------------
 * Elementwise / normalisation / optimizer kernels of the denoiser training step (HBM-bound).
 * They fuse the eager op chains of transformer.py:169-172,136-137 (under autocast) and of
 * diffusion_training.py:169-174 + utils/ema.py:114-138.  bf16 tensors are passed as void*.
 * ------------------------------------------------------------------------------------------ */
/* x_out = x_in (+ delta); y = LayerNorm(x_out)*gamma+beta as bf16; mean/rstd [T] saved.
 * delta (bf16 [T,W]) and x_out may be NULL.  W % 4 == 0, W <= 2048. */
int npcd_add_ln_fwd(const float* x_in, const void* delta, const float* gamma, const float* beta,
                    float* x_out, void* y, float* mean, float* rstd, int T, int W, float eps, void* stream);
/* LayerNorm backward: dx = LNbwd(dy) (+ dres), optionally also as bf16 (dxb); column partials
 * [npcd_ln_bwd_blocks(T)][W] of dgamma, dbeta and of dx (any may be NULL). */
int npcd_ln_bwd_blocks(int T);
int npcd_ln_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                const float* dres, float* dx, void* dxb, float* part_gamma, float* part_beta,
                float* part_col, int T, int W, void* stream);
/* out[c] (+)= sum_b part[b][c], fixed summation order.  `part` [nblk][N] must be followed by
 * npcd_colsum_scratch_rows() more rows of N floats (used as the stage buffer of the 2-stage sum). */
int npcd_colsum_scratch_rows(void);
int npcd_colsum_finalize(const float* part, int nblk, int N, float* out, int accumulate, void* stream);
/* Up to NPCD_COLSUM_MAX_JOBS independent column sums in one pair of launches (bit-identical to the single
 * call per job).  One residual block's backward emits eight: LN dgamma/dbeta x2, four Linear bias gradients
 * (the reference gets them from autograd's SumBackward nodes, transformer.py:162-172). */
#define NPCD_COLSUM_MAX_JOBS 8
typedef struct NpcdColsumJob {
    const float* part; /* [nblk + npcd_colsum_scratch_rows()][N] */
    float* out;        /* [N] */
    int nblk, N, accumulate, reserved;
} NpcdColsumJob;
int npcd_colsum_finalize_batch(const NpcdColsumJob* jobs, int njobs, void* stream);
/* GELU (exact erf form) on bf16; backward also emits column partials [npcd_colsum_blocks(T)][N] of dh */
int npcd_gelu_fwd(const void* h, void* g, int64_t numel, void* stream);
int npcd_colsum_blocks(int T);
int npcd_gelu_bwd(const void* dg, const void* h, void* dh, float* part, int T, int N, void* stream);
int npcd_colsum_bf16(const void* a, float* part, int T, int N, void* stream);
/* Weight gradient of a Linear layer with J <= 4 outputs (the PointNeRF field's heads: fields/mlp.py:38-72, last layers 256 -> 1 and
 * 256 -> 3) over T rows: part [npcd_small_wgrad_blocks(T) + npcd_colsum_scratch_rows(), J * K] fp32 partial sums of
 * dW[j][k] = sum_p dy[p][j] x[p][k] (dy [T, J], x [T, K] bf16 row-major, K a power of two in 64..2048); finish with
 * npcd_colsum_finalize(part, blocks, J * K, dW, ...).  Replaces the library GEMM (350 us for one workgroup's worth of work). */
int npcd_small_wgrad_blocks(int T);
int npcd_small_wgrad(const void* dy, const void* x, float* part, int T, int J, int K, void* stream);
/* AdamW (torch semantics) + EMA lerp + bf16 shadow copy + optional gradient zeroing, one pass.
 * ema and shadow_bf16 may be NULL; step is the 1-based step count (bias correction). */
int npcd_adamw_ema(float* p, float* g, float* m, float* v, float* ema, void* shadow_bf16, int64_t numel,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   float ema_decay, int zero_grad, void* stream);
int npcd_cast_f32_bf16(const float* src, void* dst, int64_t numel, void* stream);
/* The same kernels for either 16-bit activation type of a training run: dtype = NPCD_BF16 (bf16 autocast; the names above) or
 * NPCD_F16 (float16 autocast with loss scaling -- the reference's default --dtype, train_diffusion.py:78,
 * train/diffusion_training.py:60-62).  Same arguments otherwise; shadow_dtype = the type of the 16-bit parameter shadow. */
int npcd_add_ln_fwd_dt(const float* x_in, const void* delta, const float* gamma, const float* beta,
                       float* x_out, void* y, float* mean, float* rstd, int T, int W, float eps, int dtype, void* stream);
int npcd_ln_bwd_dt(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                   const float* dres, float* dx, void* dxb, float* part_gamma, float* part_beta,
                   float* part_col, int T, int W, int dtype, void* stream);
int npcd_gelu_fwd_dt(const void* h, void* g, int64_t numel, int dtype, void* stream);
int npcd_gelu_bwd_dt(const void* dg, const void* h, void* dh, float* part, int T, int N, int dtype, void* stream);
int npcd_colsum_dt(const void* a, float* part, int T, int N, int dtype, void* stream);
int npcd_adamw_ema_dt(float* p, float* g, float* m, float* v, float* ema, void* shadow, int shadow_dtype, int64_t numel,
                      float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                      float ema_decay, int zero_grad, void* stream);
int npcd_cast_f32_dt(const float* src, void* dst, int64_t numel, int dtype, void* stream);

/* Weight gradient of a Linear layer (the nn.Linear weight gradients of transformer.py:67-72, :107-115, :118-137, which autograd
 * computes in the reference): dW [N, K] fp32 = dy[T, N]^T x[T, K], dy / x row-major in the 16-bit `dtype`, fp32 accumulation and
 * output; N % 256 == 0, K % 256 == 0, any T.  The token range is split over npcd_wgrad_slices(T, N, K) workgroup slices that write
 * fp32 slabs (`workspace`: that many x N x K floats, unused when it is 1) which a second kernel adds in slice order: bitwise
 * reproducible.  NPCD_ERR_UNSUPPORTED for other shapes (the caller then uses the library). */
int npcd_wgrad_slices(int T, int N, int K);
int npcd_wgrad(const void* dy, const void* x, float* out, float* workspace, int T, int N, int K, int dtype, void* stream);
/* (ABI 9) `count` (1..8) such weight gradients over the SAME token range in one launch: dW_g [N_g, K_g] = dy_g[T, N_g]^T x_g[T, K_g].
 * One workgroup per 256 x 256 output tile over all T tokens: no slices, no workspace, one fixed summation order.  Meant for the token
 * count of one rank of the strong-scaling job (T = 4,104: the four Linear layers of a block are 192 tiles = one round on 192 of 256
 * CUs, beside the backward's critical path on another stream).  dy / x / out / N / K: host arrays of `count` entries. */
int npcd_wgrad_group(int count, const void* const* dy, const void* const* x, float* const* out, const int* N, const int* K, int T,
                     int dtype, void* stream);

/* Linear layers of the residual block as own NT products with fused epilogues (csrc/gemm_nt.hip): 16-bit `dtype` operands
 * (NPCD_BF16 / NPCD_F16), fp32 accumulation, 16-bit outputs; x [M, K], w [N, K] row-major (the nn.Linear weight as stored), any
 * M >= 1, N % 1024 == 0, K % 64 == 0, M * N and M * K below 2^31, pointers 16-byte aligned; NPCD_ERR_UNSUPPORTED otherwise (the
 * caller then uses the library).  Replace, in the reference, F.linear of transformer.py:67 / :107-115 / :118-137 and the autograd
 * nodes behind them:
 *   npcd_linear_fwd:        y = x w^T + bias                                   (bias [N] 16 bit, may be NULL)
 *   npcd_linear_gelu_fwd:   h = x w^T + bias (rounded to 16 bit), g = gelu_erf(h)   -- mlp.c_fc followed by nn.GELU() (:131)
 *   npcd_linear_dgelu_bwd:  dh = round16(dy wt^T) * gelu_erf'(h)               -- the data gradient of mlp.c_proj (wt = its weight
 *                           TRANSPOSED, [N = 4 W, K = W] row-major, npcd_transpose_16) followed by the GELU backward; also writes
 *                           part [npcd_linear_dgelu_rows(M)][N] fp32 column partial sums of dh (the c_fc bias gradient), to be
 *                           finished by npcd_colsum_finalize(part, npcd_linear_dgelu_rows(M), N, ...)
 * A data gradient dx = dy W is npcd_linear_fwd(dy, W^T, NULL, ...). */
int npcd_linear_fwd(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int dtype, void* stream);
int npcd_linear_gelu_fwd(const void* x, const void* w, const void* bias, void* h, void* g, int M, int N, int K, int dtype, void* stream);
int npcd_linear_dgelu_rows(int M);
int npcd_linear_dgelu_bwd(const void* dy, const void* wt, const void* h, void* dh, float* part, int M, int N, int K, int dtype, void* stream);
/* The same product y = x w^T + bias on 128 x 128 tiles (csrc/gemm_nt.hip, lin128_kernel): the form for the token counts of ONE RANK
 * of the strong-scaling job (T = 4,104 / 8,208 per rank at 8 / 4 GPUs), where the N = 1,024 products of a block are 64-68 tiles of
 * 256 x 256 on 256 CUs.  Any M >= 1 (left-over rows M mod 128 <= 32 ride on the last row tile), N % 128 == 0, K % 64 == 0, M * K and
 * N * K below 2^30 elements (32-bit byte offsets), M * N below 2^31; 16-byte aligned pointers. */
int npcd_linear128_fwd(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int dtype, void* stream);
/* out [C, R] = in [R, C]^T for 16-bit elements (the transposed shadow of the Linear weights that the data gradients read) */
int npcd_transpose_16(const void* in, void* out, int R, int C, void* stream);

/* out[i] = part[0 * numel + i] + ... + part[(S - 1) * numel + i], fp32, added in slice order (S = 2, 4 or 8; numel % 4 == 0;
 * 16-byte aligned): the sum of the row-split weight-gradient partials of the fused backbone (replaces torch.sum(part, dim=0)
 * there; the reference's nn.Linear weight gradient, summed over token slices). */
int npcd_sum_slices(const float* part, float* out, int S, int64_t numel, void* stream);

/* One DDPM reverse step of the sampler, fused (reference gaussian_diffusion.py:100-146: _predict_xstart_from_eps :127-129,
 * clamp :111-113, posterior mean :88-98, noise add :138-144):
 *   x0 = recip[t] x_t - recipm1[t] eps (clamped to [clip_lo, clip_hi] if has_clip); x_prev = coef1[t] x0 + coef2[t] x_t
 *        + [t > 0] exp(logvar[t] / 2) noise.
 * x_t, noise, x_prev (and x0_out, may be NULL) are fp32 [B, per_sample]; eps is fp32 or bf16 (eps_dtype); t is int64 [B];
 * the five tables are the process's fp32 [num_timesteps] device arrays. */
int npcd_ddpm_reverse_step(const float* x_t, const void* eps, int eps_dtype, const float* noise, float* x_prev, float* x0_out,
                           const int64_t* t, int B, int64_t per_sample, const float* tab_recip, const float* tab_recipm1,
                           const float* tab_coef1, const float* tab_coef2, const float* tab_logvar, float clip_lo, float clip_hi,
                           int has_clip, void* stream);

/* Forward process and training loss of the DDPM, fused (reference gaussian_diffusion.py:68-76 q_sample, :199-230 p_losses):
 *   npcd_q_sample    : x_t = tab_sqrt_acp[t_b] * x_0 + tab_sqrt_1macp[t_b] * noise, fp32 [B, per_sample], t int64 [B]; the products
 *                      and the sum are rounded separately like the reference's eager ops (bit-identical to them)
 *   npcd_eps_mse_fwd : loss[0] = mean((noise - eps)^2 / 2) over numel elements (eps fp32 or bf16), optional pointwise output
 *                      (fp32 [numel] or NULL); part: fp32 scratch of npcd_eps_mse_blocks() elements; fixed-order sums
 *   npcd_eps_mse_bwd : grad = -(noise - eps) / numel * upstream_dev[0]   (grad has the dtype of eps) */
int npcd_q_sample(const float* x0, const float* noise, const int64_t* t, const float* tab_sqrt_acp, const float* tab_sqrt_1macp,
                  float* x_t, int B, int64_t per_sample, void* stream);
int npcd_eps_mse_blocks(void);
int npcd_eps_mse_fwd(const float* noise, const void* eps, int eps_dtype, int64_t numel, float* pointwise, float* part, float* loss,
                     void* stream);
int npcd_eps_mse_bwd(const float* noise, const void* eps, int eps_dtype, int64_t numel, const float* upstream_dev, void* grad,
                     void* stream);

/* ray march on the compact layout of npcd_grid_query_compact (same math as npcd_ray_march).  `capacity` = rows allocated in
 * sigma / rgb / pts: when the compact lists overflowed (counter[1] != 0) a ray whose rows lie past it is marched as empty --
 * the caller discards that result and retries with larger lists, nothing is read out of bounds meanwhile. */
int npcd_ray_march_compact(const float* sigma, const float* rgb, const uint64_t* ray_bits, const float* pts,
                           const int32_t* ray_base, const float* rays_o, const float* rays_d, const float* t1,
                           int Nr, int M, int capacity, int white_back, float* mask, float* depth, float* channels,
                           float* depth_ws, void* stream);

/* Fused render (ABI 9): two launches per view fewer, the same bits (Renderer.forward, renderer.py:202-268, for every pixel of
 * `views` views per example; ray_sampler.py:10-49 + renderer.py:36-47 + aggregator.py:63-73 + renderer.py:120-185).
 *   npcd_render_rays_query     = npcd_ray_gen + npcd_grid_query_compact_ordered in one query launch: the query kernel computes its ray
 *       from extr [B * views, 4, 4] (world2cam) / intr [B * views, 3, 3] and writes rays_o / rays_d [B, R, 3], t0 / t1 [B, R]
 *       (R = views * res^2) for the later stages.  t0 / t1 of a ray that misses the cube keep the raw -1 / -2; lim_part
 *       [npcd_render_lim_words(B, R)] receives per-group (min start, max end) keys of the rays that hit.  Needs the ordered form
 *       (order_ws) and box >= the grid's range on every axis (a missing ray then has no sample inside the grid: nothing is lost by not
 *       sampling it between the global limits); NPCD_ERR_UNSUPPORTED otherwise -- the caller then uses the separate entry points.
 *   npcd_ray_march_compact_fused = npcd_ray_march_compact that ends a missing ray at the global end taken from lim_part (the
 *       renderer.py:40-43 fix-up): t0 [Nr] as written by the query, lim_part / lim_pairs = npcd_render_lim_words(B, R) / 2. */
int64_t npcd_render_lim_words(int B, int R);
int npcd_render_rays_query(const npcd_grid_params* grid, const void* workspace, const float* points, int B, int N, const float* extr,
                           const float* intr, int views, int res, float box, int S, int M, int k, float r, float* rays_o, float* rays_d,
                           float* t0, float* t1, int32_t* counter, int32_t capacity, int32_t* ray_base, int32_t* ray_nsel,
                           uint64_t* ray_bits, int32_t* nb_idx, float* pts, void* order_ws, uint32_t* lim_part, void* stream);
int npcd_ray_march_compact_fused(const float* sigma, const float* rgb, const uint64_t* ray_bits, const float* pts, const int32_t* ray_base,
                                 const float* rays_o, const float* rays_d, const float* t0, const float* t1, const uint32_t* lim_part,
                                 int lim_pairs, int Nr, int M, int capacity, int white_back, float* mask, float* depth,
                                 float* channels, float* depth_ws, void* stream);

/* ---- stage-1 training path: the data movement around the per-pair MLP (aggregators/mlp.py:36-125,
 * positional_encoder.py:16-20, aggregator.py:122-144).  Pairs (shading point, neighbour) are compact and ordered by point:
 * flat [Q] = global neighbour index, owner [Q] = shading-point index, off / cnt [P] = first pair and number of pairs of a point.
 *   npcd_pair_input_fwd : x0 [Q, F + 3 + 6 nf] = [feat[flat] | rel | sin / cos bands of rel], w [Q] = 1 / (|rel| + 1e-5),
 *                         rel = pts[owner] - kp_pos[flat]
 *   npcd_pair_input_bwd : dfeat [B*N, F] += dx0[:, :F] scattered by flat (float atomics; dfeat must be zeroed by the caller)
 *   npcd_pair_aggregate : backward == 0: dst = agg [P, C] = sum over a point's pairs of w / (sum w) * src[q]  (src = local [Q, C]);
 *                         backward != 0: dst = dlocal [Q, C] = w / (sum w) * src[p]                      (src = dagg [P, C]) */
int npcd_pair_input_fwd(const int64_t* flat, const int64_t* owner, const float* pts, const float* kp_pos, const float* kp_feat,
                        int feat_dim, int n_freqs, int64_t n_pairs, float* x0, float* w, void* stream);
int npcd_pair_input_bwd(const int64_t* flat, const float* dx0, int feat_dim, int n_cols, int64_t n_pairs, float* dfeat, void* stream);
/* LeakyReLU backward fused with the bias-gradient column partials (stage-1 MLPs, utils/model.py:22-36): dy = dz * (z > 0 ? 1 :
 * slope) with z the activation output, part [npcd_leaky_bwd_blocks(rows) + npcd_colsum_scratch_rows()][N] = column partials of
 * dy for npcd_colsum_finalize.  dtype NPCD_F32 or NPCD_BF16; N * element size a multiple of 16 bytes that divides 4096. */
int npcd_leaky_bwd_blocks(int64_t rows);
int npcd_leaky_bwd_colsum(const void* dz, const void* z, void* dy, float* part, int64_t rows, int N, float slope, int dtype,
                          void* stream);
int npcd_pair_aggregate(int backward, const float* src, const float* w, const int64_t* off, const int64_t* cnt, int channels,
                        int64_t n_points, float* dst, void* stream);

/* ---- stage-1 training path: the per-pair aggregator MLP itself (aggregators/mlp.py:36-100; utils/model.py:22-36; pointnerf.py:
 * 174-179: Linear(F+63,256) + 3 x Linear(256,256), LeakyReLU(0.01) after each) on the matrix cores, forward AND backward, bf16
 * operands with fp32 accumulation (csrc/pairs_mlp.hip).  Pairs as above: row q of every [Q, .] array, ordered by point.
 *   npcd_pair_mlp_pack : fp32 DEVICE weights W_l [256, in_l] (in_0 = F + 63) / biases b_l [256], l = 0..3  ->  wpack (device,
 *                        npcd_pair_mlp_wpack_bytes()): bf16 fragment order of W_l (forward) and W_l^T (data gradient) + fp32 biases
 *   npcd_pair_mlp_fwd  : nb_idx [P,k] int64 (-1 pad, valid first), pts [P,3], kp_pos [B*N,3], kp_feat [B*N,F] fp32, off [P] int64
 *                        -> G [P,256] fp32 = inverse-distance weighted mean over a point's pairs of the 4th layer's output
 *                        (aggregators/mlp.py:102-125; the network's 5th, linear layer is applied by the caller on points);
 *                        saved for the backward: x0 [Q, F+64] bf16, acts [4][Q][256] bf16, wn [Q] fp32 (normalised weights)
 *   npcd_pair_mlp_bwd  : dG [P,256] fp32, owner [Q] int64 -> dfeat [Q,F] fp32 (gradient w.r.t. the gathered feature rows, to be
 *                        scattered with npcd_pair_input_bwd), dW[l] [256, in_l] fp32, db[l] [256] fp32 (overwritten; slabs summed
 *                        in a fixed order: bitwise reproducible).  dact: 2 x [Q,256] bf16 scratch; part: fp32 scratch of
 *                        npcd_pair_mlp_bwd_workspace_floats() elements.  F in {32, 128}.
 * `precision` (ABI 8) of every entry point:
 *   NPCD_PAIR_MLP_BF16 (0): bf16 operands, fp32 accumulation -- narrower than the reference's fp32 (train_pointnerf.py has no autocast);
 *   NPCD_PAIR_MLP_X2   (1): fp32-class -- every operand (weights, activations, gradients) as two bf16 halves hi + lo, every product
 *                        as three matrix instructions hi*hi + hi*lo + lo*hi accumulated in fp32: ~1e-5 relative per product, fp32's
 *                        exponent range.  All 16-bit arrays then hold TWO planes (hi, then lo): x0 [2][Q][F+64], acts [4][2][Q][256],
 *                        dact 2 x [2][Q][256]; wpack twice the matrices.  Same calls, same outputs (G, dfeat, dW, db: fp32).
 * Forward only (rendering with fp32-class shading): x0 = acts = wn = NULL -- nothing is saved. */
#define NPCD_PAIR_MLP_BF16 0
#define NPCD_PAIR_MLP_X2 1
int64_t npcd_pair_mlp_wpack_bytes(int feat_dim, int precision);
int npcd_pair_mlp_pack(const float* const* weights_dev, const float* const* biases_dev, int feat_dim, int precision, void* wpack_dev,
                       void* stream);
int npcd_pair_mlp_fwd(const void* wpack, int feat_dim, int precision, const int64_t* nb_idx, const float* pts, const float* kp_pos,
                      const float* kp_feat, const int64_t* off, int64_t n_points, int k, int64_t n_pairs, void* x0, void* acts,
                      float* wn, float* G, void* stream);
int npcd_pair_mlp_bwd_slabs(int64_t n_pairs, int precision);
int64_t npcd_pair_mlp_bwd_workspace_floats(int feat_dim, int64_t n_pairs, int precision);
int npcd_pair_mlp_bwd(const void* wpack, int feat_dim, int precision, const float* dG, const int64_t* owner, const float* wn,
                      const void* x0, const void* acts, int64_t n_pairs, void* dact, float* dfeat, float* part, float* const* dW,
                      float* const* db, void* stream);

/* ---- fp32-class forward of the denoiser's Linear layers (sampling in the reference's numerics class, diffusion_model.py:108-133;
 * transformer.py:67,107-115,118-137): x W^T on fp32 operands as ONE bf16 library GEMM over the three cross products of split operands,
 * [xh | xl | xh] [Wh | Wh | Wl]^T (x = xh + xl, bf16 halves; fp32 accumulation and output): 3e-6 relative, 2.3-3.5 x the fp32 GEMM's rate.
 * npcd_split3_bf16: the activation side in one pass: y = x [rows, K] fp32 (+ bias [K]) (-> exact-erf GELU if gelu != 0)  ->
 * out [rows, 3 K] bf16 = [hi(y) | lo(y) | hi(y)].  K % 8 == 0, 16-byte aligned pointers.  (ABI 8) */
int npcd_split3_bf16(const float* x, const float* bias, void* out, int64_t rows, int K, int gelu, void* stream);
/* The same hand-over behind a LayerNorm (transformer.py:169-172: x = x + attn(ln_1(x)); x = x + mlp(ln_2(x))), one pass:
 * xnew = x (+ o + bias, written when o is given);  out [rows, 3 W] = [hi | lo | hi] of LayerNorm(xnew) * gamma + beta (fp32 statistics, biased
 * variance).  o / bias / xnew: all three or none.  W in {256, 512, 768, 1024, 2048, 4096}.  (ABI 8) */
int npcd_add_ln_split3_bf16(const float* x, const float* o, const float* bias, const float* gamma, const float* beta, float* xnew,
                            void* out, int64_t rows, int W, float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NPCD_HIP_H */
