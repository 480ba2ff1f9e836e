// Stage-1 training path, the data movement around the per-pair MLP: building the MLP input of every
// (shading point, neighbour) pair and the inverse-distance aggregation of its output, forward and backward.  The Linear
// layers between them are library GEMMs under autograd (npcd/models/pointnerf/train_path.py); these four kernels replace
// the ~25 gather / sin / cos / cat / index_add / index_put launches torch needs for the same thing.
//   pair_input  (aggregators/mlp.py:36-100, positional_encoder.py:16-20, aggregator.py:122-144)
//       x0[q] = [ feat[nb] (F) | rel (3) | per coordinate: sin(rel_c 2^i pi), i < nf, then cos(...) ],  rel = pt - pos[nb]
//       w[q]  = 1 / (|rel| + 1e-5);   backward: dfeat[nb] += dx0[q, :F]   (positions are constants in stage 1)
//   aggregate   (aggregators/mlp.py:102-125)
//       agg[p] = sum_{q in p} w[q] / (sum_{q' in p} w[q']) * local[q];   backward: dlocal[q] = w_n[q] * dagg[p]
// Pairs are compact and ordered by point, so a point's pairs are the rows off[p] .. off[p] + cnt[p]: no atomics in the
// aggregation (the reference's index_add_ is order-dependent on a GPU); the feature gradient is a float atomicAdd like the
// reference's.  All fp32 (the reference's numerics for this stage).  Bound: HBM (x0 is 380 B per pair).
#include "common.h"

namespace npcd {

__global__ __launch_bounds__(256) void pair_input_fwd_kernel(const int64_t* __restrict__ flat, const int64_t* __restrict__ owner,
                                                             const float* __restrict__ pts, const float* __restrict__ kp_pos,
                                                             const float* __restrict__ kp_feat, int F, int nf, int64_t Q,
                                                             float* __restrict__ x0, float* __restrict__ w) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const int64_t nb = flat[q], p = owner[q];
    float rel[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) rel[c] = pts[p * 3 + c] - kp_pos[nb * 3 + c];
    const int ncol = F + 3 + 6 * nf;
    float* row = x0 + q * ncol;
    for (int col = lane; col < ncol; col += 64) {
        float v;
        if (col < F) {
            v = kp_feat[nb * F + col];
        } else if (col < F + 3) {
            v = rel[col - F];
        } else {
            const int e = col - F - 3, c = e / (2 * nf), wi = e - c * 2 * nf, f = wi < nf ? wi : wi - nf;
            const float spec = (c == 0 ? rel[0] : c == 1 ? rel[1] : rel[2]) * ldexpf(3.14159265358979323846f, f);
            v = wi < nf ? sinf(spec) : cosf(spec);
        }
        row[col] = v;
    }
    if (lane == 0) w[q] = 1.f / (sqrtf(rel[0] * rel[0] + rel[1] * rel[1] + rel[2] * rel[2]) + 1e-5f);
}

__global__ __launch_bounds__(256) void pair_input_bwd_kernel(const int64_t* __restrict__ flat, const float* __restrict__ dx0, int F, int ncol,
                                                             int64_t Q, float* __restrict__ dfeat) {
    const int lane = threadIdx.x & 63;
    const int64_t q = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const int64_t nb = flat[q];
    for (int col = lane; col < F; col += 64) atomicAdd(dfeat + nb * F + col, dx0[q * ncol + col]);
}

template <bool BWD>
__global__ __launch_bounds__(256) void aggregate_kernel(const float* __restrict__ src, const float* __restrict__ w, const int64_t* __restrict__ off,
                                                        const int64_t* __restrict__ cnt, int C, int64_t P, float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const int64_t q0 = off[p];
    const int n = (int)cnt[p];
    float wsum = 0.f;
    for (int j = 0; j < n; ++j) wsum += w[q0 + j];
    for (int c = lane * 4; c < C; c += 256) {
        if (BWD) {      // src = dagg [P, C] -> dst = dlocal [Q, C]
            const f32x4 g = *reinterpret_cast<const f32x4*>(src + p * C + c);
            for (int j = 0; j < n; ++j) *reinterpret_cast<f32x4*>(dst + (q0 + j) * C + c) = g * (w[q0 + j] / wsum);
        } else {        // src = local [Q, C] -> dst = agg [P, C]
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < n; ++j) acc += *reinterpret_cast<const f32x4*>(src + (q0 + j) * C + c) * (w[q0 + j] / wsum);
            *reinterpret_cast<f32x4*>(dst + p * C + c) = acc;
        }
    }
}

// LeakyReLU backward + bias-gradient column partials in one pass (utils/model.py:22-36: Linear -> LeakyReLU(0.01)):
//   dy = dz * (z > 0 ? 1 : slope)   (z = the activation OUTPUT: same sign as its input for slope > 0),
//   part[workgroup][c] = sum over the workgroup's rows of dy[:, c]  (then npcd_colsum_finalize: fixed order).
// A thread owns 16 bytes of a row; the workgroup's 256 threads are (N * sizeof(T) / 16) column threads x row lanes, the row
// lanes of all workgroups interleave over the rows (one contiguous band at a time, cf. the column-sum kernels of the denoiser).
template <class T>
__global__ __launch_bounds__(256) void leaky_bwd_colsum_kernel(const T* __restrict__ dz, const T* __restrict__ z, T* __restrict__ dy,
                                                               float* __restrict__ part, int64_t R, int N, float slope) {
    constexpr int VEC = 16 / sizeof(T);
    typedef T TV __attribute__((ext_vector_type(VEC)));
    extern __shared__ float red[];                   // [row lanes][N]
    const int CT = N / VEC, RL = 256 / CT;
    const int ct = threadIdx.x % CT, rl = threadIdx.x / CT, col = ct * VEC;
    const int64_t slots = (int64_t)gridDim.x * RL;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * RL + rl; row < R; row += slots) {
        const TV g = *reinterpret_cast<const TV*>(dz + row * N + col);
        const TV a = *reinterpret_cast<const TV*>(z + row * N + col);
        TV o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            o[j] = (T)((float)g[j] * ((float)a[j] > 0.f ? 1.f : slope));
            acc[j] += (float)o[j];
        }
        *reinterpret_cast<TV*>(dy + row * N + col) = o;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[rl * N + col + j] = acc[j];
    __syncthreads();
    if (rl == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float v = 0.f;
            for (int q = 0; q < RL; ++q) v += red[q * N + col + j];
            part[(int64_t)blockIdx.x * N + col + j] = v;
        }
    }
}

}  // namespace npcd

using namespace npcd;

extern "C" int npcd_leaky_bwd_blocks(int64_t rows) {
    const int64_t b = (rows + 63) / 64;
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int npcd_leaky_bwd_colsum(const void* dz, const void* z, void* dy, float* part, int64_t rows, int N, float slope, int dtype,
                                     void* stream) {
    if (!dz || !z || !dy || !part || rows <= 0 || N <= 0) return NPCD_ERR_ARG;
    const int esz = dtype == NPCD_F32 ? 4 : 2;
    if (dtype != NPCD_F32 && dtype != NPCD_BF16) return NPCD_ERR_UNSUPPORTED;
    const int ct = N * esz / 16;
    if ((N * esz) % 16 != 0 || ct < 1 || ct > 256 || 256 % ct != 0) return NPCD_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dz) & 15) || (reinterpret_cast<uintptr_t>(z) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15)) return NPCD_ERR_UNSUPPORTED;
    const dim3 grid(npcd_leaky_bwd_blocks(rows));
    const size_t lds = (size_t)(256 / ct) * N * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == NPCD_F32)
        hipLaunchKernelGGL(leaky_bwd_colsum_kernel<float>, grid, dim3(256), lds, st, static_cast<const float*>(dz), static_cast<const float*>(z),
                           static_cast<float*>(dy), part, rows, N, slope);
    else
        hipLaunchKernelGGL(leaky_bwd_colsum_kernel<__bf16>, grid, dim3(256), lds, st, static_cast<const __bf16*>(dz), static_cast<const __bf16*>(z),
                           static_cast<__bf16*>(dy), part, rows, N, slope);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_pair_input_fwd(const int64_t* flat, const int64_t* owner, const float* pts, const float* kp_pos, const float* kp_feat,
                                   int feat_dim, int n_freqs, int64_t n_pairs, float* x0, float* w, void* stream) {
    if (n_pairs == 0) return NPCD_OK;
    if (!flat || !owner || !pts || !kp_pos || !kp_feat || !x0 || !w || n_pairs < 0 || feat_dim <= 0 || n_freqs < 0) return NPCD_ERR_ARG;
    hipLaunchKernelGGL(pair_input_fwd_kernel, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), flat, owner, pts,
                       kp_pos, kp_feat, feat_dim, n_freqs, n_pairs, x0, w);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_pair_input_bwd(const int64_t* flat, const float* dx0, int feat_dim, int n_cols, int64_t n_pairs, float* dfeat, void* stream) {
    if (n_pairs == 0) return NPCD_OK;
    if (!flat || !dx0 || !dfeat || n_pairs < 0 || feat_dim <= 0 || n_cols < feat_dim) return NPCD_ERR_ARG;
    hipLaunchKernelGGL(pair_input_bwd_kernel, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), flat, dx0, feat_dim,
                       n_cols, n_pairs, dfeat);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_pair_aggregate(int backward, const float* src, const float* w, const int64_t* off, const int64_t* cnt, int channels,
                                   int64_t n_points, float* dst, void* stream) {
    if (n_points == 0) return NPCD_OK;
    if (!src || !w || !off || !cnt || !dst || n_points < 0 || channels <= 0) return NPCD_ERR_ARG;
    if (channels % 4 != 0 || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return NPCD_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((n_points + 3) / 4));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (backward) hipLaunchKernelGGL(aggregate_kernel<true>, grid, dim3(256), 0, st, src, w, off, cnt, channels, n_points, dst);
    else hipLaunchKernelGGL(aggregate_kernel<false>, grid, dim3(256), 0, st, src, w, off, cnt, channels, n_points, dst);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
