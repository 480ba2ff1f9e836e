// Geometry kernels of the PointNeRF render path (gfx950): ray generation + box limits, voxel-grid
// build, neighbour query, ray marching.  HBM-bound / latency-bound integer + fp32 work: no MFMA.
//
// This file is compiled with -ffp-contract=off: the integer outputs of the neighbour query must be
// bit-exact against oracle/voxel_grid.py, which evaluates every fp32 step as a separately rounded
// operation in a fixed order.
//
// Reference call sites replaced:
//   ray generation      npcd/models/pointnerf/renderers/ray_sampler.py:10-49
//   box limits          renderers/math_utils.py:46-97, renderers/renderer.py:36-47
//   depth samples       renderers/renderer.py:49-77, math_utils.py:100-117
//   VoxelGrid           torch_knnquery (pointnerf.py:20,67-75; aggregator.py:42-73)
//   depth / ray march   renderers/renderer.py:96-110,120-185, volume_renderer.py:23-39
#include <math.h>

#include <stdlib.h>

#include "common.h"

namespace npcd {

// monotone float <-> uint32 key (for atomicMin/atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// ============================================================================================
// rays
// ============================================================================================

// One camera ray and its limits against the cube [-box, box]^3: E = world2cam [4 x 4], K = intrinsics [3 x 3], `ray` = row-major pixel
// number.  tmin = -1, tmax = -2 for a ray that misses.  Shared by ray_gen_kernel and the fused query of a render
// (grid_query_wave_kernel with QueryArgs::extr): the same operations in the same order, i.e. the same bits.
__device__ __forceinline__ void gen_ray(const float* __restrict__ E, const float* __restrict__ K, int res, float box, int ray, float (&o)[3],
                                        float (&d)[3], float& tmin, float& tmax) {
    tmin = -1.f;
    tmax = -2.f;
    const float fx = K[0], sk = K[1], cx = K[2], fy = K[4], cy = K[5];
    const float x = (float)(ray % res) + 0.5f, y = (float)(ray / res) + 0.5f;
    // ray_sampler.py:27-28
    const float xl = (x - cx + cy * sk / fy - sk * y / fy) / fx;
    const float yl = (y - cy) / fy;
    // cam2world = [R^T | -R^T t]  (ray_sampler.py:35-39); world = cam2world * (xl, yl, 1, 1)
    float w[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float r0 = E[0 * 4 + a], r1 = E[1 * 4 + a], r2 = E[2 * 4 + a];  // row a of R^T
        o[a] = -(r0 * E[3] + r1 * E[7] + r2 * E[11]);
        w[a] = r0 * xl + r1 * yl + r2 * 1.f + o[a] * 1.f;
    }
    d[0] = w[0] - o[0]; d[1] = w[1] - o[1]; d[2] = w[2] - o[2];
    const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);  // F.normalize eps
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = d[a] / nrm;
    // slab test (math_utils.py:46-97)
    float lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = 1.f / d[a];
        const bool neg = inv < 0.f;
        lo[a] = ((neg ? box : -box) - o[a]) * inv;
        hi[a] = ((neg ? -box : box) - o[a]) * inv;
    }
    bool ok = true;
    float a0 = lo[0], a1 = hi[0];
    if (a0 > hi[1] || lo[1] > a1) ok = false;
    a0 = fmaxf(a0, lo[1]);   // torch.max/min propagate NaN; NaN only arises for degenerate rays
    a1 = fminf(a1, hi[1]);
    if (a0 > hi[2] || lo[2] > a1) ok = false;
    a0 = fmaxf(a0, lo[2]);
    a1 = fminf(a1, hi[2]);
    if (ok) { tmin = a0; tmax = a1; }
}

// pixel_ids (may be NULL): the n_ids pixels (row-major ids, the same for every view) to generate rays for -- the training
// path renders ~100 random pixels per view; without it all res^2 pixels of every view are generated
__global__ __launch_bounds__(256) void ray_gen_kernel(const float* __restrict__ extr, const float* __restrict__ intr, int V, int res,
                                                      float box, const int32_t* __restrict__ pixel_ids, int n_ids, float* __restrict__ rays_o,
                                                      float* __restrict__ rays_d, float* __restrict__ t0, float* __restrict__ t1, uint32_t* ws) {
    const int R = pixel_ids ? n_ids : res * res;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool hit = false;
    float tmin = -1.f, tmax = -2.f;
    if (gid < (int64_t)V * R) {
        const int v = (int)(gid / R);
        const int ray = pixel_ids ? pixel_ids[gid % R] : (int)(gid % R);
        float o[3], d[3];
        gen_ray(extr + v * 16, intr + v * 9, res, box, ray, o, d, tmin, tmax);
        hit = tmax > tmin;
        rays_o[gid * 3 + 0] = o[0]; rays_o[gid * 3 + 1] = o[1]; rays_o[gid * 3 + 2] = o[2];
        rays_d[gid * 3 + 0] = d[0]; rays_d[gid * 3 + 1] = d[1]; rays_d[gid * 3 + 2] = d[2];
        t0[gid] = tmin;
        t1[gid] = tmax;
    }
    // global min(start) / max(end) over the rays that hit (renderer.py:40-43): wave reduce, one atomic per wave
    uint32_t kmin = hit ? fkey(tmin) : 0xffffffffu, kmax = hit ? fkey(tmax) : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
    }
    // one (min, max) pair per workgroup behind the four result words; ray_limits_fix_kernel combines them (no atomics on one
    // cache line, no initialisation launch)
    __shared__ uint32_t wg_lo[4], wg_hi[4];
    if ((threadIdx.x & 63) == 0) { wg_lo[threadIdx.x >> 6] = kmin; wg_hi[threadIdx.x >> 6] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws[4 + 2 * blockIdx.x] = min(min(wg_lo[0], wg_lo[1]), min(wg_lo[2], wg_lo[3]));
        ws[5 + 2 * blockIdx.x] = max(max(wg_hi[0], wg_hi[1]), max(wg_hi[2], wg_hi[3]));
    }
}

// every workgroup combines the per-workgroup limits ray_gen_kernel left in ws[4 ..] (gridDim.x pairs: the two kernels share their
// grid); workgroup 0 leaves min / max / "any ray hit" in ws[0 .. 2]
__global__ __launch_bounds__(256) void ray_limits_fix_kernel(int64_t n, float* __restrict__ t0, float* __restrict__ t1, uint32_t* ws) {
    __shared__ uint32_t red_lo[4], red_hi[4];
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) {
        kmin = min(kmin, ws[4 + 2 * i]);
        kmax = max(kmax, ws[5 + 2 * i]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
    }
    if ((threadIdx.x & 63) == 0) { red_lo[threadIdx.x >> 6] = kmin; red_hi[threadIdx.x >> 6] = kmax; }
    __syncthreads();
    kmin = min(min(red_lo[0], red_lo[1]), min(red_lo[2], red_lo[3]));
    kmax = max(max(red_hi[0], red_hi[1]), max(red_hi[2], red_hi[3]));
    if (blockIdx.x == 0 && threadIdx.x == 0) { ws[0] = kmin; ws[1] = kmax; ws[2] = kmax != 0u ? 1u : 0u; ws[3] = 0u; }
    if (gid >= n || kmax == 0u) return;
    if (!(t1[gid] > t0[gid])) {
        t0[gid] = fkey_inv(kmin);
        t1[gid] = fkey_inv(kmax);
    }
}

// ============================================================================================
// voxel grid
// ============================================================================================
// packed point record: ix | iy<<10 | iz<<20 | kept<<30
__device__ __forceinline__ int pack_coord(int ix, int iy, int iz, bool kept) { return ix | (iy << 10) | (iz << 20) | ((int)kept << 30); }

struct FineCoord {
    int c[3];
    bool ok;
};
__device__ __forceinline__ FineCoord fine_coord(const npcd_grid_params& g, float x, float y, float z) {
    FineCoord f;
    const float p[3] = {x, y, z};
    f.ok = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float q = floorf((p[a] - g.range_min[a]) / g.voxel_size[a]);
        const bool in = (q >= 0.f) && (q < (float)g.dims[a]);  // NaN -> false
        f.ok = f.ok && in;
        f.c[a] = in ? (int)q : 0;
    }
    return f;
}

static inline int occ_words(const npcd_grid_params& g) { return (g.cdims[0] * g.cdims[1] * g.cdims[2] + 31) / 32; }

// Dense fine-voxel -> point table (4 x int16 per voxel, -1 = empty): the kept points of a voxel, in index
// order.  With it a query only tests the <= 4*27 points of the 3^3 neighbourhood instead of scanning the cloud.
// Available when <= 4 points per voxel are kept, the kernel window fits one wave and the table is <= 32 MB.
static inline int64_t table_voxels(const npcd_grid_params& g) { return (int64_t)g.dims[0] * g.dims[1] * g.dims[2]; }
static inline bool table_ok(const npcd_grid_params& g) {
    return g.max_points_per_voxel <= 4 && g.kernel_size[0] * g.kernel_size[1] * g.kernel_size[2] <= 64 && table_voxels(g) * 8 <= (32ll << 20);
}

// one workgroup per example.  LDS: lin[N] int32, flag[N] uint8-as-int, bitmap[occ_words]
__global__ __launch_bounds__(256) void grid_build_kernel(npcd_grid_params g, const float* __restrict__ points, const int32_t* __restrict__ counts,
                                                         int N, int nwords, int32_t* __restrict__ pcoord, uint32_t* __restrict__ occ,
                                                         int16_t* __restrict__ table) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    int* lin = reinterpret_cast<int*>(dsmem);
    int* first = lin + N;
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(first + N);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(counts ? counts[b] : N, N);
    const float* P = points + (int64_t)b * N * 3;
    for (int w = tid; w < nwords; w += blockDim.x) bitmap[w] = 0u;
    for (int i = tid; i < N; i += blockDim.x) {
        int l = -1;
        if (i < n) {
            const FineCoord f = fine_coord(g, P[i * 3 + 0], P[i * 3 + 1], P[i * 3 + 2]);
            if (f.ok) l = (f.c[0] * g.dims[1] + f.c[1]) * g.dims[2] + f.c[2];
        }
        lin[i] = l;
    }
    __syncthreads();
    // rank among the points of the same fine voxel, in ascending point index
    for (int i = tid; i < N; i += blockDim.x) {
        const int l = lin[i];
        int rank = 0;
        if (l >= 0)
            for (int j = 0; j < i; ++j) rank += (lin[j] == l);
        first[i] = (l >= 0) ? rank : -1;
    }
    __syncthreads();
    const bool need_cap = n > g.max_occ_voxels_per_example;  // otherwise #occupied voxels <= n <= cap
    for (int i = tid; i < N; i += blockDim.x) {
        const int l = lin[i], rank = first[i];
        bool kept = (l >= 0) && (rank < g.max_points_per_voxel);
        if (kept && need_cap) {
            int vrank = 0;  // number of distinct occupied voxels with a smaller linear id
            for (int j = 0; j < N; ++j) vrank += (first[j] == 0 && lin[j] < l);
            kept = vrank < g.max_occ_voxels_per_example;
        }
        int cz = 0, cy = 0, cx = 0;
        if (l >= 0) {
            cz = l % g.dims[2];
            cy = (l / g.dims[2]) % g.dims[1];
            cx = l / (g.dims[2] * g.dims[1]);
        }
        pcoord[(int64_t)b * N + i] = pack_coord(cx, cy, cz, kept);
        if (kept && table) table[((int64_t)b * g.dims[0] * g.dims[1] * g.dims[2] + l) * 4 + rank] = (int16_t)i;
        if (kept) {
            const int ccx = cx / g.voxel_scale[0], ccy = cy / g.voxel_scale[1], ccz = cz / g.voxel_scale[2];
            const int hx = (g.kernel_size[0] - 1) / 2, hy = (g.kernel_size[1] - 1) / 2, hz = (g.kernel_size[2] - 1) / 2;
            for (int dx = -hx; dx <= hx; ++dx)
                for (int dy = -hy; dy <= hy; ++dy)
                    for (int dz = -hz; dz <= hz; ++dz) {
                        const int ox = ccx + dx, oy = ccy + dy, oz = ccz + dz;
                        if (ox < 0 || oy < 0 || oz < 0 || ox >= g.cdims[0] || oy >= g.cdims[1] || oz >= g.cdims[2]) continue;
                        const int bit = (ox * g.cdims[1] + oy) * g.cdims[2] + oz;
                        atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
                    }
        }
    }
    __syncthreads();
    for (int w = tid; w < nwords; w += blockDim.x) occ[(int64_t)b * nwords + w] = bitmap[w];
}

// top-8 list kept sorted by (dist^2, index); candidates arrive in ascending index order, so a
// strict '<' insertion keeps the lower index first on ties.
struct Top8 {
    float d[8];
    int i[8];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int t = 0; t < 8; ++t) { d[t] = INFINITY; i[t] = -1; }
    }
    __device__ __forceinline__ void insert(float dist, int idx) {
        d[7] = dist;
        i[7] = idx;
#pragma unroll
        for (int t = 7; t > 0; --t) {
            if (d[t] < d[t - 1]) {
                const float td = d[t]; d[t] = d[t - 1]; d[t - 1] = td;
                const int ti = i[t]; i[t] = i[t - 1]; i[t - 1] = ti;
            }
        }
    }
};

struct QueryArgs {
    npcd_grid_params g;
    const int32_t* pcoord;
    const uint32_t* occ;
    const int16_t* table;   // dense voxel -> points table or nullptr
    const float* points;
    int B, N, R, S, M, k, nwords;
    float r2;
    const float *x, *rays_o, *rays_d, *t0, *t1;
    int32_t* sample_idx;
    float* sample_loc;
    int32_t* slot_sample;
    int32_t* nsel;
    int rpw;                // rays per wave of grid_query_wave_kernel (0 / 1: one)
    // fused ray generation (npcd_render_rays_query; round 6): when extr is set the kernel computes its ray from the camera of view
    // r / (res res) of example b instead of reading rays_o / rays_d / t0 / t1, and lane 0 writes it to gen_* for the later stages
    // (t0 = -1, t1 = -2 for a ray that misses the cube: the march substitutes the global end, see ray_march_wave_kernel)
    const float *extr, *intr;
    int res, views;         // pixels per image edge, views per example (R = views * res * res)
    float box;
    float *gen_o, *gen_d, *gen_t0, *gen_t1;
};

// position of depth sample s of a ray: renderer.py:49-77 (eval) + volume_renderer.py:70
__device__ __forceinline__ void sample_pos(const QueryArgs& a, int64_t ray, int s, const float o[3], const float d[3], float t0, float t1, float p[3]) {
    if (a.x) {
        const float* xp = a.x + (ray * a.S + s) * 3;
        p[0] = xp[0]; p[1] = xp[1]; p[2] = xp[2];
    } else {
        const float step = (float)s / (float)(a.S - 1);
        const float depth = t0 + step * (t1 - t0);
        p[0] = o[0] + depth * d[0];
        p[1] = o[1] + depth * d[1];
        p[2] = o[2] + depth * d[2];
    }
}

// scan all points of the example (LDS resident, broadcast reads) for one query position
template <bool GRID>
__device__ __forceinline__ void scan_points(const QueryArgs& a, const float4* pts, const float p[3], const FineCoord& fc, Top8& top) {
    const int hx = (a.g.kernel_size[0] - 1) / 2, hy = (a.g.kernel_size[1] - 1) / 2, hz = (a.g.kernel_size[2] - 1) / 2;
    for (int j = 0; j < a.N; ++j) {
        const float4 q = pts[j];
        const float dx = p[0] - q.x, dy = p[1] - q.y, dz = p[2] - q.z;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        bool cand = d2 < a.r2;
        if (GRID) {
            const int pc = __float_as_int(q.w);
            const int ix = pc & 1023, iy = (pc >> 10) & 1023, iz = (pc >> 20) & 1023;
            cand = cand && ((pc >> 30) & 1) && (abs(ix - fc.c[0]) <= hx) && (abs(iy - fc.c[1]) <= hy) && (abs(iz - fc.c[2]) <= hz);
        }
        if (cand && d2 < top.d[7]) top.insert(d2, j);
    }
}

// one wave per ray, 4 rays per workgroup (all of the same example when R % 4 == 0; otherwise the
// workgroup reloads the point set per wave -- handled by making the grid per example).
template <bool GRID>
__global__ __launch_bounds__(256) void grid_query_kernel(QueryArgs a, int blocks_per_example) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    float4* pts = reinterpret_cast<float4*>(dsmem);
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(pts + a.N);
    int* sel = reinterpret_cast<int*>(bitmap + a.nwords);  // [4][64] selected sample ids per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / blocks_per_example;
    const int rb = blockIdx.x % blocks_per_example;
    for (int j = tid; j < a.N; j += blockDim.x) {
        const float* P = a.points + ((int64_t)b * a.N + j) * 3;
        pts[j] = make_float4(P[0], P[1], P[2], __int_as_float(GRID ? a.pcoord[(int64_t)b * a.N + j] : 0));
    }
    if (GRID)
        for (int w = tid; w < a.nwords; w += blockDim.x) bitmap[w] = a.occ[(int64_t)b * a.nwords + w];
    __syncthreads();
    const int r = rb * 4 + wave;
    if (r >= a.R) return;
    const int64_t ray = (int64_t)b * a.R + r;
    float o[3] = {0, 0, 0}, d[3] = {0, 0, 0}, t0 = 0, t1 = 0;
    if (!a.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[c] = a.rays_o[ray * 3 + c]; d[c] = a.rays_d[ray * 3 + c]; }
        t0 = a.t0[ray];
        t1 = a.t1[ray];
    }
    int* mysel = sel + wave * 64;
    int32_t* out_idx = a.sample_idx + ray * a.M * a.k;
    float* out_loc = a.sample_loc + ray * a.M * 3;
    int32_t* out_ss = a.slot_sample + ray * a.M;
    int nsel = 0;

    if (GRID) {
        // pass A: occupancy of every depth sample, first M occupied samples -> slots
        for (int s0 = 0; s0 < a.S && nsel < a.M; s0 += 64) {
            const int s = s0 + lane;
            bool occ = false;
            if (s < a.S) {
                float p[3];
                sample_pos(a, ray, s, o, d, t0, t1, p);
                const FineCoord fc = fine_coord(a.g, p[0], p[1], p[2]);
                if (fc.ok) {
                    const int bit = ((fc.c[0] / a.g.voxel_scale[0]) * a.g.cdims[1] + fc.c[1] / a.g.voxel_scale[1]) * a.g.cdims[2] + fc.c[2] / a.g.voxel_scale[2];
                    occ = (bitmap[bit >> 5] >> (bit & 31)) & 1u;
                }
            }
            const unsigned long long m = __ballot(occ);
            const int slot = nsel + __popcll(m & ((1ull << lane) - 1ull));
            if (occ && slot < a.M) mysel[slot] = s;
            nsel = min(a.M, nsel + __popcll(m));
        }
        // pass B: lane j handles slot j
        const bool active = lane < nsel;
        Top8 top;
        top.init();
        float p[3] = {0, 0, 0};
        int s = -1;
        if (active) {
            s = mysel[lane];
            sample_pos(a, ray, s, o, d, t0, t1, p);
            const FineCoord fc = fine_coord(a.g, p[0], p[1], p[2]);
            scan_points<true>(a, pts, p, fc, top);
        }
        if (lane < a.M) {
            const int gbase = b * a.N;
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (t < a.k) out_idx[lane * a.k + t] = (active && top.i[t] >= 0) ? gbase + top.i[t] : -1;
            out_loc[lane * 3 + 0] = active ? p[0] : 0.f;
            out_loc[lane * 3 + 1] = active ? p[1] : 0.f;
            out_loc[lane * 3 + 2] = active ? p[2] : 0.f;
            out_ss[lane] = s;
        }
    } else {
        // brute-force branch: a sample is valid iff it has a neighbour within the radius; the first M
        // valid samples fill slots 0.. (aggregator.py:42-58)
        for (int s0 = 0; s0 < a.S && nsel < a.M; s0 += 64) {
            const int s = s0 + lane;
            Top8 top;
            top.init();
            float p[3] = {0, 0, 0};
            if (s < a.S) {
                sample_pos(a, ray, s, o, d, t0, t1, p);
                FineCoord fc{};
                scan_points<false>(a, pts, p, fc, top);
            }
            const bool valid = top.i[0] >= 0;
            const unsigned long long m = __ballot(valid);
            const int slot = nsel + __popcll(m & ((1ull << lane) - 1ull));
            if (valid && slot < a.M) {
                const int gbase = b * a.N;
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (t < a.k) out_idx[slot * a.k + t] = (top.i[t] >= 0) ? gbase + top.i[t] : -1;
                out_loc[slot * 3 + 0] = p[0]; out_loc[slot * 3 + 1] = p[1]; out_loc[slot * 3 + 2] = p[2];
                out_ss[slot] = s;
            }
            nsel = min(a.M, nsel + __popcll(m));
        }
        for (int slot = nsel + lane; slot < a.M; slot += 64) {
            for (int t = 0; t < a.k; ++t) out_idx[slot * a.k + t] = -1;
            out_loc[slot * 3 + 0] = 0.f; out_loc[slot * 3 + 1] = 0.f; out_loc[slot * 3 + 2] = 0.f;
            out_ss[slot] = -1;
        }
    }
    if (lane == 0) a.nsel[ray] = nsel;
}

// ---- wave-cooperative neighbour search (grid semantics) -------------------------------------------
// One wave per ray.  Pass A as above (occupancy -> first M selected samples).  Pass B walks the selected
// samples one at a time and lets the 64 LANES test 64 points in parallel (conflict-free 16-byte LDS reads
// instead of 512 serial broadcast reads per lane); the few candidates inside the radius (ballot mask) are
// inserted one by one into a sorted top-k list that lives across lanes 0..7 (lane t = t-th best), keyed by
// (dist^2 bits << 32 | point index): dist^2 >= 0, so the unsigned order of the key is exactly the
// (dist^2, index) order of the spec.  ~5x fewer instructions than one lane per sample.
struct CompactOut {
    int32_t* counter;     // [4]: number of compact points, overflow flag, shading status (zeroed), reserved (zeroed)
    int32_t capacity;     // rows available in nb / pts
    int32_t* ray_base;    // [B*R]
    int32_t* ray_nsel;    // [B*R]
    unsigned long long* ray_bits;  // [B*R] valid-slot mask
    int32_t* nb;          // [capacity][k]
    float* pts;           // [capacity][3]
    // ordered form (npcd_grid_query_compact_ordered): the query kernel leaves a ray's rows at [ray][0 .. cnt) of a per-ray staging
    // area and its count in ray_cnt; compact_ordered_kernel turns the counts into bases by a prefix sum IN RAY ORDER and moves the
    // rows -- no atomics, the lists are bit-identical from run to run
    int32_t* ray_cnt;     // [B*R] or nullptr (atomic form)
    int32_t* st_nb;       // [B*R][M][k]
    float* st_pts;        // [B*R][M][3]
    // fused render (round 6): compact_ordered_kernel also reduces the start / end of the rays that hit the cube over its 64 rays into
    // lim_part[2 j], lim_part[2 j + 1] (monotone keys; 0xffffffff / 0 when none hit) -- what ray_limits_fix_kernel did in a launch of
    // its own; the march combines the pairs (renderer.py:40-43)
    const float *lim_t0, *lim_t1;
    uint32_t* lim_part;   // [2 * ceil(B R / 64)] or nullptr
};

template <bool COMPACT>
__global__ __launch_bounds__(256) void grid_query_wave_kernel(QueryArgs a, CompactOut co, int blocks_per_example) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dsmem[];
    float4* pts = reinterpret_cast<float4*>(dsmem);
    uint32_t* bitmap = reinterpret_cast<uint32_t*>(pts + a.N);
    int* sel = reinterpret_cast<int*>(bitmap + a.nwords);          // [4][64]
    // [4][64] position + packed cell (x | y << 10 | z << 20) of every selected sample, written by pass A: pass B used to
    // recompute both per sample on all 64 lanes (two IEEE divisions per axis: ~80 of its ~300 instructions per trip)
    float4* selp = reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(sel + 4 * 64) + ((16 - ((a.N * 16 + a.nwords * 4 + 4 * 64 * 4) & 15)) & 15));
    unsigned long long* cpk = reinterpret_cast<unsigned long long*>(selp + 4 * 64);   // [4][64] a slot pair's in-radius candidates, packed
    int* stage_idx = reinterpret_cast<int*>(cpk + 4 * 64);          // [4][64][8]   (COMPACT)
    float* stage_pos = reinterpret_cast<float*>(stage_idx + 4 * 64 * 8);  // [4][64][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / blocks_per_example;
    const int rb = blockIdx.x % blocks_per_example;
    for (int j = tid; j < a.N; j += blockDim.x) {
        const float* P = a.points + ((int64_t)b * a.N + j) * 3;
        pts[j] = make_float4(P[0], P[1], P[2], __int_as_float(a.pcoord[(int64_t)b * a.N + j]));
    }
    for (int w = tid; w < a.nwords; w += blockDim.x) bitmap[w] = a.occ[(int64_t)b * a.nwords + w];
    __syncthreads();
    // a.rpw rays per wave, one after the other (round 6 experiment, default 1: more than one measured slower, see the launch code).
    // Every ray is computed by the same code on wave-private LDS: the lists are the same bits whatever rpw is.
    const int rpw = a.rpw > 0 ? a.rpw : 1;
    for (int rr = 0; rr < rpw; ++rr) {
    const int r = (rb * rpw + rr) * 4 + wave;
    if (r >= a.R) break;
    const int64_t ray = (int64_t)b * a.R + r;
    float o[3] = {0, 0, 0}, d[3] = {0, 0, 0}, t0 = 0, t1 = 0;
    bool ray_hits = true;
    if (a.extr) {
        // the ray of pixel r % (res res) of view r / (res res) of this example, computed by every lane (wave-uniform values)
        const int rv = a.res * a.res, view = b * a.views + r / rv;
        gen_ray(a.extr + view * 16, a.intr + view * 9, a.res, a.box, r % rv, o, d, t0, t1);
        ray_hits = t1 > t0;           // a ray that misses the cube has no sample inside the grid's range (the launch checks box >= range)
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { a.gen_o[ray * 3 + c] = o[c]; a.gen_d[ray * 3 + c] = d[c]; }
            a.gen_t0[ray] = t0;
            a.gen_t1[ray] = t1;
        }
    } else if (!a.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[c] = a.rays_o[ray * 3 + c]; d[c] = a.rays_d[ray * 3 + c]; }
        t0 = a.t0[ray];
        t1 = a.t1[ray];
    }
    int* mysel = sel + wave * 64;
    float4* myselp = selp + wave * 64;
    const bool unit_scale = (a.g.voxel_scale[0] | a.g.voxel_scale[1] | a.g.voxel_scale[2]) == 1;      // the scaled reading
    int nsel = 0;
    for (int s0 = 0; ray_hits && s0 < a.S && nsel < a.M; s0 += 64) {
        const int s = s0 + lane;
        bool occ = false;
        float p[3] = {0.f, 0.f, 0.f};
        int cell = 0;
        if (s < a.S) {
            sample_pos(a, ray, s, o, d, t0, t1, p);
            const FineCoord fc = fine_coord(a.g, p[0], p[1], p[2]);
            if (fc.ok) {
                int cc[3] = {fc.c[0], fc.c[1], fc.c[2]};
                if (!unit_scale) {                     // (wave-uniform; a division by a run-time integer is ~25 instructions)
                    cc[0] /= a.g.voxel_scale[0]; cc[1] /= a.g.voxel_scale[1]; cc[2] /= a.g.voxel_scale[2];
                }
                const int bit = (cc[0] * a.g.cdims[1] + cc[1]) * a.g.cdims[2] + cc[2];
                occ = (bitmap[bit >> 5] >> (bit & 31)) & 1u;
                cell = fc.c[0] | (fc.c[1] << 10) | (fc.c[2] << 20);
            }
        }
        const unsigned long long m = __ballot(occ);
        const int slot = nsel + __popcll(m & ((1ull << lane) - 1ull));
        if (occ && slot < a.M) {
            mysel[slot] = s;
            myselp[slot] = make_float4(p[0], p[1], p[2], __int_as_float(cell));
        }
        nsel = min(a.M, nsel + __popcll(m));
    }
    // what pass A left for slot `slot`: position and cell (a selected sample is in range by construction)
    auto slot_geom = [&](int slot, float (&p)[3], FineCoord& fc) {
        const float4 g4 = myselp[slot];                 // same-wave LDS write -> read is ordered
        p[0] = g4.x; p[1] = g4.y; p[2] = g4.z;
        const int cell = __float_as_int(g4.w);
        fc.c[0] = cell & 1023; fc.c[1] = (cell >> 10) & 1023; fc.c[2] = (cell >> 20) & 1023;
        fc.ok = true;
    };
    const int hx = (a.g.kernel_size[0] - 1) / 2, hy = (a.g.kernel_size[1] - 1) / 2, hz = (a.g.kernel_size[2] - 1) / 2;
    const int gbase = b * a.N;
    int32_t* out_idx = COMPACT ? nullptr : a.sample_idx + ray * a.M * a.k;
    float* out_loc = COMPACT ? nullptr : a.sample_loc + ray * a.M * 3;
    int32_t* out_ss = COMPACT ? nullptr : a.slot_sample + ray * a.M;
    unsigned long long valid_bits = 0ull;
    auto process = [&](int slot, u32x2 raw_in) {
        const int s = mysel[slot];                     // same-wave LDS write -> read is ordered
        float p[3];
        FineCoord fc;
        slot_geom(slot, p, fc);                        // wave-uniform values
        uint32_t key_hi = 0xffffffffu, key_lo = 0xffffffffu;   // lanes 0..7 hold the sorted list
        // insert the lanes flagged in `cm` (dist^2 in d2, point index in jj) into the cross-lane sorted list
        auto insert_all = [&](unsigned long long cm, float d2, int jj) {
            while (cm) {                                // wave-uniform loop over the (few) candidates
                const int bl = __ffsll((long long)cm) - 1;
                cm &= cm - 1;
                const uint32_t ch = __builtin_amdgcn_readlane(__float_as_uint(d2), bl);
                const uint32_t cl = (uint32_t)__builtin_amdgcn_readlane(jj, bl);
                const bool le = (key_hi < ch) || (key_hi == ch && key_lo < cl);   // existing entry sorts before the candidate
                const int pos = __popcll(__ballot(le) & 0xffull);
                if (pos < 8) {
                    // shift lanes pos..6 up by one with a DPP row shift (lanes 0..7 share a 16-lane row): one VALU op
                    // per word instead of an LDS-crossbar shuffle
                    const uint32_t up_hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key_hi, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
                    const uint32_t up_lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key_lo, 0x111, 0xf, 0xf, false);
                    if (lane == pos) { key_hi = ch; key_lo = cl; }
                    else if (lane > pos) { key_hi = up_hi; key_lo = up_lo; }
                }
            }
        };
        if (a.table) {
            // lanes 0..26: one fine voxel of the 3^3 window each -> its <= 4 kept points
            const int ky = a.g.kernel_size[1], kz = a.g.kernel_size[2], nk = a.g.kernel_size[0] * ky * kz;
            (void)raw_in;
            int cand_idx[4] = {-1, -1, -1, -1};
            if (lane < nk) {
                const int vx = fc.c[0] + lane / (ky * kz) - hx, vy = fc.c[1] + (lane / kz) % ky - hy, vz = fc.c[2] + lane % kz - hz;
                if (vx >= 0 && vy >= 0 && vz >= 0 && vx < a.g.dims[0] && vy < a.g.dims[1] && vz < a.g.dims[2]) {
                    const int64_t vox = (int64_t)b * a.g.dims[0] * a.g.dims[1] * a.g.dims[2] + ((int64_t)vx * a.g.dims[1] + vy) * a.g.dims[2] + vz;
                    const u32x2 raw = *reinterpret_cast<const u32x2*>(a.table + vox * 4);
                    cand_idx[0] = (int)(int16_t)(raw[0] & 0xffffu); cand_idx[1] = (int)(int16_t)(raw[0] >> 16);
                    cand_idx[2] = (int)(int16_t)(raw[1] & 0xffffu); cand_idx[3] = (int)(int16_t)(raw[1] >> 16);
                }
            }
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                const int j = cand_idx[qi];
                bool cand = false;
                float d2 = 0.f;
                if (j >= 0) {
                    const float4 q = pts[j];
                    const float dx = p[0] - q.x, dy = p[1] - q.y, dz = p[2] - q.z;
                    d2 = (dx * dx + dy * dy) + dz * dz;
                    cand = d2 < a.r2;                   // "kept" and voxel adjacency are implied by the table lookup
                }
                insert_all(__ballot(cand), d2, j);
            }
        } else {
            for (int j0 = 0; j0 < a.N; j0 += 64) {
                const int j = j0 + lane;
                bool cand = false;
                float d2 = 0.f;
                if (j < a.N) {
                    const float4 q = pts[j];
                    const float dx = p[0] - q.x, dy = p[1] - q.y, dz = p[2] - q.z;
                    d2 = (dx * dx + dy * dy) + dz * dz;
                    const int pc = __float_as_int(q.w);
                    const int ix = pc & 1023, iy = (pc >> 10) & 1023, iz = (pc >> 20) & 1023;
                    cand = (d2 < a.r2) && ((pc >> 30) & 1) && (abs(ix - fc.c[0]) <= hx) && (abs(iy - fc.c[1]) <= hy) && (abs(iz - fc.c[2]) <= hz);
                }
                insert_all(__ballot(cand), d2, j);
            }
        }
        const bool has = key_lo != 0xffffffffu;
        const int first = __builtin_amdgcn_readfirstlane(has ? 1 : 0);   // lane 0 = best entry
        if (COMPACT) {
            if (lane < 8) stage_idx[(wave * 64 + slot) * 8 + lane] = has ? gbase + (int)key_lo : -1;
            if (lane < 3) stage_pos[(wave * 64 + slot) * 4 + lane] = lane == 0 ? p[0] : (lane == 1 ? p[1] : p[2]);
            if (first) valid_bits |= (1ull << slot);
        } else {
            if (lane < a.k) out_idx[slot * a.k + lane] = has ? gbase + (int)key_lo : -1;
            if (lane < 3) out_loc[slot * 3 + lane] = lane == 0 ? p[0] : (lane == 1 ? p[1] : p[2]);
            if (lane == 0) out_ss[slot] = s;
        }
    };
    // Table path with a window of <= 32 voxels (the reference's 3^3): TWO slots per trip, one per 32-lane half.  Lanes 0..26 of
    // a half fetch its slot's window voxels, the sorted list of a half lives in its lanes 0..7, and one trip of the insertion
    // loop takes one candidate of each half.  Same arithmetic per slot as `process`, i.e. the same bits out.
    // lane constants of the two-slot form: the window cell a lane fetches (relative), this example's table
    const int w2_l = lane & 31, w2_ky = a.g.kernel_size[1], w2_kz = a.g.kernel_size[2], w2_nk = a.g.kernel_size[0] * w2_ky * w2_kz;
    const int w2_dx = w2_l / (w2_ky * w2_kz) - hx, w2_dy = (w2_l / w2_kz) % w2_ky - hy, w2_dz = w2_l % w2_kz - hz;
    const int16_t* w2_table = a.table ? a.table + (int64_t)b * a.g.dims[0] * a.g.dims[1] * a.g.dims[2] * 4 : nullptr;
    auto process2 = [&](int pair) {
        const int h = lane >> 5, l32 = lane & 31;
        const int slot = 2 * pair + h;
        const bool live = slot < nsel;                                  // uniform per half
        const int s = mysel[live ? slot : 2 * pair];
        float p[3];
        FineCoord fc;
        slot_geom(live ? slot : 2 * pair, p, fc);
        int cand_idx[4] = {-1, -1, -1, -1};
        if (live && l32 < w2_nk) {
            const int vx = fc.c[0] + w2_dx, vy = fc.c[1] + w2_dy, vz = fc.c[2] + w2_dz;
            if (vx >= 0 && vy >= 0 && vz >= 0 && vx < a.g.dims[0] && vy < a.g.dims[1] && vz < a.g.dims[2]) {
                const int vox = (vx * a.g.dims[1] + vy) * a.g.dims[2] + vz;        // (the table has < 2^22 cells per example: table_ok)
                const u32x2 raw = *reinterpret_cast<const u32x2*>(w2_table + (int64_t)vox * 4);
                cand_idx[0] = (int)(int16_t)(raw[0] & 0xffffu); cand_idx[1] = (int)(int16_t)(raw[0] >> 16);
                cand_idx[2] = (int)(int16_t)(raw[1] & 0xffffu); cand_idx[3] = (int)(int16_t)(raw[1] >> 16);
            }
        }
        // Selection BY RANK: a candidate's place in the sorted list is the number of candidates of its half with a smaller
        // (dist^2 bits, index) key -- keys are distinct, so ranks are too.  The in-radius candidates (a few of the up to 4 x 27
        // fetched points) are first PACKED to the front of their half through LDS, one per lane; then every lane compares its
        // candidate with all of its half, four broadcast keys per LDS read: ~3 vector instructions per candidate, where the
        // sorted-insertion form shifted an eight-entry list across lanes for every candidate (~40).  Same neighbours in the
        // same order.  (More than 32 candidates in a half -- dense clouds -- take the broadcast loop below.)
        unsigned long long key[4] = {~0ull, ~0ull, ~0ull, ~0ull};
        uint32_t d2b[4];
        unsigned long long cmask[4];
        int n0 = 0, n1 = 0;                                             // candidates of the lower / upper half
        unsigned long long* mycpk = cpk + wave * 64;
        float4 cq[4];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) cq[qi] = pts[max(cand_idx[qi], 0)];      // all four reads in flight (empty entries read point 0)
        bool candf[4];
#pragma unroll
        for (int qi = 0; qi < 4; ++qi) {
            const int jq = cand_idx[qi];
            const float4 q = cq[qi];
            const float dx = p[0] - q.x, dy = p[1] - q.y, dz = p[2] - q.z;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            candf[qi] = jq >= 0 && d2 < a.r2;
            d2b[qi] = __float_as_uint(d2);
            cmask[qi] = __ballot(candf[qi]);
            n0 += __popc((uint32_t)cmask[qi]);
            n1 += __popc((uint32_t)(cmask[qi] >> 32));
        }
        int rank[4] = {0, 0, 0, 0};
        const int nmax = max(n0, n1);
        const bool packed = nmax <= 32;
        unsigned long long mine = ~0ull;                                // packed form: the candidate this lane ranks
        if (nmax > 0) {          // (most selected samples lie in the dilated shell around the cloud and have no candidate at all)
            int b0 = 0, b1 = 0;
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                key[qi] = candf[qi] ? (((unsigned long long)d2b[qi] << 32) | (uint32_t)cand_idx[qi]) : ~0ull;
                const uint32_t c0 = (uint32_t)cmask[qi], c1 = (uint32_t)(cmask[qi] >> 32);
                // my place among the candidates of my half: those of earlier rounds + those of this round on lower lanes
                const int below = (int)__builtin_amdgcn_mbcnt_hi(c1, __builtin_amdgcn_mbcnt_lo(c0, 0u));
                const int pos = (h ? b1 + below - __popc(c0) : b0 + below);
                if (candf[qi] && pos < 32) mycpk[h * 32 + pos] = key[qi];
                b0 += __popc(c0);
                b1 += __popc(c1);
            }
        }
        if (nmax == 0) {
            // nothing to rank
        } else if (packed) {
            if (l32 < (h ? n1 : n0)) mine = mycpk[lane];                // same-wave LDS write -> read is ordered
            const unsigned long long* half = mycpk + h * 32;
            const int nhh = h ? n1 : n0;
            for (int i0 = 0; i0 < nmax; i0 += 4) {                      // wave-uniform
                // four keys of the own half per trip, two broadcast 16-byte reads issued together; entries past the half's count
                // are stale (or belong to the next region of LDS) and are masked AFTER the read -- a guarded read is a branch and
                // an LDS round trip per key
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 ka = *reinterpret_cast<const u64x2*>(half + i0), kb = *reinterpret_cast<const u64x2*>(half + i0 + 2);
                const unsigned long long ck[4] = {ka[0], ka[1], kb[0], kb[1]};
#pragma unroll
                for (int u = 0; u < 4; ++u) rank[0] += (i0 + u < nhh && ck[u] < mine) ? 1 : 0;
            }
        } else {
#pragma unroll
            for (int qi = 0; qi < 4; ++qi) {
                uint32_t c0 = (uint32_t)cmask[qi], c1 = (uint32_t)(cmask[qi] >> 32);
                while (c0 | c1) {                                       // wave-uniform
                    const bool v0 = c0 != 0u, v1 = c1 != 0u;
                    const int b0 = v0 ? __ffs((int)c0) - 1 : 0, b1 = v1 ? __ffs((int)c1) - 1 : 0;
                    c0 &= c0 - 1u;                                      // (0 stays 0)
                    c1 &= c1 - 1u;
                    const uint32_t ch0 = __builtin_amdgcn_readlane(d2b[qi], b0), cl0 = (uint32_t)__builtin_amdgcn_readlane(cand_idx[qi], b0);
                    const uint32_t ch1 = __builtin_amdgcn_readlane(d2b[qi], 32 + b1), cl1 = (uint32_t)__builtin_amdgcn_readlane(cand_idx[qi], 32 + b1);
                    const unsigned long long k0 = v0 ? (((unsigned long long)ch0 << 32) | cl0) : ~0ull;      // scalar
                    const unsigned long long k1 = v1 ? (((unsigned long long)ch1 << 32) | cl1) : ~0ull;
                    const unsigned long long ck = h ? k1 : k0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) rank[q] += (ck < key[q]) ? 1 : 0;
                }
            }
        }
        const int nh = h ? n1 : n0;
        if (COMPACT) {
            if (nmax > 0) {          // (a slot without neighbours is never copied out of the staging rows: nothing to write)
                int* row = stage_idx + (wave * 64 + slot) * 8;
                if (live && l32 < 8 && l32 >= nh) row[l32] = -1;
                if (packed) {
                    if (mine != ~0ull && rank[0] < 8) row[rank[0]] = gbase + (int)(uint32_t)mine;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (key[q] != ~0ull && rank[q] < 8) row[rank[q]] = gbase + cand_idx[q];
                }
                if (live && l32 < 3) stage_pos[(wave * 64 + slot) * 4 + l32] = l32 == 0 ? p[0] : (l32 == 1 ? p[1] : p[2]);
            }
            if (n0 > 0) valid_bits |= (1ull << (2 * pair));
            if (n1 > 0) valid_bits |= (1ull << (2 * pair + 1));          // a dead upper half has no candidates
        } else {
            if (live && l32 < a.k && l32 >= nh) out_idx[slot * a.k + l32] = -1;
            if (packed) {
                if (mine != ~0ull && rank[0] < a.k) out_idx[slot * a.k + rank[0]] = gbase + (int)(uint32_t)mine;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (key[q] != ~0ull && rank[q] < a.k) out_idx[slot * a.k + rank[q]] = gbase + cand_idx[q];
            }
            if (live && l32 < 3) out_loc[slot * 3 + l32] = l32 == 0 ? p[0] : (l32 == 1 ? p[1] : p[2]);
            if (live && l32 == 0) out_ss[slot] = s;
        }
    };
    if (a.table && a.g.kernel_size[0] * a.g.kernel_size[1] * a.g.kernel_size[2] <= 32) {
        for (int pair = 0; 2 * pair < nsel; ++pair) process2(pair);
    } else {
        for (int slot = 0; slot < nsel; ++slot) process(slot, u32x2{0u, 0u});
    }
    if (COMPACT && co.ray_cnt) {
        // ordered form: rows to this ray's staging area, count to ray_cnt; bases come from compact_ordered_kernel
        const int cnt = __popcll(valid_bits);
        if (lane < nsel && ((valid_bits >> lane) & 1ull)) {
            const int64_t row = ray * a.M + __popcll(valid_bits & ((1ull << lane) - 1ull));
            for (int t = 0; t < a.k; ++t) co.st_nb[row * a.k + t] = stage_idx[(wave * 64 + lane) * 8 + t];
#pragma unroll
            for (int c = 0; c < 3; ++c) co.st_pts[row * 3 + c] = stage_pos[(wave * 64 + lane) * 4 + c];
        }
        if (lane == 0) {
            co.ray_cnt[ray] = cnt;
            co.ray_nsel[ray] = nsel;
            co.ray_bits[ray] = valid_bits;
        }
    } else if (COMPACT) {
        const int cnt = __popcll(valid_bits);
        int base = 0;
        if (cnt > 0) {
            if (lane == 0) base = atomicAdd(co.counter, cnt);
            base = __builtin_amdgcn_readfirstlane(base);
            // Lists too small: raise the flag (the host retries with larger lists and discards this attempt).  Every row below
            // the capacity must still hold DEFINED data, because the shading kernels run on rows 0 .. min(count, capacity) - 1
            // before the host sees the flag: the one ray that straddles the end of the lists marks its rows as neighbour-less
            // (-1) instead of leaving them uninitialised (a garbage index would be a wild gather).
            const bool over = base + cnt > co.capacity;
            if (over && lane == 0) atomicOr(co.counter + 1, 1);
            if (lane < nsel && ((valid_bits >> lane) & 1ull)) {
                const int row = base + __popcll(valid_bits & ((1ull << lane) - 1ull));
                if (row < co.capacity) {
                    for (int t = 0; t < a.k; ++t) co.nb[(int64_t)row * a.k + t] = over ? -1 : stage_idx[(wave * 64 + lane) * 8 + t];
#pragma unroll
                    for (int c = 0; c < 3; ++c) co.pts[(int64_t)row * 3 + c] = stage_pos[(wave * 64 + lane) * 4 + c];
                }
            }
        }
        if (lane == 0) {
            co.ray_base[ray] = base;
            co.ray_nsel[ray] = nsel;
            co.ray_bits[ray] = valid_bits;
        }
    } else {
        for (int slot = nsel + lane; slot < a.M; slot += 64) {
            for (int t = 0; t < a.k; ++t) out_idx[slot * a.k + t] = -1;
            out_loc[slot * 3 + 0] = 0.f; out_loc[slot * 3 + 1] = 0.f; out_loc[slot * 3 + 2] = 0.f;
            out_ss[slot] = -1;
        }
        if (lane == 0) a.nsel[ray] = nsel;
    }
    }   // rays of this wave
}

// Ordered compaction (second launch of npcd_grid_query_compact_ordered).  Workgroup j owns the 64 rays [64 j, 64 j + 64).  Its
// base is the sum of the counts of all rays before it, which it adds up itself (integers: no order dependence; no flags / spinning
// between workgroups, no atomics, nothing to zero beforehand).  A wave scan over the own 64 counts gives every ray its base; then one thread per OUTPUT row finds its ray (the
// last one whose base is <= the row: binary search over the 64 bases in LDS) and moves the row -- all loads of a thread are
// independent of other rows.  Rows past `capacity` are not written and raise the overflow flag (the host retries with larger
// lists), the total goes to counter[0].
constexpr int kOrdRays = 64;
// Above kOrdTwoLevel rays per call a first pass adds the counts up per block of kOrdBlock rays (one workgroup each, no atomics, nothing
// to zero) and the compaction workgroups sum block sums + the counts of their own block only: without it every workgroup re-reads all
// counts before its rays -- quadratic in the rays, 0.5 GB of L2 reads for the 131,072 rays of an 8-view evaluation batch.
// Integer sums: the bases, and with them the lists, are the same bits either way.
constexpr int kOrdBlock = 1024, kOrdTwoLevel = 32768;
__global__ __launch_bounds__(256) void ray_block_sum_kernel(const int32_t* ray_cnt, int32_t* blk_sum, int nrays) {
    __shared__ int red[4];
    const int tid = threadIdx.x, i = blockIdx.x * kOrdBlock + tid * 4;
    int acc = 0;
    if (i < nrays) {        // (the counts are padded to 16 bytes; rays past the end of the last group of four were never written)
        const int4 c = *reinterpret_cast<const int4*>(ray_cnt + i);
        acc = (c.x + (i + 1 < nrays ? c.y : 0)) + ((i + 2 < nrays ? c.z : 0) + (i + 3 < nrays ? c.w : 0));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) blk_sum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void compact_ordered_kernel(CompactOut co, int nrays, int M, int k, const int32_t* blk_sum) {
    __shared__ int red[4];
    __shared__ int scan[kOrdRays + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int first = blockIdx.x * kOrdRays;
    // (round 4: the counts themselves, 16 bytes per thread and trip -- at most 64 KB from L2 for the last workgroup of a 128 x 128
    // view -- instead of per-group sums that the query kernel accumulated with atomics into a buffer a fill launch had to zero)
    int acc = 0;
    int from = 0;
    if (blk_sum) {          // (kernel-uniform) whole blocks before this workgroup's rays come from the first pass
        from = first / kOrdBlock * kOrdBlock;
        for (int b = tid; b < first / kOrdBlock; b += 256) acc += blk_sum[b];
    }
    for (int i = from + tid * 4; i < first; i += 1024) {
        const int4 c = *reinterpret_cast<const int4*>(co.ray_cnt + i);
        acc += (c.x + c.y) + (c.z + c.w);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    const int base0 = (red[0] + red[1]) + (red[2] + red[3]);
    if (wave == 0) {
        const int ray = first + lane;
        const int cnt = ray < nrays ? co.ray_cnt[ray] : 0;
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o, 64);
            if (lane >= o) inc += v;
        }
        const int base = base0 + inc - cnt;
        scan[lane] = base;
        if (lane == 63) scan[64] = base0 + inc;
        if (ray < nrays) co.ray_base[ray] = base;
        if (co.lim_part) {          // (kernel-uniform) limits of this workgroup's rays that hit the cube
            uint32_t kmin = 0xffffffffu, kmax = 0u;
            if (ray < nrays) {
                const float a0 = co.lim_t0[ray], a1 = co.lim_t1[ray];
                if (a1 > a0) { kmin = fkey(a0); kmax = fkey(a1); }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
                kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
            }
            if (lane == 0) { co.lim_part[2 * blockIdx.x] = kmin; co.lim_part[2 * blockIdx.x + 1] = kmax; }
        }
        if (ray == nrays - 1) {
            co.counter[0] = base + cnt;
            co.counter[1] = (base + cnt > co.capacity) ? 1 : 0;
            co.counter[2] = 0;          // the call's status word for the shading kernels (npcd_shade_points `status`), [3] reserved
            co.counter[3] = 0;
        }
    }
    __syncthreads();
    const int end = min(scan[64], co.capacity);
    const int nloc = min(kOrdRays, nrays - first);
    for (int row = base0 + tid; row < end; row += 256) {
        int lo = 0, hi = nloc - 1;                       // largest r with scan[r] <= row (an empty ray shares its base with its
        while (lo < hi) {                                // successor: the LAST of equal bases is the one that owns the row)
            const int mid = (lo + hi + 1) >> 1;
            if (scan[mid] <= row) lo = mid; else hi = mid - 1;
        }
        const int64_t src = (int64_t)(first + lo) * M + (row - scan[lo]);
        if (k == 8) {
            const int4* sp = reinterpret_cast<const int4*>(co.st_nb + src * 8);
            int4* dp = reinterpret_cast<int4*>(co.nb + (int64_t)row * 8);
            const int4 a = sp[0], b = sp[1];
            dp[0] = a;
            dp[1] = b;
        } else {
            for (int t = 0; t < k; ++t) co.nb[(int64_t)row * k + t] = co.st_nb[src * k + t];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) co.pts[(int64_t)row * 3 + c] = co.st_pts[src * 3 + c];
    }
}

// ============================================================================================
// ray march
// ============================================================================================
__global__ void march_init_kernel(uint32_t* ws) {
    ws[0] = 0xffffffffu;  // min depth key
    ws[1] = 0u;           // max depth key
}

// COMPACT: slot validity comes from a per-ray 64-bit mask and slot positions from the compact point list
template <bool COMPACT>
__global__ __launch_bounds__(256) void ray_march_kernel(const float* __restrict__ sigma, const float* __restrict__ rgb, const uint8_t* __restrict__ slot_valid,
                                                        const float* __restrict__ slot_loc, const int32_t* __restrict__ point_base,
                                                        const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ t1,
                                                        int Nr, int M, int capacity, int white_back, float* __restrict__ mask, float* __restrict__ depth,
                                                        float* __restrict__ channels, uint32_t* ws) {
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    if (ray < Nr) {
        const float o[3] = {rays_o[ray * 3], rays_o[ray * 3 + 1], rays_o[ray * 3 + 2]};
        const float d[3] = {rays_d[ray * 3], rays_d[ray * 3 + 1], rays_d[ray * 3 + 2]};
        const float ray_end = t1[ray];
        const uint8_t* sv = COMPACT ? nullptr : slot_valid + (int64_t)ray * M;
        const float* sl = COMPACT ? nullptr : slot_loc + (int64_t)ray * M * 3;
        unsigned long long bits = COMPACT ? reinterpret_cast<const unsigned long long*>(slot_valid)[ray] : 0ull;
        int cp = point_base[ray];
        // compact lists that overflowed (grid_query_wave_kernel sets the flag, the host retries with larger buffers): a ray
        // whose rows do not all exist is marched as empty instead of reading past the lists
        if (COMPACT && cp + __popcll(bits) > capacity) bits = 0ull;
        float run_max = -INFINITY;
        float T = 1.f, total = 0.f, wd = 0.f, cr = 0.f, cg = 0.f, cb = 0.f;
        float pd = 0.f, ps = 0.f, pr = 0.f, pg = 0.f, pb = 0.f;  // previous slot
        bool pv = false;
        for (int j = 0; j < M; ++j) {
            const bool valid = COMPACT ? ((bits >> j) & 1ull) != 0 : sv[j] != 0;
            float sg = 0.f, r_ = 0.f, g_ = 0.f, b_ = 0.f;
            if (valid) {
                // depth = nanmean_xyz((p - o) / d)   (renderer.py:103)
                float acc = 0.f;
                int cnt = 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float pc = COMPACT ? slot_loc[(int64_t)cp * 3 + c] : sl[j * 3 + c];
                    const float q = (pc - o[c]) / d[c];
                    if (q == q) { acc += q; ++cnt; }
                }
                const float dep = acc / (float)cnt;  // cnt == 0 -> NaN, like torch.nanmean
                run_max = fmaxf(run_max, dep);
                sg = sigma[cp]; r_ = rgb[cp * 3]; g_ = rgb[cp * 3 + 1]; b_ = rgb[cp * 3 + 2];
                ++cp;
            }
            const float dj = (run_max == -INFINITY) ? ray_end : run_max;
            kmin = min(kmin, fkey(dj));
            kmax = max(kmax, fkey(dj));
            if (j > 0) {
                const float alpha = 1.f - expf(-(ps * (dj - pd)));
                const float w = alpha * T;
                total += w;
                wd += w * pd;
                if (pv) { cr += w * pr; cg += w * pg; cb += w * pb; }
                T *= (1.f - alpha + 1e-10f);
            }
            pd = dj; ps = sg; pr = r_; pg = g_; pb = b_; pv = valid;
        }
        // the last slot has delta = 0 -> alpha = 0 -> contributes nothing (volume_renderer.py:35)
        mask[ray] = total;
        depth[ray] = wd / total;  // NaN when total == 0; fixed up by depth_clamp_kernel
        const float bg = white_back ? 1.f - total : 0.f;
        channels[ray * 3 + 0] = cr + bg;
        channels[ray * 3 + 1] = cg + bg;
        channels[ray * 3 + 2] = cb + bg;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
    }
    if ((threadIdx.x & 63) == 0 && kmax != 0u) {
        atomicMin(&ws[0], kmin);
        atomicMax(&ws[1], kmax);
    }
}

// The same march with one WAVE per ray (lane = slot, M <= 64).  The thread-per-ray form above walks its <= 64 slots one after the
// other behind dependent loads on 64 workgroups for a 128^2 view; here the running depth maximum is a wave prefix maximum, the
// transmittance an exclusive prefix product, the outputs wave sums -- all on the vector ALU by DPP (row shifts inside the 16-lane
// rows, row_bcast 15 / 31 across them, wave_shr 1 for "the previous slot"): a scan is 6 instructions, where the ds_bpermute form
// (__shfl_up) costs an LDS round trip per step (measured: 28.6 us, no better than the loop).  Sums and products are taken in scan
// / tree order instead of slot order: same values to a few fp32 ulps.  The depth limits leave the kernel as one (min, max) pair per
// workgroup behind the two result words of `ws`; depth_clamp_kernel combines them (atomics on the two global words made that cache
// line the bottleneck of the kernel: 104 us with a pair per ray, 47 / 30 / 20 us with a conditional pair per 4 / 8 / 16 rays).
#define NPCD_DPP_F(IDENT, X, CTRL, ROWMASK) \
    __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(IDENT), __float_as_int(X), CTRL, ROWMASK, 0xf, false))
template <class Op>
__device__ __forceinline__ float wave_scan_incl(float x, float ident, Op op) {
    x = op(x, NPCD_DPP_F(ident, x, 0x111 /* row_shr:1 */, 0xf));
    x = op(x, NPCD_DPP_F(ident, x, 0x112 /* row_shr:2 */, 0xf));
    x = op(x, NPCD_DPP_F(ident, x, 0x114 /* row_shr:4 */, 0xf));
    x = op(x, NPCD_DPP_F(ident, x, 0x118 /* row_shr:8 */, 0xf));
    x = op(x, NPCD_DPP_F(ident, x, 0x142 /* row_bcast:15 */, 0xa));     // rows 1 and 3 take the total of the row before them
    x = op(x, NPCD_DPP_F(ident, x, 0x143 /* row_bcast:31 */, 0xc));     // rows 2 and 3 take the total of rows 0-1
    return x;
}
__device__ __forceinline__ float wave_prev(float x, float first) {     // lane j gets lane j - 1's value, lane 0 gets `first`
    return NPCD_DPP_F(first, x, 0x138 /* wave_shr:1 */, 0xf);
}
__device__ __forceinline__ float wave_total(float x) {                  // sum over the wave, on every lane (as csrc/elementwise.hip)
    x += NPCD_DPP_F(0.f, x, 0x128 /* row_ror:8 */, 0xf);
    x += NPCD_DPP_F(0.f, x, 0x124, 0xf);
    x += NPCD_DPP_F(0.f, x, 0x122, 0xf);
    x += NPCD_DPP_F(0.f, x, 0x121, 0xf);
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
#ifndef NPCD_MARCH_RAYS
#define NPCD_MARCH_RAYS 1
#endif
constexpr int kMarchRays = NPCD_MARCH_RAYS;
// the fused render's extras of the march (null / 0 otherwise): the start of every ray + the per-workgroup limit pairs of the compaction
// kernel (a ray that misses the cube ends at the global end)
struct MarchFused {
    const float* t0;
    const uint32_t* lim_part;
    int lim_n;
};
template <bool COMPACT>
__global__ __launch_bounds__(256) void ray_march_wave_kernel(const float* __restrict__ sigma, const float* __restrict__ rgb,
                                                             const uint8_t* __restrict__ slot_valid, const float* __restrict__ slot_loc,
                                                             const int32_t* __restrict__ point_base, const float* __restrict__ rays_o,
                                                             const float* __restrict__ rays_d, const float* __restrict__ t1, int Nr, int M,
                                                             int capacity, int white_back, float* __restrict__ mask, float* __restrict__ depth,
                                                             float* __restrict__ channels, uint32_t* ws, MarchFused mf) {
    __shared__ uint32_t wg_min[4], wg_max[4];
    __shared__ uint32_t lim_hi[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool in = lane < M;
    // fused render: the global end of the rays that hit the cube, from the pairs compact_ordered_kernel left (a few hundred words)
    float global_end = 0.f;
    bool any_hit = false;
    if (mf.lim_part) {
        uint32_t kmax = 0u;
        for (int i = threadIdx.x; i < mf.lim_n; i += 256) kmax = max(kmax, mf.lim_part[2 * i + 1]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
        if (lane == 0) lim_hi[wave] = kmax;
        __syncthreads();
        kmax = max(max(lim_hi[0], lim_hi[1]), max(lim_hi[2], lim_hi[3]));
        any_hit = kmax != 0u;
        global_end = fkey_inv(kmax);
    }
    float dmin = INFINITY, dmax = -INFINITY;           // this lane's slot depths over the wave's rays
    bool any = false;
#pragma unroll 1
    for (int it = 0; it < kMarchRays; ++it) {
        const int ray = (blockIdx.x * 4 + wave) * kMarchRays + it;      // wave-uniform
        if (ray >= Nr) break;
        const float o[3] = {rays_o[ray * 3], rays_o[ray * 3 + 1], rays_o[ray * 3 + 2]};
        const float d[3] = {rays_d[ray * 3], rays_d[ray * 3 + 1], rays_d[ray * 3 + 2]};
        float ray_end = t1[ray];
        if (mf.lim_part && any_hit && !(ray_end > mf.t0[ray])) ray_end = global_end;      // a ray that misses the cube (renderer.py:40-43)
        unsigned long long bits;
        if (COMPACT) bits = reinterpret_cast<const unsigned long long*>(slot_valid)[ray];
        else bits = __ballot(in && slot_valid[(int64_t)ray * M + lane] != 0);
        const int cp0 = point_base[ray];
        if (COMPACT && cp0 + __popcll(bits) > capacity) bits = 0ull;      // overflowed lists: the ray is marched as empty (see above)
        const bool valid = in && ((bits >> lane) & 1ull) != 0;
        const int cp = cp0 + __popcll(bits & ((1ull << lane) - 1ull));
        float dep = -INFINITY, sg = 0.f, r_ = 0.f, g_ = 0.f, b_ = 0.f;
        if (valid) {
            const float* pp = COMPACT ? slot_loc + (int64_t)cp * 3 : slot_loc + ((int64_t)ray * M + lane) * 3;
            float acc = 0.f;
            int cnt = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float q = (pp[c] - o[c]) / d[c];
                if (q == q) { acc += q; ++cnt; }
            }
            const float dv = acc / (float)cnt;       // cnt == 0 -> NaN, which the running maximum ignores (fmaxf)
            dep = dv == dv ? dv : -INFINITY;
            sg = sigma[cp]; r_ = rgb[cp * 3]; g_ = rgb[cp * 3 + 1]; b_ = rgb[cp * 3 + 2];
        }
        const float run = wave_scan_incl(dep, -INFINITY, [](float a, float b2) { return fmaxf(a, b2); });
        const float dj = (run == -INFINITY) ? ray_end : run;
        const float pd = wave_prev(dj, 0.f), ps = wave_prev(sg, 0.f), pr = wave_prev(r_, 0.f), pg = wave_prev(g_, 0.f), pb = wave_prev(b_, 0.f);
        const bool pv = wave_prev(valid ? 1.f : 0.f, 0.f) != 0.f;
        const bool step = in && lane >= 1;            // slot j >= 1 closes the interval that slot j - 1 opened
        const float alpha = step ? 1.f - expf(-(ps * (dj - pd))) : 0.f;
        const float prod = wave_scan_incl(step ? (1.f - alpha + 1e-10f) : 1.f, 1.f, [](float a, float b2) { return a * b2; });
        const float w = alpha * wave_prev(prod, 1.f);
        const float total = wave_total(w), wd = wave_total(w * pd);
        const float cr = wave_total(pv ? w * pr : 0.f), cg = wave_total(pv ? w * pg : 0.f), cb = wave_total(pv ? w * pb : 0.f);
        if (in) { dmin = fminf(dmin, dj); dmax = fmaxf(dmax, dj); any = true; }
        if (lane == 0) {
            mask[ray] = total;
            depth[ray] = wd / total;                  // NaN when total == 0; fixed up by depth_clamp_kernel
            const float bg = white_back ? 1.f - total : 0.f;
            channels[ray * 3 + 0] = cr + bg;
            channels[ray * 3 + 1] = cg + bg;
            channels[ray * 3 + 2] = cb + bg;
        }
    }
    uint32_t kmin = any ? fkey(dmin) : 0xffffffffu, kmax = any ? fkey(dmax) : 0u;      // fkey is monotone
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
    }
    if (lane == 0) { wg_min[wave] = kmin; wg_max[wave] = kmax; }
    __syncthreads();
    if (threadIdx.x == 0) {           // this workgroup's depth limits: depth_clamp_kernel combines them (no atomics, no initialisation)
        ws[2 + 2 * blockIdx.x] = min(min(wg_min[0], wg_min[1]), min(wg_min[2], wg_min[3]));
        ws[3 + 2 * blockIdx.x] = max(max(wg_max[0], wg_max[1]), max(wg_max[2], wg_max[3]));
    }
    // (Round 6 tried the depth clamp HERE -- the last workgroup to finish, found through a ticket counter behind a device-scope fence,
    //  combining the limits and clamping all depths -- to save depth_clamp_kernel's launch: a view went from 0.354 to 0.626 ms.  The march
    //  is 4,096 workgroups of four rays; a device-scope release fence on gfx950 writes the XCD's L2 back, once per workgroup.  Removed:
    //  docs/experiments.md R6.6.)
}

// Backward of the ray march w.r.t. the compact densities and colours (stage-1 training; slot positions, ray geometry and hence
// all depths are constants there).  One thread per ray, three passes over its <= 64 slots: slot depths; alpha_i, T_i, w_i
// (kept in thread-private arrays); then in reverse, with G_i = dL/dw_i and the suffix sum S_i = sum_{j>i} G_j w_j,
//     dL/dalpha_i = G_i T_i - S_i / (1 - alpha_i + 1e-10),   dsigma_i = dL/dalpha_i * delta_i * exp(-sigma_i delta_i),
//     drgb_i = g_chan * w_i,    G_i = g_mask + g_depth (d_i - depth) / mask [only where the output depth is the unclamped,
//     finite ratio] + sum_c g_chan_c (rgb_ic [valid] - white_back).
// A ray without any weight (mask == 0) gets no depth gradient (torch's autograd gives NaN there; the reference's loss does
// not use the depth).
__global__ __launch_bounds__(64) void ray_march_bwd_kernel(const float* __restrict__ sigma, const float* __restrict__ rgb,
                                                           const uint8_t* __restrict__ slot_valid, const float* __restrict__ slot_loc,
                                                           const int32_t* __restrict__ point_base, const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d, const float* __restrict__ t1, int Nr, int M,
                                                           int white_back, const uint32_t* __restrict__ ws, const float* __restrict__ g_mask,
                                                           const float* __restrict__ g_depth, const float* __restrict__ g_chan,
                                                           float* __restrict__ dsigma, float* __restrict__ drgb) {
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (ray >= Nr) return;
    const float o[3] = {rays_o[ray * 3], rays_o[ray * 3 + 1], rays_o[ray * 3 + 2]};
    const float d[3] = {rays_d[ray * 3], rays_d[ray * 3 + 1], rays_d[ray * 3 + 2]};
    const float ray_end = t1[ray];
    const uint8_t* sv = slot_valid + (int64_t)ray * M;
    const float* sl = slot_loc + (int64_t)ray * M * 3;
    const int cp0 = point_base[ray];
    float dep[64], al[64], Tt[64];
    // pass 1: per-slot depths (renderer.py:96-110)
    float run_max = -INFINITY;
    for (int j = 0; j < M; ++j) {
        if (sv[j]) {
            float acc = 0.f;
            int cnt = 0;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float q = (sl[j * 3 + c] - o[c]) / d[c];
                if (q == q) { acc += q; ++cnt; }
            }
            run_max = fmaxf(run_max, acc / (float)cnt);
        }
        dep[j] = (run_max == -INFINITY) ? ray_end : run_max;
    }
    // pass 2: alpha, transmittance, the two sums of the forward
    float T = 1.f, total = 0.f, wd = 0.f;
    int cp = cp0;
    for (int j = 0; j < M; ++j) {
        const float sg = sv[j] ? sigma[cp] : 0.f;
        cp += sv[j] ? 1 : 0;
        const float delta = j + 1 < M ? dep[j + 1] - dep[j] : 0.f;
        const float alpha = 1.f - expf(-(sg * delta));
        al[j] = alpha;
        Tt[j] = T;
        total += alpha * T;
        wd += alpha * T * dep[j];
        T *= (1.f - alpha + 1e-10f);
    }
    const float lo = fkey_inv(ws[0]), hi = fkey_inv(ws[1]);
    const float draw = wd / total;
    const bool depth_live = (total != 0.f) && (draw == draw) && draw >= lo && draw <= hi && fabsf(draw) != INFINITY;
    const float gm = g_mask[ray], gd = depth_live ? g_depth[ray] / total : 0.f;
    const float gc[3] = {g_chan[ray * 3], g_chan[ray * 3 + 1], g_chan[ray * 3 + 2]};
    const float gbg = white_back ? gc[0] + gc[1] + gc[2] : 0.f;
    // pass 3: reverse
    float S = 0.f;
    for (int j = M - 1; j >= 0; --j) {
        const bool valid = sv[j] != 0;
        cp -= valid ? 1 : 0;                                     // compact index of slot j
        const float w = al[j] * Tt[j];
        float G = gm - gbg;
        if (depth_live) G += gd * (dep[j] - draw);               // (not a multiplication by 0: draw is NaN on an empty ray)
        if (valid) {
            const float r_ = rgb[cp * 3], g_ = rgb[cp * 3 + 1], b_ = rgb[cp * 3 + 2];
            G += gc[0] * r_ + gc[1] * g_ + gc[2] * b_;
            const float dalpha = G * Tt[j] - S / (1.f - al[j] + 1e-10f);
            const float delta = j + 1 < M ? dep[j + 1] - dep[j] : 0.f;
            dsigma[cp] = dalpha * delta * (1.f - al[j]);         // exp(-sigma delta) = 1 - alpha
            drgb[cp * 3 + 0] = gc[0] * w;
            drgb[cp * 3 + 1] = gc[1] * w;
            drgb[cp * 3 + 2] = gc[2] * w;
        }
        S += G * w;
    }
}

// nan -> +inf -> clamp to the global [min, max] of the per-slot depths (renderer.py:151-156)
// nparts > 0: ws[2 + 2 i], ws[3 + 2 i] hold the limits of march workgroup i; every workgroup here combines them (a few thousand
// words from L2) and workgroup 0 leaves the result in ws[0], ws[1] for the backward
__global__ __launch_bounds__(256) void depth_clamp_kernel(int Nr, float* __restrict__ depth, uint32_t* ws, int nparts) {
    __shared__ uint32_t red_lo[4], red_hi[4];
    const int ray = blockIdx.x * blockDim.x + threadIdx.x;
    if (nparts > 0) {
        uint32_t kmin = 0xffffffffu, kmax = 0u;
        for (int i = threadIdx.x; i < nparts; i += 256) {
            kmin = min(kmin, ws[2 + 2 * i]);
            kmax = max(kmax, ws[3 + 2 * i]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off, 64));
            kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, off, 64));
        }
        if ((threadIdx.x & 63) == 0) { red_lo[threadIdx.x >> 6] = kmin; red_hi[threadIdx.x >> 6] = kmax; }
        __syncthreads();
        kmin = min(min(red_lo[0], red_lo[1]), min(red_lo[2], red_lo[3]));
        kmax = max(max(red_hi[0], red_hi[1]), max(red_hi[2], red_hi[3]));
        if (blockIdx.x == 0 && threadIdx.x == 0) { ws[0] = kmin; ws[1] = kmax; }
        if (ray >= Nr) return;
        float x = depth[ray];
        if (x != x) x = INFINITY;
        depth[ray] = fminf(fmaxf(x, fkey_inv(kmin)), fkey_inv(kmax));
        return;
    }
    if (ray >= Nr) return;
    const float lo = fkey_inv(ws[0]), hi = fkey_inv(ws[1]);
    float x = depth[ray];
    if (x != x) x = INFINITY;
    depth[ray] = fminf(fmaxf(x, lo), hi);
}

}  // namespace npcd

using namespace npcd;

// --------------------------------------------------------------------------------------------
static int ray_gen_launch(const float* extr, const float* intr, int V, int res, float box, const int32_t* pixel_ids, int n_ids, float* rays_o,
                          float* rays_d, float* t0, float* t1, float* limits_ws, void* stream) {
    if (!extr || !intr || !rays_o || !rays_d || !t0 || !t1 || !limits_ws || V <= 0 || res <= 0) return NPCD_ERR_ARG;
    if (pixel_ids && n_ids <= 0) return NPCD_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)V * (pixel_ids ? (int64_t)n_ids : (int64_t)res * res);
    const int grid = (int)((n + 255) / 256);
    uint32_t* ws = reinterpret_cast<uint32_t*>(limits_ws);
    hipLaunchKernelGGL(ray_gen_kernel, dim3(grid), dim3(256), 0, st, extr, intr, V, res, box, pixel_ids, n_ids, rays_o, rays_d, t0, t1, ws);
    hipLaunchKernelGGL(ray_limits_fix_kernel, dim3(grid), dim3(256), 0, st, n, t0, t1, ws);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int64_t npcd_ray_gen_ws_floats(int V, int res, int n_ids) {
    if (V <= 0 || res <= 0 || n_ids < 0) return -1;
    const int64_t n = (int64_t)V * (n_ids ? (int64_t)n_ids : (int64_t)res * res);
    return 4 + 2 * ((n + 255) / 256);
}

extern "C" int npcd_ray_gen(const float* extr, const float* intr, int V, int res, float box, float* rays_o, float* rays_d,
                            float* t0, float* t1, float* limits_ws, void* stream) {
    return ray_gen_launch(extr, intr, V, res, box, nullptr, 0, rays_o, rays_d, t0, t1, limits_ws, stream);
}

extern "C" int npcd_ray_gen_subset(const float* extr, const float* intr, int V, int res, float box, const int32_t* pixel_ids, int n_ids,
                                   float* rays_o, float* rays_d, float* t0, float* t1, float* limits_ws, void* stream) {
    if (!pixel_ids) return NPCD_ERR_ARG;
    return ray_gen_launch(extr, intr, V, res, box, pixel_ids, n_ids, rays_o, rays_d, t0, t1, limits_ws, stream);
}

// The grid the kernels work on.  NPCD_GRID_SCALED is the fine-grid machinery run on ONE grid of edge voxel_size * voxel_scale
// (one fp32 product per axis) with the coarse dimensions and voxel_scale 1: lists, caps, occupancy and the candidate window all
// live there.  The query radius always comes from the caller's UNSCALED voxel_size (grid_radius).
static inline npcd_grid_params effective_grid(const npcd_grid_params& g) {
    if (g.grid_level != NPCD_GRID_SCALED) return g;
    npcd_grid_params e = g;
    for (int a = 0; a < 3; ++a) {
        e.voxel_size[a] = g.voxel_size[a] * (float)g.voxel_scale[a];
        e.voxel_scale[a] = 1;
        e.dims[a] = g.cdims[a];
    }
    e.grid_level = NPCD_GRID_FINE;
    return e;
}
static inline float grid_radius(const npcd_grid_params& g, float r) {
    float vmax = g.voxel_size[0];
    if (g.voxel_size[1] > vmax) vmax = g.voxel_size[1];
    if (g.voxel_size[2] > vmax) vmax = g.voxel_size[2];
    return (float)((double)r * (double)vmax);   // aggregator.py:20, evaluated in float64 then rounded
}

static int grid_check(const npcd_grid_params* g, int B, int N) {
    if (!g || B <= 0 || N <= 0) return NPCD_ERR_ARG;
    if (g->grid_level != NPCD_GRID_FINE && g->grid_level != NPCD_GRID_SCALED) return NPCD_ERR_ARG;
    for (int a = 0; a < 3; ++a) {
        if (g->dims[a] <= 0 || g->dims[a] > 1023 || g->cdims[a] <= 0) return NPCD_ERR_UNSUPPORTED;
        if (g->voxel_scale[a] <= 0 || g->kernel_size[a] <= 0 || (g->kernel_size[a] & 1) == 0) return NPCD_ERR_ARG;
        if (!(g->voxel_size[a] > 0.f)) return NPCD_ERR_ARG;
    }
    if (N > 4096) return NPCD_ERR_UNSUPPORTED;                     // point set must fit LDS
    if (occ_words(*g) * 4 > 65536) return NPCD_ERR_UNSUPPORTED;    // coarse bitmap must fit LDS
    return NPCD_OK;
}

static inline int64_t table_offset(const npcd_grid_params& g, int B, int N) {   // 16-byte aligned
    return (((int64_t)B * N * 4 + (int64_t)B * occ_words(g) * 4) + 15) / 16 * 16;
}

extern "C" int64_t npcd_grid_workspace_bytes(const npcd_grid_params* g_in, int B, int N) {
    if (grid_check(g_in, B, N) != NPCD_OK) return -1;
    const npcd_grid_params ge = effective_grid(*g_in);
    const npcd_grid_params* g = &ge;
    if (grid_check(g, B, N) != NPCD_OK) return -1;
    return table_offset(*g, B, N) + (table_ok(*g) ? (int64_t)B * table_voxels(*g) * 8 : 0);
}

extern "C" int npcd_grid_build(const npcd_grid_params* g_in, const float* points, const int32_t* counts, int B, int N,
                               void* workspace, void* stream) {
    int rc = grid_check(g_in, B, N);
    if (rc != NPCD_OK) return rc;
    const npcd_grid_params ge = effective_grid(*g_in);
    const npcd_grid_params* g = &ge;
    rc = grid_check(g, B, N);
    if (rc != NPCD_OK) return rc;
    if (!points || !workspace) return NPCD_ERR_ARG;
    const int nwords = occ_words(*g);
    int32_t* pcoord = static_cast<int32_t*>(workspace);
    uint32_t* occ = reinterpret_cast<uint32_t*>(pcoord + (int64_t)B * N);
    const size_t lds = (size_t)N * 8 + (size_t)nwords * 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static DynLds lds_attr;
    if (lds > 65536) NPCD_HIP_CHECK(lds_attr.ensure(reinterpret_cast<const void*>(grid_build_kernel), lds));
    int16_t* table = nullptr;
    if (table_ok(*g)) {
        table = reinterpret_cast<int16_t*>(static_cast<unsigned char*>(workspace) + table_offset(*g, B, N));
        NPCD_HIP_CHECK(hipMemsetAsync(table, 0xff, (size_t)B * table_voxels(*g) * 8, st));
    }
    hipLaunchKernelGGL(grid_build_kernel, dim3(B), dim3(256), lds, st, *g, points, counts, N, nwords, pcoord, occ, table);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_grid_query(const npcd_grid_params* g_in, const void* workspace, const float* points, int B, int N, int R, int S,
                               int M, int k, float r, int mode, const float* x, const float* rays_o, const float* rays_d,
                               const float* t0, const float* t1, int32_t* sample_idx, float* sample_loc, int32_t* slot_sample,
                               int32_t* nsel, void* stream) {
    int rc = grid_check(g_in, B, N);
    if (rc != NPCD_OK) return rc;
    const npcd_grid_params ge = effective_grid(*g_in);
    const npcd_grid_params* g = &ge;
    rc = grid_check(g, B, N);
    if (rc != NPCD_OK) return rc;
    if (!points || !sample_idx || !sample_loc || !slot_sample || !nsel) return NPCD_ERR_ARG;
    // explicit positions may come one per "ray" (the TV loss queries every point's own neighbourhood); depth sampling along
    // rays needs at least two samples
    if (R <= 0 || S <= 0 || (!x && S <= 1) || M <= 0 || k <= 0 || !(r > 0.f)) return NPCD_ERR_ARG;
    if (M > 64 || k > 8) return NPCD_ERR_UNSUPPORTED;
    if (!x && (!rays_o || !rays_d || !t0 || !t1)) return NPCD_ERR_ARG;
    if (mode == 0 && !workspace) return NPCD_ERR_ARG;
    if (mode != 0 && mode != 1) return NPCD_ERR_ARG;
    QueryArgs a{};
    a.g = *g;
    a.pcoord = static_cast<const int32_t*>(workspace);
    a.nwords = occ_words(*g);
    a.occ = reinterpret_cast<const uint32_t*>(a.pcoord + (int64_t)B * N);
    a.table = (mode == 0 && workspace && table_ok(*g))
                  ? reinterpret_cast<const int16_t*>(static_cast<const unsigned char*>(workspace) + table_offset(*g, B, N)) : nullptr;
    a.points = points;
    a.B = B; a.N = N; a.R = R; a.S = S; a.M = M; a.k = k;
    const float radius = mode == 0 ? grid_radius(*g_in, r) : r;
    a.r2 = radius * radius;
    a.x = x; a.rays_o = rays_o; a.rays_d = rays_d; a.t0 = t0; a.t1 = t1;
    a.sample_idx = sample_idx; a.sample_loc = sample_loc; a.slot_sample = slot_sample; a.nsel = nsel;
    const int bpe = (R + 3) / 4;
    // grid_query_kernel (mode 1): the cloud, the bitmap words and [4][64] selected sample ids -- nothing else
    const size_t lds = (size_t)N * 16 + (size_t)a.nwords * 4 + 4 * 64 * 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static DynLds lds_attr;
    if (mode == 1 && lds > 65536) NPCD_HIP_CHECK(lds_attr.ensure(reinterpret_cast<const void*>(grid_query_kernel<false>), lds));
    if (mode == 0) {
        // grid_query_wave_kernel: + the selected samples' geometry, the packed candidates and the per-ray staging rows
        const size_t lds2 = lds + 16 + 4 * 64 * 16 + 4 * 64 * 8 + 4 * 64 * 8 * 4 + 4 * 64 * 4 * 4;
        static DynLds lds2_attr;
        if (lds2 > 65536) NPCD_HIP_CHECK(lds2_attr.ensure(reinterpret_cast<const void*>(grid_query_wave_kernel<false>), lds2));
        hipLaunchKernelGGL(grid_query_wave_kernel<false>, dim3(B * bpe), dim3(256), lds2, st, a, CompactOut{}, bpe);
    }
    else hipLaunchKernelGGL(grid_query_kernel<false>, dim3(B * bpe), dim3(256), lds, st, a, bpe);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

static int march_parts(int Nr) { return (Nr + 4 * kMarchRays - 1) / (4 * kMarchRays); }
extern "C" int64_t npcd_ray_march_ws_floats(int Nr) { return Nr <= 0 ? -1 : 2 + 2 * (int64_t)march_parts(Nr); }

extern "C" int npcd_ray_march(const float* sigma, const float* rgb, const uint8_t* slot_valid, const float* slot_loc,
                              const int32_t* point_base, const float* rays_o, const float* rays_d, const float* t1, int Nr, int M,
                              int white_back, float* mask, float* depth, float* channels, float* depth_ws, void* stream) {
    if (!sigma || !rgb || !slot_valid || !slot_loc || !point_base || !rays_o || !rays_d || !t1 || !mask || !depth || !channels || !depth_ws)
        return NPCD_ERR_ARG;
    if (Nr <= 0 || M <= 0) return NPCD_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t* ws = reinterpret_cast<uint32_t*>(depth_ws);
    const int grid = (Nr + 255) / 256;
    int nparts = 0;
    if (M <= 64 && !getenv("NPCD_MARCH_PER_THREAD")) {        // (A/B switch: the thread-per-ray form)
        nparts = march_parts(Nr);
        hipLaunchKernelGGL(ray_march_wave_kernel<false>, dim3(nparts), dim3(256), 0, st, sigma, rgb, slot_valid, slot_loc, point_base, rays_o, rays_d,
                           t1, Nr, M, 0, white_back, mask, depth, channels, ws, MarchFused{});
    } else {
        hipLaunchKernelGGL(march_init_kernel, dim3(1), dim3(1), 0, st, ws);
        hipLaunchKernelGGL(ray_march_kernel<false>, dim3(grid), dim3(256), 0, st, sigma, rgb, slot_valid, slot_loc, point_base, rays_o, rays_d, t1,
                           Nr, M, 0, white_back, mask, depth, channels, ws);
    }
    hipLaunchKernelGGL(depth_clamp_kernel, dim3(grid), dim3(256), 0, st, Nr, depth, ws, nparts);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_ray_march_bwd(const float* sigma, const float* rgb, const uint8_t* slot_valid, const float* slot_loc,
                                  const int32_t* point_base, const float* rays_o, const float* rays_d, const float* t1, int Nr, int M,
                                  int white_back, const float* depth_ws, const float* g_mask, const float* g_depth, const float* g_chan,
                                  float* dsigma, float* drgb, void* stream) {
    if (!sigma || !rgb || !slot_valid || !slot_loc || !point_base || !rays_o || !rays_d || !t1 || !depth_ws || !g_mask || !g_depth || !g_chan ||
        !dsigma || !drgb)
        return NPCD_ERR_ARG;
    if (Nr <= 0 || M <= 0) return NPCD_ERR_ARG;
    if (M > 64) return NPCD_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(ray_march_bwd_kernel, dim3((Nr + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), sigma, rgb, slot_valid, slot_loc,
                       point_base, rays_o, rays_d, t1, Nr, M, white_back, reinterpret_cast<const uint32_t*>(depth_ws), g_mask, g_depth, g_chan,
                       dsigma, drgb);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

// Fused-render form of the neighbour query: compact shading-point lists instead of the dense [ray, slot]
// arrays.  counter [4]: counter[0] receives the number of compact points, counter[1] an overflow flag (capacity too small;
// nothing is written for the overflowing rays), counter[2] and [3] are zeroed (ABI 8: [2] is the word the caller hands to the
// shading kernels as their range-guard `status`, zeroed here so that it costs no launch of its own).  Rows of one ray are contiguous and in slot order.
struct RenderRays {          // fused ray generation of npcd_render_rays_query
    const float *extr, *intr;
    int views, res;
    float box;
    uint32_t* lim_part;
};
static int grid_query_compact_launch(const npcd_grid_params* g_in, const void* workspace, const float* points, int B, int N, int R, int S,
                                     int M, int k, float r, const float* rays_o, const float* rays_d, const float* t0, const float* t1,
                                     int32_t* counter, int32_t capacity, int32_t* ray_base, int32_t* ray_nsel, uint64_t* ray_bits,
                                     int32_t* nb_idx, float* pts, void* order_ws, void* stream, const RenderRays* rr = nullptr) {
    int rc = grid_check(g_in, B, N);
    if (rc != NPCD_OK) return rc;
    const npcd_grid_params ge = effective_grid(*g_in);
    const npcd_grid_params* g = &ge;
    rc = grid_check(g, B, N);
    if (rc != NPCD_OK) return rc;
    if (!workspace || !points || !rays_o || !rays_d || !t0 || !t1 || !counter || !ray_base || !ray_nsel || !ray_bits || !nb_idx || !pts)
        return NPCD_ERR_ARG;
    if (R <= 0 || S <= 1 || M <= 0 || k <= 0 || capacity <= 0 || !(r > 0.f)) return NPCD_ERR_ARG;
    if (M > 64 || k > 8) return NPCD_ERR_UNSUPPORTED;
    QueryArgs a{};
    a.g = *g;
    a.pcoord = static_cast<const int32_t*>(workspace);
    a.nwords = occ_words(*g);
    a.occ = reinterpret_cast<const uint32_t*>(a.pcoord + (int64_t)B * N);
    a.table = table_ok(*g) ? reinterpret_cast<const int16_t*>(static_cast<const unsigned char*>(workspace) + table_offset(*g, B, N)) : nullptr;
    a.points = points;
    a.B = B; a.N = N; a.R = R; a.S = S; a.M = M; a.k = k;
    const float radius = grid_radius(*g_in, r);
    a.r2 = radius * radius;
    a.rays_o = rays_o; a.rays_d = rays_d; a.t0 = t0; a.t1 = t1;
    CompactOut co{};
    co.counter = counter; co.capacity = capacity; co.ray_base = ray_base; co.ray_nsel = ray_nsel;
    co.ray_bits = reinterpret_cast<unsigned long long*>(ray_bits); co.nb = nb_idx; co.pts = pts;
    if (rr) {       // the rays are OUTPUTS of the query kernel here
        a.extr = rr->extr; a.intr = rr->intr; a.views = rr->views; a.res = rr->res; a.box = rr->box;
        a.gen_o = const_cast<float*>(rays_o); a.gen_d = const_cast<float*>(rays_d);
        a.gen_t0 = const_cast<float*>(t0); a.gen_t1 = const_cast<float*>(t1);
        co.lim_t0 = t0; co.lim_t1 = t1; co.lim_part = rr->lim_part;
    }
    // rays per wave (NPCD_QUERY_RPW, an A/B switch; default 1).  Round 6 measured 2 / 4 / 8 rays per wave on the bench view: the kernel
    // got SLOWER -- 63.4 -> 70.4 / 103.5 / 102.7 us at 128 depth samples, 40.7 -> 40.9 / 58.5 / 58.8 at 64: the rays of a view differ by
    // an order of magnitude in cost, a wave that draws several long ones finishes last, and the 12 KB of staging per workgroup it saves
    // were never the limit (docs/experiments.md R6.7).
    static const int rpw_env = [] { const char* e = getenv("NPCD_QUERY_RPW"); return e ? atoi(e) : 0; }();
    int rpw = rpw_env > 0 ? rpw_env : 1;
    a.rpw = rpw;
    const int bpe = (R + 4 * rpw - 1) / (4 * rpw);
    const size_t lds = (size_t)N * 16 + (size_t)a.nwords * 4 + 4 * 64 * 4 + 16 + 4 * 64 * 16 + 4 * 64 * 8 + 4 * 64 * 8 * 4 + 4 * 64 * 4 * 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    static DynLds lds_attr;
    if (lds > 65536) NPCD_HIP_CHECK(lds_attr.ensure(reinterpret_cast<const void*>(grid_query_wave_kernel<true>), lds));
    if (order_ws) {
        // staging: group sums, [B R] counts (each padded to 16 bytes), [B R][M][k] indices, [B R][M][3] positions
        const int64_t nrays = (int64_t)B * R, nblk = (nrays + kOrdRays - 1) / kOrdRays;
        unsigned char* w = static_cast<unsigned char*>(order_ws);
        w += (nblk * 4 + 15) / 16 * 16;                 // (the group sums of rounds 1-3: unused, the layout of the workspace is unchanged)
        co.ray_cnt = reinterpret_cast<int32_t*>(w);
        co.st_nb = reinterpret_cast<int32_t*>(w + (nrays * 4 + 15) / 16 * 16);
        co.st_pts = reinterpret_cast<float*>(co.st_nb + nrays * M * k);
        hipLaunchKernelGGL(grid_query_wave_kernel<true>, dim3(B * bpe), dim3(256), lds, st, a, co, bpe);
        int32_t* blk_sum = nullptr;
        if (nrays > kOrdTwoLevel) {                     // (they fit into the former group-sum area: one word per 1,024 rays of nblk * 4 bytes)
            blk_sum = static_cast<int32_t*>(order_ws);
            hipLaunchKernelGGL(ray_block_sum_kernel, dim3((unsigned)((nrays + kOrdBlock - 1) / kOrdBlock)), dim3(256), 0, st, co.ray_cnt,
                               blk_sum, (int)nrays);
        }
        hipLaunchKernelGGL(compact_ordered_kernel, dim3((unsigned)nblk), dim3(256), 0, st, co, (int)nrays, M, k, blk_sum);
    } else {
        NPCD_HIP_CHECK(hipMemsetAsync(counter, 0, 4 * sizeof(int32_t), st));
        hipLaunchKernelGGL(grid_query_wave_kernel<true>, dim3(B * bpe), dim3(256), lds, st, a, co, bpe);
    }
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}

extern "C" int npcd_grid_query_compact(const npcd_grid_params* g_in, const void* workspace, const float* points, int B, int N, int R, int S,
                                       int M, int k, float r, const float* rays_o, const float* rays_d, const float* t0, const float* t1,
                                       int32_t* counter, int32_t capacity, int32_t* ray_base, int32_t* ray_nsel, uint64_t* ray_bits,
                                       int32_t* nb_idx, float* pts, void* stream) {
    return grid_query_compact_launch(g_in, workspace, points, B, N, R, S, M, k, r, rays_o, rays_d, t0, t1, counter, capacity, ray_base, ray_nsel,
                                     ray_bits, nb_idx, pts, nullptr, stream);
}
extern "C" int64_t npcd_grid_query_order_ws_bytes(int B, int R, int M, int k) {
    if (B <= 0 || R <= 0 || M <= 0 || k <= 0) return -1;
    const int64_t nrays = (int64_t)B * R;
    return ((nrays + kOrdRays - 1) / kOrdRays * 4 + 15) / 16 * 16 + (nrays * 4 + 15) / 16 * 16 + nrays * M * (int64_t)(k * 4 + 12);
}
extern "C" int npcd_grid_query_compact_ordered(const npcd_grid_params* g_in, const void* workspace, const float* points, int B, int N, int R,
                                               int S, int M, int k, float r, const float* rays_o, const float* rays_d, const float* t0,
                                               const float* t1, int32_t* counter, int32_t capacity, int32_t* ray_base, int32_t* ray_nsel,
                                               uint64_t* ray_bits, int32_t* nb_idx, float* pts, void* order_ws, void* stream) {
    if (!order_ws || (reinterpret_cast<uintptr_t>(order_ws) & 15) || (reinterpret_cast<uintptr_t>(nb_idx) & 15)) return NPCD_ERR_ARG;
    return grid_query_compact_launch(g_in, workspace, points, B, N, R, S, M, k, r, rays_o, rays_d, t0, t1, counter, capacity, ray_base, ray_nsel,
                                     ray_bits, nb_idx, pts, order_ws, stream);
}

// Ray march on the compact layout produced by npcd_grid_query_compact.
static int ray_march_compact_launch(const float* sigma, const float* rgb, const uint64_t* ray_bits, const float* pts, const int32_t* ray_base,
                                    const float* rays_o, const float* rays_d, const float* t1, int Nr, int M, int capacity, int white_back,
                                    float* mask, float* depth, float* channels, float* depth_ws, const MarchFused& mf, void* stream) {
    if (!sigma || !rgb || !ray_bits || !pts || !ray_base || !rays_o || !rays_d || !t1 || !mask || !depth || !channels || !depth_ws)
        return NPCD_ERR_ARG;
    if (Nr <= 0 || M <= 0 || M > 64 || capacity <= 0) return NPCD_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t* ws = reinterpret_cast<uint32_t*>(depth_ws);
    const int grid = (Nr + 255) / 256;
    int nparts = 0;
    const bool per_thread = getenv("NPCD_MARCH_PER_THREAD") != nullptr;
    if (per_thread && mf.lim_part) return NPCD_ERR_UNSUPPORTED;
    if (!per_thread) {
        nparts = march_parts(Nr);
        hipLaunchKernelGGL(ray_march_wave_kernel<true>, dim3(nparts), dim3(256), 0, st, sigma, rgb, reinterpret_cast<const uint8_t*>(ray_bits), pts,
                           ray_base, rays_o, rays_d, t1, Nr, M, capacity, white_back, mask, depth, channels, ws, mf);
    } else {
        hipLaunchKernelGGL(march_init_kernel, dim3(1), dim3(1), 0, st, ws);
        hipLaunchKernelGGL(ray_march_kernel<true>, dim3(grid), dim3(256), 0, st, sigma, rgb, reinterpret_cast<const uint8_t*>(ray_bits), pts, ray_base,
                           rays_o, rays_d, t1, Nr, M, capacity, white_back, mask, depth, channels, ws);
    }
    hipLaunchKernelGGL(depth_clamp_kernel, dim3(grid), dim3(256), 0, st, Nr, depth, ws, nparts);
    NPCD_HIP_CHECK(hipGetLastError());
    return NPCD_OK;
}
extern "C" int npcd_ray_march_compact(const float* sigma, const float* rgb, const uint64_t* ray_bits, const float* pts, const int32_t* ray_base,
                                      const float* rays_o, const float* rays_d, const float* t1, int Nr, int M, int capacity, int white_back,
                                      float* mask, float* depth, float* channels, float* depth_ws, void* stream) {
    return ray_march_compact_launch(sigma, rgb, ray_bits, pts, ray_base, rays_o, rays_d, t1, Nr, M, capacity, white_back, mask, depth, channels,
                                    depth_ws, MarchFused{}, stream);
}

// ---- fused render (round 6, ABI 9): ray generation inside the neighbour query, the box-limit fix-up inside the march -- two launches
// per view fewer (ray_gen_kernel, ray_limits_fix_kernel), the same bits.
// npcd_render_rays_query = npcd_ray_gen + npcd_grid_query_compact_ordered for all views * res^2 pixels of every example, except that
// t0 / t1 of a ray that misses the cube keep their raw values (-1, -2) and lim_part [npcd_render_lim_words(B, R)] receives per-group
// (min start, max end) keys; npcd_ray_march_compact_fused takes those.  Needs box >= the grid's range on every axis (a ray that misses
// the cube then has no sample in the grid).
extern "C" int64_t npcd_render_lim_words(int B, int R) {
    if (B <= 0 || R <= 0) return -1;
    return 2 * (((int64_t)B * R + kOrdRays - 1) / kOrdRays);
}
extern "C" int npcd_render_rays_query(const npcd_grid_params* g_in, const void* workspace, const float* points, int B, int N, const float* extr,
                                      const float* intr, int views, int res, float box, int S, int M, int k, float r, float* rays_o,
                                      float* rays_d, float* t0, float* t1, int32_t* counter, int32_t capacity, int32_t* ray_base,
                                      int32_t* ray_nsel, uint64_t* ray_bits, int32_t* nb_idx, float* pts, void* order_ws, uint32_t* lim_part,
                                      void* stream) {
    if (!g_in || !extr || !intr || !lim_part || views <= 0 || res <= 0) return NPCD_ERR_ARG;
    if (!order_ws || (reinterpret_cast<uintptr_t>(order_ws) & 15) || (reinterpret_cast<uintptr_t>(nb_idx) & 15)) return NPCD_ERR_ARG;
    if ((int64_t)views * res * res > (1 << 28)) return NPCD_ERR_UNSUPPORTED;
    for (int a = 0; a < 3; ++a)
        if (!(box >= g_in->range_max[a]) || !(-box <= g_in->range_min[a])) return NPCD_ERR_UNSUPPORTED;
    RenderRays rr{extr, intr, views, res, box, lim_part};
    return grid_query_compact_launch(g_in, workspace, points, B, N, views * res * res, S, M, k, r, rays_o, rays_d, t0, t1, counter, capacity,
                                     ray_base, ray_nsel, ray_bits, nb_idx, pts, order_ws, stream, &rr);
}
extern "C" int npcd_ray_march_compact_fused(const float* sigma, const float* rgb, const uint64_t* ray_bits, const float* pts,
                                            const int32_t* ray_base, const float* rays_o, const float* rays_d, const float* t0, const float* t1,
                                            const uint32_t* lim_part, int lim_pairs, int Nr, int M, int capacity,
                                            int white_back, float* mask, float* depth, float* channels, float* depth_ws, void* stream) {
    if (!t0 || !lim_part || lim_pairs <= 0) return NPCD_ERR_ARG;
    MarchFused mf{t0, lim_part, lim_pairs};
    return ray_march_compact_launch(sigma, rgb, ray_bits, pts, ray_base, rays_o, rays_d, t1, Nr, M, capacity, white_back, mask, depth, channels,
                                    depth_ws, mf, stream);
}
