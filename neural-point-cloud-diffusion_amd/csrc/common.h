// Shared device/host helpers for libnpcd_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/npcd_hip.h"

namespace npcd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int kWave = 64;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// ---- element traits: the 16-bit MFMA input types -------------------------------------------
typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_raw;
typedef __attribute__((address_space(3))) fp16x4_raw lds_f16x4;

struct BF16 {
    using elem = __bf16;
    using vec8 = bf16x8;
    using vec4 = bf16x4;
    // ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block, delivered column-major
    static __device__ __forceinline__ vec4 tr_read(const unsigned char* p) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    // c + a.lo * b.lo + a.hi * b.hi on a packed pair (v_dot2c_f32_bf16)
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        typedef __attribute__((ext_vector_type(2))) __bf16 v2;
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
    }
    static __device__ __forceinline__ float lo(uint32_t a) { return __uint_as_float(a << 16); }
    static __device__ __forceinline__ float hi(uint32_t a) { return __uint_as_float(a & 0xffff0000u); }
};
struct F16 {
    using elem = _Float16;
    using vec8 = f16x8;
    using vec4 = f16x4;
    static __device__ __forceinline__ vec4 tr_read(const unsigned char* p) {
        return __builtin_bit_cast(vec4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4*)p));
    }
    static __device__ __forceinline__ f32x16 mfma32(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
        typedef __attribute__((ext_vector_type(2))) _Float16 v2;
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(v2, a), __builtin_bit_cast(v2, b), c, false);
    }
    static __device__ __forceinline__ float lo(uint32_t a) {
        typedef __attribute__((ext_vector_type(2))) _Float16 v2;
        return (float)__builtin_bit_cast(v2, a)[0];
    }
    static __device__ __forceinline__ float hi(uint32_t a) {
        typedef __attribute__((ext_vector_type(2))) _Float16 v2;
        return (float)__builtin_bit_cast(v2, a)[1];
    }
};

// Row r of a 32x32 MFMA accumulator register i on lane-half hh:  (i&3) + 8*(i>>2) + 4*hh
__device__ __forceinline__ int acc_row(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }

// Blocks b and b+8 share an XCD (round-robin dispatch, MI355X_MICROARCH.md).  Give each XCD one
// contiguous chunk of the logical grid so that neighbouring work items share an L2.  Bijective
// for any grid size.  Speed only: results never depend on the placement.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, rem = nblocks & 7, xcd = bid & 7, j = bid >> 3;
    const int base = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    return base + j;
}

// lanes l and l^32 exchange a value
__device__ __forceinline__ float swap_half(float x) { return __shfl_xor(x, 32, 64); }

// 64-wide tiles of 16-bit elements live in LDS as 128-byte rows; the 16-byte chunk index is XOR-swizzled
// with f(row) = bit-permutation of (row>>1)&7 chosen so that BOTH access patterns are conflict-free:
//   * row reads (ds_read_b128, lanes = 32 consecutive rows, same logical chunk): every 16-lane group of
//     the instruction touches 16 distinct 16-byte slots of the 256-byte bank row;
//   * transposed reads (ds_read_b64_tr_b16, a half-wave = 4 consecutive rows x 4 consecutive chunks): rows
//     q and q+2 (same bank-row half) get XOR values that differ in bit 2, i.e. disjoint chunk sets.
__device__ __forceinline__ int tile_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * 128 + (((chunk ^ tile_swz(row)) & 7) << 4); }

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// host-side error latch
void set_hip_error(hipError_t e);
#define NPCD_HIP_CHECK(expr)                          \
    do {                                              \
        hipError_t _e = (expr);                       \
        if (_e != hipSuccess) {                       \
            ::npcd::set_hip_error(_e);                \
            return NPCD_ERR_HIP;                      \
        }                                             \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: remember, per device, the largest size set
// so far.  Two host threads (autograd's backward thread and the main thread) may race here: both then set the same value.
struct DynLds {
    static constexpr int kMaxDevices = 64;
    std::atomic<size_t> set[kMaxDevices];
    DynLds() { for (auto& s : set) s.store(0); }
    hipError_t ensure(const void* kernel, size_t bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
        if (bytes <= set[dev].load(std::memory_order_acquire)) return hipSuccess;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) set[dev].store(bytes, std::memory_order_release);
        return e;
    }
};

}  // namespace npcd
