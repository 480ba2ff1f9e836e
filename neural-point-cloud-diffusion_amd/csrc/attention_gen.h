// Attention kernels for head dims other than 64 (csrc/attention_gen.hip); called from the npcd_attn_* entry points of attention.hip.
#pragma once
#include <stdint.h>

namespace npcd {

bool attn_gen_supported(int d);      // 32, 64, 128
// same argument meaning as npcd_attn_fwd / npcd_attn_bwd_pass (16-bit dtypes only); `passes`: bit 0 = dq (+ delta), bit 1 = dk / dv
int attn_gen_fwd(const void* q, const void* k, const void* v, void* out, float* lse, int B, int n, int H, int d, int64_t sb, int64_t sn,
                 int64_t sh, int64_t osb, int64_t osn, int64_t osh, float scale, int dtype, void* stream);
int attn_gen_bwd(int passes, const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, void* dq, void* dk,
                 void* dv, float* delta, int B, int n, int H, int d, int64_t sb, int64_t sn, int64_t sh, int64_t osb, int64_t osn, int64_t osh,
                 int64_t gsb, int64_t gsn, int64_t gsh, float scale, int dtype, void* stream);

}  // namespace npcd
