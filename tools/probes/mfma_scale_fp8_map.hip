// Dev probe: operand lane map of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit block scales.
// Hypothesis: lane l = (i = l & 31, kg = l >> 5) holds A[i][32 kg + 4 d + b] / B[32 kg + 4 d + b][i] in byte b of dword d (d = 0..7).
// Exact integer data (0..7 are exact in e4m3); prints the number of mismatching outputs under the hypothesis.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __host__ inline uint8_t e4m3_of_small_int(int v) {   // 0..8 exactly: sign 0, exp bias 7
    if (v == 0) return 0;
    int e = 0; while ((1 << (e + 1)) <= v) ++e;                 // v = 2^e * (1 + m/8)
    int m = ((v << 3) >> e) - 8;
    return (uint8_t)(((e + 7) << 3) | m);
}
__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int scale_exp_b) {
    const int l = threadIdx.x, i = l & 31, kg = l >> 5;
    v8i a, b;
    for (int d = 0; d < 8; ++d) {
        uint32_t x = 0, y = 0;
        for (int by = 0; by < 4; ++by) {
            const int k = 32 * kg + 4 * d + by;
            x |= (uint32_t)A[i * 64 + k] << (8 * by);
            y |= (uint32_t)B[k * 32 + i] << (8 * by);
        }
        a[d] = (int)x; b[d] = (int)y;
    }
    f32x16 c = {0};
    const int sa = 0x7f7f7f7f, sb = (127 + scale_exp_b) * 0x01010101;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + i] = c[r];   // row = acc_row(r, kg), col = i
}
int main() {
    uint8_t hA[32 * 64], hB[64 * 32]; int iA[32 * 64], iB[64 * 32];
    srand(1);
    for (int j = 0; j < 32 * 64; ++j) { iA[j] = rand() % 8; hA[j] = e4m3_of_small_int(iA[j]); iB[j] = rand() % 8; hB[j] = e4m3_of_small_int(iB[j]); }
    uint8_t *dA, *dB; float* dD; float hD[32 * 32];
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int se = 0; se >= -8; se -= 8) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, se);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        int bad = 0; double mx = 0;
        for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) {
            double ref = 0; for (int k = 0; k < 64; ++k) ref += (double)iA[m * 64 + k] * iB[k * 32 + n];
            ref *= (se == 0 ? 1.0 : 1.0 / 256);
            if (hD[m * 32 + n] != (float)ref) { ++bad; if (bad < 4) printf("  D[%d][%d] = %g want %g\n", m, n, hD[m * 32 + n], ref); }
            mx = ref > mx ? ref : mx;
        }
        printf("scale_b 2^%d: %d of 1024 outputs differ from the hypothesis (max |ref| %g)\n", se, bad, mx);
    }
    return 0;
}
