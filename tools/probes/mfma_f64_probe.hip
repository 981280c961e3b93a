// Probe: v_mfma_f64_16x16x4_f64 issue rate on gfx950, alone and interleaved with FP64 VALU work.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA only, 1: VALU only, 2: both interleaved
__global__ __launch_bounds__(256) void probe(double* out, int iters, double seed) {
  double4_t acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
  double a = seed + threadIdx.x * 1e-9, b = 1.0 - a;
  double v0 = a, v1 = b, v2 = a * b, v3 = a + b, v4 = a - b, v5 = a * 3, v6 = b * 5, v7 = a * 7;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0 || MODE == 2) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
    }
    if (MODE == 1 || MODE == 2) {
      // 26 independent-ish FP64 FMAs (13 per MFMA, like the exp part of one energy step)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        v0 = fma(v0, 0.999, 1e-3); v1 = fma(v1, 0.998, 1e-3); v2 = fma(v2, 0.997, 1e-3); v3 = fma(v3, 0.996, 1e-3);
        v4 = fma(v4, 0.995, 1e-3); v5 = fma(v5, 0.994, 1e-3); v6 = fma(v6, 0.993, 1e-3); v7 = fma(v7, 0.992, 1e-3);
      }
      v0 = fma(v0, 0.991, 1e-3); v1 = fma(v1, 0.99, 1e-3);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 2048 * sizeof(double));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 2048;   // 8 blocks/CU -> 8 waves/SIMD
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 1) {
        double waves = blocks * 4.0;
        double mfma = (mode != 1) ? waves * iters * 2 : 0, valu = (mode != 0) ? waves * iters * 26 : 0;
        // cycles per SIMD at 2.4 GHz: ms * 2.4e6 ; per-SIMD instruction counts
        double cyc = ms * 2.4e6, per_simd = 1024.0;
        printf("mode %d: %.2f ms | cycles/MFMA(if only) %.1f | cycles/VALU(if only) %.2f | MFMA TFLOPs %.1f  VALU TFLOPs %.1f\n", mode, ms,
               mfma ? cyc / (mfma / per_simd) : 0.0, valu ? cyc / (valu / per_simd) : 0.0,
               mfma * 2048 / (ms * 1e-3) / 1e12, valu * 64 * 2 / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
