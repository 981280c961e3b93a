// The quantum-noise sample of a detector pixel (internal; shared by noise.hip - dexct_add_noise, the pass of its own - and by
// the projection kernels that draw the sample themselves, in the registers that hold the detected signal and its variance:
// rows16_kernel, the cone-beam kernels, the detection pass of the material groups).  One definition, so that a sinogram does
// not depend on which kernel sampled it.
//
// Model (noise.hip): per energy bin the detected photons are Poisson(lambda_e), each carrying the detector signal gain_e: the
// signal has mean sum_e gain_e lambda_e (the noise-free count) and variance sum_e gain_e^2 lambda_e; it is drawn as
// mean + sqrt(variance) z, z standard normal, clipped at a tiny positive number so that the log sinogram stays finite.
// RNG: Philox4x32-10, counter = (global view, row, channel, 0), key = seed.  ONE block per pixel serves all
// DEXCT_MAX_SPECTRA spectra (round 6; one block per pixel AND spectrum before): Box-Muller on the word pairs (0, 1) and
// (2, 3), both of its outputs used - spectrum 0 = r01 cos, 1 = r01 sin, 2 = r23 cos, 3 = r23 sin (independent standard
// normals).  Every sample depends only on what it belongs to, so any view sharding, either memory layout and every kernel
// reproduce the same sinogram.  The logarithm, the square roots and the sine / cosine are the hardware's (v_log_f32,
// v_sqrt_f32, v_sin_f32 / v_cos_f32 on an argument in revolutions: 1 ulp-class, far inside what a noise sample asks for):
// the sample costs ~130 vector instructions per pixel, not ~400.
#pragma once
#include "common.h"

namespace dexct {

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// z[s], s < N <= DEXCT_MAX_SPECTRA: the standard normals of the pixel's spectra
template <int N>
__device__ __forceinline__ void pixel_normals(uint32_t view, uint32_t row, uint32_t chan, uint32_t seed_lo, uint32_t seed_hi,
                                              float (&z)[N]) {
  static_assert(N >= 1 && N <= 4, "one Philox block = four words = four normals");
  uint32_t c[4] = {view, row, chan, 0u};
  philox4x32_10(c, seed_lo, seed_hi);
#pragma unroll
  for (int p = 0; 2 * p < N; ++p) {
    const float u1 = ((float)c[2 * p] + 1.0f) * 2.3283064365386963e-10f;      // (0, 1]
    const float u2 = (float)c[2 * p + 1] * 2.3283064365386963e-10f;           // [0, 1]: the angle in revolutions
    // sqrt(-2 ln u1), ln = ln 2 x log2
    const float rad = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));
    z[2 * p] = rad * __builtin_amdgcn_cosf(u2);
    if (2 * p + 1 < N) z[2 * p + 1] = rad * __builtin_amdgcn_sinf(u2);
  }
}

__device__ __forceinline__ float noisy_count(float mean, float variance, float z) {
  return fmaxf(fmaf(__builtin_amdgcn_sqrtf(fmaxf(variance, 0.0f)), z, mean), 1.0e-20f);
}

}  // namespace dexct
