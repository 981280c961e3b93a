// Quantum noise on the detected sinogram (the reference scales spectra to a dose per view,
// main.py:64-69, and treats sino_raw as photon counts, matdecomp.py:30,179 - its own noise code sits in the
// absent x-tomo-sim submodule).
//
// Model: per energy bin the detected photons are Poisson(lambda_e) and each carries the detector signal
// gain_e (E for an energy-integrating detector, 1 for a counting one), so the signal has mean
// sum_e gain_e lambda_e (= the noise-free count the projector writes) and variance sum_e gain_e^2 lambda_e
// (written by the projector when asked).  The sum over ~10^2 bins of 10^3..10^6 photons is drawn as
// mean + sqrt(variance) * z with z standard normal; results are clipped at a tiny positive number so
// that the log sinogram stays finite.
// RNG: Philox4x32-10, one block per pixel (noise_sample.h): every sample depends only on what it belongs to, so any view
// sharding and either memory layout reproduce the same sinogram - and so do the projection kernels that draw the sample
// themselves (dexct_noise.sample), which share the definition.
#include "common.h"
#include "noise_sample.h"

namespace dexct {

__global__ __launch_bounds__(256) void add_noise_kernel(float* __restrict__ counts, const float* __restrict__ variance,
                                                        int n_spectra, int n_views, int n_rows, int n_channels,
                                                        int layout, int view_offset, uint32_t seed_lo, uint32_t seed_hi) {
  const size_t n_rays = (size_t)n_views * n_rows * n_channels;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rays * n_spectra) return;
  const uint32_t s = (uint32_t)(i / n_rays);
  size_t ray = i - (size_t)s * n_rays;
  uint32_t v, r, c;
  if (layout == 0) { c = ray % n_channels; ray /= n_channels; r = ray % n_rows; v = (uint32_t)(ray / n_rows); }
  else             { r = ray % n_rows; ray /= n_rows; c = ray % n_channels; v = (uint32_t)(ray / n_channels); }
  float z[DEXCT_MAX_SPECTRA];
  pixel_normals<DEXCT_MAX_SPECTRA>(v + (uint32_t)view_offset, r, c, seed_lo, seed_hi, z);
  const float zs = s == 0u ? z[0] : (s == 1u ? z[1] : (s == 2u ? z[2] : z[3]));      // (selects: a run-time index would put z[] in scratch)
  counts[i] = noisy_count(counts[i], variance[i], zs);
}

}  // namespace dexct

using namespace dexct;

extern "C" int dexct_add_noise(float* counts, const float* variance, int32_t n_spectra, int32_t n_views, int32_t n_rows,
                               int32_t n_channels, int32_t layout, int32_t view_offset, uint64_t seed, void* stream) {
  if (!counts || !variance || n_spectra < 1 || n_views < 1 || n_rows < 1 || n_channels < 1) return DEXCT_EINVAL;
  if (layout != 0 && layout != 1) return DEXCT_EINVAL;
  const size_t n = (size_t)n_spectra * n_views * n_rows * n_channels;
  const size_t nblk = (n + 255) / 256;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), counts, variance, n_spectra,
                     n_views, n_rows, n_channels, layout, view_offset, (uint32_t)seed, (uint32_t)(seed >> 32));
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

// sino_log = ln(air / counts) as a pass of its own: for the noisy sinograms (their counts exist only after the sampling
// above) and for callers that hold counts without having projected them here.
namespace dexct {
struct AirValues { float v[DEXCT_MAX_SPECTRA]; };
__global__ __launch_bounds__(256) void sino_log_kernel(const float* __restrict__ counts, AirValues air, size_t n_rays,
                                                       size_t n_total, int vec_ok, float* __restrict__ out) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n_total) return;
  if (i + 4 <= n_total && vec_ok) {
    const float a = air.v[i / n_rays];
    const float4 c = *reinterpret_cast<const float4*>(counts + i);
    *reinterpret_cast<float4*>(out + i) = make_float4(log_ratio(a, c.x), log_ratio(a, c.y), log_ratio(a, c.z), log_ratio(a, c.w));
  } else {
    for (size_t k = i; k < n_total && k < i + 4; ++k) out[k] = log_ratio(air.v[k / n_rays], counts[k]);
  }
}
}  // namespace dexct

extern "C" int dexct_sino_log(const float* counts, const float* air, int32_t n_spectra, int64_t n_rays, float* sino_log,
                              void* stream) {
  using namespace dexct;
  if (!counts || !air || !sino_log || n_spectra < 1 || n_rays < 1) return DEXCT_EINVAL;
  if (n_spectra > DEXCT_MAX_SPECTRA) return DEXCT_ERANGE;
  AirValues av;
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) av.v[s] = s < n_spectra ? air[s] : 1.0f;
  const size_t n_total = (size_t)n_spectra * (size_t)n_rays;
  const size_t nblk = (n_total / 4 + 256) / 256;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  // 16-byte loads and stores only when a spectrum's block starts on a multiple of 4 values AND both buffers are 16-byte
  // aligned (a view into a larger tensor need not be); scalar otherwise, same values
  const int vec_ok = (n_rays & 3) == 0 && ((reinterpret_cast<uintptr_t>(counts) | reinterpret_cast<uintptr_t>(sino_log)) & 15u) == 0;
  hipLaunchKernelGGL(sino_log_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), counts, av, (size_t)n_rays,
                     n_total, vec_ok, sino_log);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

// [batch][rows][cols] -> [batch][cols][rows] of float32 counts AND, from the same registers, the log sinogram of the transposed
// counts: what turns the row-parallel kernels' native [view][channel][row] sinograms into get_sino's two outputs
// ([view][row][channel], main.py:120-122) in ONE pass - read once, written twice - instead of a log written natively and two
// transposes (13.1 GB of traffic at the benchmark's size, and 3.3 GB of projection stores, become 9.8 GB).  64 x 64 tiles through
// LDS (pitch 65: conflict-free both ways), 16-byte global loads and stores.  ln(air / c) by the same log_ratio as every
// detection store: the same bits as the log the projection kernel would have written.
namespace dexct {
template <bool WITH_LOG>
__global__ __launch_bounds__(256) void transpose_log_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                            float* __restrict__ log_dst, AirValues air, int per_spectrum,
                                                            int batch0, int rows, int cols) {
  __shared__ float tile[64][65];
  const int b = batch0 + blockIdx.z;
  const size_t base = (size_t)b * rows * cols;
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int q = threadIdx.x & 15, k4 = threadIdx.x >> 4;        // 16 x 16
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = r0 + k4 + 16 * p, c = c0 + 4 * q;
    if (r < rows && c < cols) {                                 // (cols % 4 == 0: a float4 is inside or outside)
      const float4 x = *reinterpret_cast<const float4*>(src + base + (size_t)r * cols + c);
      tile[k4 + 16 * p][4 * q] = x.x; tile[k4 + 16 * p][4 * q + 1] = x.y;
      tile[k4 + 16 * p][4 * q + 2] = x.z; tile[k4 + 16 * p][4 * q + 3] = x.w;
    }
  }
  __syncthreads();
  const float a = WITH_LOG ? air.v[b / per_spectrum] : 1.0f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = c0 + k4 + 16 * p, r = r0 + 4 * q;
    if (c < cols && r < rows) {                                 // (rows % 4 == 0)
      const float4 x = make_float4(tile[4 * q][k4 + 16 * p], tile[4 * q + 1][k4 + 16 * p], tile[4 * q + 2][k4 + 16 * p],
                                   tile[4 * q + 3][k4 + 16 * p]);
      const size_t at = base + (size_t)c * rows + r;
      *reinterpret_cast<float4*>(dst + at) = x;
      if (WITH_LOG)
        *reinterpret_cast<float4*>(log_dst + at) = make_float4(log_ratio(a, x.x), log_ratio(a, x.y), log_ratio(a, x.z), log_ratio(a, x.w));
    }
  }
}
}  // namespace dexct

extern "C" int dexct_transpose_log(const float* src, float* dst, float* log_dst, const float* air, int32_t n_spectra,
                                   int64_t batch_per_spectrum, int32_t rows, int32_t cols, void* stream) {
  using namespace dexct;
  if (!src || !dst || n_spectra < 1 || batch_per_spectrum < 1 || rows < 1 || cols < 1) return DEXCT_EINVAL;
  if (log_dst && !air) return DEXCT_EINVAL;
  if (n_spectra > DEXCT_MAX_SPECTRA) return DEXCT_ERANGE;
  const int64_t batch = (int64_t)n_spectra * batch_per_spectrum;
  if (batch > 0x7FFFFFFFll || batch_per_spectrum > 0x7FFFFFFFll) return DEXCT_ERANGE;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(log_dst);
  if ((rows & 3) || (cols & 3) || (ptrs & 15u)) {
    // shapes or views the 16-byte accesses do not fit: the generic transpose, then the log of the transposed counts (same values)
    int rc = dexct_transpose_batched(src, dst, batch, rows, cols, 4, stream);
    if (rc == DEXCT_OK && log_dst) rc = dexct_sino_log(dst, air, n_spectra, batch_per_spectrum * rows * cols, log_dst, stream);
    return rc;
  }
  AirValues av;
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) av.v[s] = (air && s < n_spectra) ? air[s] : 1.0f;
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {             // gridDim.z is limited to 65535: walk the batch in slices
    const int nb = (int)((batch - b0) < 65535 ? (batch - b0) : 65535);
    const dim3 grid((cols + 63) / 64, (rows + 63) / 64, nb);
    if (log_dst)
      hipLaunchKernelGGL(transpose_log_kernel<true>, grid, dim3(256), 0, as_stream(stream), src, dst, log_dst, av,
                         (int)batch_per_spectrum, (int)b0, rows, cols);
    else
      hipLaunchKernelGGL(transpose_log_kernel<false>, grid, dim3(256), 0, as_stream(stream), src, dst, log_dst, av,
                         (int)batch_per_spectrum, (int)b0, rows, cols);
    DEXCT_LAUNCH_CHECK();
  }
  return DEXCT_OK;
}

// ---------------------------------------------------------------------------------------------
// Exact model for photon-starved rays: per energy bin N_e ~ Poisson(lambda_e), lambda_e = photons[s][e] *
// exp(-sum_m mu[m][e] L_m), signal = sum_e gain[e] * N_e.  One Philox block per (ray, spectrum, energy).
// Sampling: sequential inversion below lambda = 30 (exact; the loop is short because lambda is small),
// rounded normal above (relative skewness error < 1/sqrt(30) of one count).
namespace dexct {

__device__ __forceinline__ float poisson_draw(float lambda, uint32_t r0, uint32_t r1, uint32_t r2) {
  if (!(lambda > 0.0f)) return 0.0f;
  if (lambda < 30.0f) {
    // 53-bit uniform from two words keeps the tail of the inversion honest
    const double u = ((double)r0 * 4294967296.0 + (double)r1 + 0.5) * (1.0 / 18446744073709551616.0);
    double p = exp(-(double)lambda), cdf = p;
    int k = 0;
    while (u > cdf && k < 200) {
      ++k;
      p *= (double)lambda / k;
      cdf += p;
    }
    return (float)k;
  }
  const float u1 = ((float)r0 + 1.0f) * 2.3283064365386963e-10f, u2 = (float)r2 * 2.3283064365386963e-10f;
  const float z = sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
  return fmaxf(floorf(fmaf(sqrtf(lambda), z, lambda) + 0.5f), 0.0f);
}

template <int MB>   // material capacity of the register array
__global__ __launch_bounds__(256) void poisson_detect_kernel(const float* __restrict__ pathlen, const float* __restrict__ mu,
                                                             const float* __restrict__ photons,
                                                             const float* __restrict__ gain, int n_mat, int n_e,
                                                             int n_spectra, int n_views, int n_rows, int n_channels,
                                                             int layout, int view_offset, uint32_t seed_lo,
                                                             uint32_t seed_hi, float* __restrict__ counts) {
  const size_t n_rays = (size_t)n_views * n_rows * n_channels;
  const size_t ray = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (ray >= n_rays) return;
  size_t q = ray;
  uint32_t v, r, c;
  if (layout == 0) { c = q % n_channels; q /= n_channels; r = q % n_rows; v = (uint32_t)(q / n_rows); }
  else             { r = q % n_rows; q /= n_rows; c = q % n_channels; v = (uint32_t)(q / n_channels); }
  float L[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) L[m] = m < n_mat ? pathlen[ray * n_mat + m] * 1.44269504088896340736f : 0.0f;
  float acc[DEXCT_MAX_SPECTRA];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) acc[s] = 0.0f;
  for (int e = 0; e < n_e; ++e) {
    float pe = 0.0f;
#pragma unroll
    for (int m = 0; m < MB; ++m)
      if (m < n_mat) pe = fmaf(mu[m * n_e + e], L[m], pe);
    const float t = __builtin_amdgcn_exp2f(-pe);
    const float ge = gain[e];
    for (int s = 0; s < n_spectra; ++s) {
      const float lambda = photons[s * n_e + e] * t;
      if (lambda > 0.0f) {
        uint32_t ctr[4] = {v + (uint32_t)view_offset, r, c, ((uint32_t)s << 24) | (uint32_t)e};
        philox4x32_10(ctr, seed_lo, seed_hi ^ 0x9E3779B9u);        // stream distinct from the Gaussian sampler's
        acc[s] = fmaf(ge, poisson_draw(lambda, ctr[0], ctr[1], ctr[2]), acc[s]);
      }
    }
  }
  for (int s = 0; s < n_spectra; ++s) counts[ray + (size_t)s * n_rays] = fmaxf(acc[s], 1.0e-20f);
}

}  // namespace dexct

extern "C" int dexct_poisson_detect(const float* pathlen, const float* mu, const float* photons, const float* gain,
                                    int32_t n_materials, int32_t n_energies, int32_t n_spectra, int32_t n_views,
                                    int32_t n_rows, int32_t n_channels, int32_t layout, int32_t view_offset, uint64_t seed,
                                    float* counts, void* stream) {
  using namespace dexct;
  if (!pathlen || !mu || !photons || !gain || !counts) return DEXCT_EINVAL;
  if (n_materials < 1 || n_energies < 1 || n_energies >= (1 << 24) || n_spectra < 1 || n_views < 1 || n_rows < 1 ||
      n_channels < 1)
    return DEXCT_EINVAL;
  if (n_materials > DEXCT_MAX_MATERIALS || n_spectra > DEXCT_MAX_SPECTRA) return DEXCT_ERANGE;
  if (layout != 0 && layout != 1) return DEXCT_EINVAL;
  const size_t n_rays = (size_t)n_views * n_rows * n_channels;
  const size_t nblk = (n_rays + 255) / 256;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  const dim3 grid((unsigned)nblk), block(256);
  hipStream_t st = as_stream(stream);
  const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
#define DEXCT_LAUNCH_POISSON(MB)                                                                                       \
  hipLaunchKernelGGL(poisson_detect_kernel<MB>, grid, block, 0, st, pathlen, mu, photons, gain, n_materials, n_energies, \
                     n_spectra, n_views, n_rows, n_channels, layout, view_offset, lo, hi, counts)
  if (n_materials <= 4) DEXCT_LAUNCH_POISSON(4);
  else if (n_materials <= 16) DEXCT_LAUNCH_POISSON(16);
  else if (n_materials <= 48) DEXCT_LAUNCH_POISSON(48);
  else DEXCT_LAUNCH_POISSON(DEXCT_MAX_MATERIALS);          // (the lengths of up to 256 materials: a per-lane array in scratch)
#undef DEXCT_LAUNCH_POISSON
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}
