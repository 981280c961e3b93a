// Equiangular fan-beam filtered back-projection for gfx950.
//
// Replaces get_recon (reference call sites main.py:134,168; source in the un-vendored x-tomo-sim
// submodule, described at README.md:30-31).  Kak & Slaney 3.4.1, restated in oracle/fbp_oracle.py:
//   filter:        Q(line, n) = dg * sum_m R(line, m) * D cos(gamma_m) * g[n - m]
//   back-project:  f(x, y)    = dbeta * sum_views Q(view, gamma'(x, y)) / L^2,   linear interpolation
// A "line" is one (view, row) of the log sinogram.  The filter is a direct convolution (N_channels is
// ~10^3: 0.6 MFLOP per line, LDS-resident taps and line; no FFT library needed and the result does not
// depend on an FFT's rounding); back-projection is pixel driven with the geometry (source offset, fan
// angle, interpolation weight) in float64 and the filtered values in float32.
#include "common.h"

namespace dexct {

constexpr int kFbpBlock = 256;

// One workgroup per line.  LDS: weighted line [n] + taps [2n - 1].
__global__ __launch_bounds__(kFbpBlock) void fbp_filter_kernel(const float* __restrict__ sino,
                                                               const float* __restrict__ taps,
                                                               const float* __restrict__ weight, int n, float dgamma,
                                                               float* __restrict__ q) {
  extern __shared__ float lds[];
  float* line = lds;            // [n]
  float* g = lds + n;           // [2n - 1], g[k] = tap of offset k - (n - 1)
  const size_t base = (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += kFbpBlock) line[i] = sino[base + i] * weight[i];
  for (int i = threadIdx.x; i < 2 * n - 1; i += kFbpBlock) g[i] = taps[i];
  __syncthreads();
  for (int k = threadIdx.x; k < n; k += kFbpBlock) {
    float acc0 = 0.0f, acc1 = 0.0f;
    const float* gk = g + k + (n - 1);     // gk[-m] = g[(k - m) + n - 1]
    int m = 0;
    for (; m + 1 < n; m += 2) {
      acc0 = fmaf(line[m], gk[-m], acc0);
      acc1 = fmaf(line[m + 1], gk[-m - 1], acc1);
    }
    if (m < n) acc0 = fmaf(line[m], gk[-m], acc0);
    q[base + k] = (acc0 + acc1) * dgamma;
  }
}

struct FbpGeom {
  int n_views, n_channels, n_rows, n_matrix;
  double sid, dgamma, dbeta, pixel;   // pixel = FOV / n_matrix
};

// One thread per pixel of one slice (row).  q is [view][row][channel].
__global__ __launch_bounds__(kFbpBlock) void fbp_backproject_kernel(const float* __restrict__ q,
                                                                    const double* __restrict__ view_cs, FbpGeom g,
                                                                    float* __restrict__ img) {
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63);
  const int iy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int row = blockIdx.z;
  if (ix >= g.n_matrix || iy >= g.n_matrix) return;
  const double x = (ix - 0.5 * g.n_matrix + 0.5) * g.pixel, y = (iy - 0.5 * g.n_matrix + 0.5) * g.pixel;
  const double inv_dg = 1.0 / g.dgamma, half = 0.5 * (g.n_channels - 1);
  float acc = 0.0f;
  for (int v = 0; v < g.n_views; ++v) {
    const double cb = view_cs[2 * v], sb = view_cs[2 * v + 1];   // wave-uniform
    const double dx = x - g.sid * cb, dy = y - g.sid * sb;
    const double dot = -(cb * dx + sb * dy), cross = -(cb * dy - sb * dx);
    const double pos = atan2(cross, dot) * inv_dg + half;
    const double fl = floor(pos);
    const int k = (int)fl;
    if (k >= 0 && k < g.n_channels - 1) {
      const float w = (float)(pos - fl);
      const float* ql = q + ((size_t)v * g.n_rows + row) * g.n_channels + k;
      const float val = (1.0f - w) * ql[0] + w * ql[1];
      acc += val / (float)(dx * dx + dy * dy);
    }
  }
  img[((size_t)row * g.n_matrix + iy) * g.n_matrix + ix] = acc * (float)g.dbeta;
}

}  // namespace dexct

using namespace dexct;

extern "C" {

int dexct_fbp_filter(const float* sino, const float* taps, const float* weight, int64_t n_lines, int32_t n_channels,
                     double dgamma, float* q, void* stream) {
  if (!sino || !taps || !weight || !q || n_lines <= 0 || n_channels < 2) return DEXCT_EINVAL;
  if (n_lines > 0x7FFFFFFFll || n_channels > 5000) return DEXCT_ERANGE;   // 3n floats of LDS <= 60 KB
  const size_t lds = (size_t)(3 * n_channels - 1) * sizeof(float);
  hipLaunchKernelGGL(fbp_filter_kernel, dim3((unsigned)n_lines), dim3(kFbpBlock), lds, as_stream(stream), sino, taps,
                     weight, n_channels, (float)dgamma, q);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_fbp_backproject(const float* q, const double* view_cs, int32_t n_views, int32_t n_channels, int32_t n_rows,
                          double sid, double dgamma, double dbeta, int32_t n_matrix, double fov, float* image,
                          void* stream) {
  if (!q || !view_cs || !image || n_views <= 0 || n_channels < 2 || n_rows <= 0 || n_matrix <= 0) return DEXCT_EINVAL;
  if (n_rows > 65535 || (n_matrix + 3) / 4 > 65535) return DEXCT_ERANGE;
  if (!(sid > 0) || !(dgamma > 0) || !(fov > 0)) return DEXCT_EINVAL;
  FbpGeom g{n_views, n_channels, n_rows, n_matrix, sid, dgamma, dbeta, fov / n_matrix};
  dim3 grid((n_matrix + 63) / 64, (n_matrix + 3) / 4, n_rows);
  hipLaunchKernelGGL(fbp_backproject_kernel, grid, dim3(kFbpBlock), 0, as_stream(stream), q, view_cs, g, image);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

}  // extern "C"

// Virtual monoenergetic image from two basis-material images (plots.py:136-144 of the reference):
// vmi = u1 * m1 + u2 * m2, optionally in Hounsfield units against water.
namespace dexct {
__global__ __launch_bounds__(256) void vmi_kernel(const float* __restrict__ m1, const float* __restrict__ m2, int64_t n,
                                                  float u1, float u2, float u_water, int hu, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = u1 * m1[i] + u2 * m2[i];
  if (hu) v = 1000.0f * (v - u_water) / u_water;
  out[i] = v;
}
}  // namespace dexct

extern "C" int dexct_vmi(const float* m1, const float* m2, int64_t n, double u1, double u2, double u_water, int32_t hu,
                         float* out, void* stream) {
  using namespace dexct;
  if (!m1 || !m2 || !out || n <= 0 || (hu && !(u_water > 0))) return DEXCT_EINVAL;
  const int64_t nblk = (n + 255) / 256;
  if (nblk > 0x7FFFFFFFll) return DEXCT_ERANGE;
  hipLaunchKernelGGL(vmi_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), m1, m2, n, (float)u1, (float)u2,
                     (float)u_water, hu, out);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}
