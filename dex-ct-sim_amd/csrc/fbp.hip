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

// One thread per pixel (x, y) and R consecutive slices (rows).  q is [view][row][channel].  The geometry of a
// (pixel, view) pair - source distance, fan angle, interpolation weight, all in float64 - is the same for every
// row of a stacked fan, so it is computed once and applied to R rows (R = 8 for volumes, 1 for a single slice).
template <int R>
__global__ __launch_bounds__(kFbpBlock) void fbp_backproject_kernel(const float* __restrict__ q,
                                                                    const double* __restrict__ view_cs, FbpGeom g,
                                                                    float* __restrict__ img) {
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63);
  const int iy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int row0 = blockIdx.z * R;
  if (ix >= g.n_matrix || iy >= g.n_matrix) return;
  const double x = (ix - 0.5 * g.n_matrix + 0.5) * g.pixel, y = (iy - 0.5 * g.n_matrix + 0.5) * g.pixel;
  const double inv_dg = 1.0 / g.dgamma, half = 0.5 * (g.n_channels - 1);
  const int n_rows_here = g.n_rows - row0 < R ? g.n_rows - row0 : R;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.0f;
  for (int v = 0; v < g.n_views; ++v) {
    const double cb = view_cs[2 * v], sb = view_cs[2 * v + 1];   // wave-uniform
    const double dx = x - g.sid * cb, dy = y - g.sid * sb;
    const double dot = -(cb * dx + sb * dy), cross = -(cb * dy - sb * dx);
    const double pos = atan2(cross, dot) * inv_dg + half;
    const double fl = floor(pos);
    const int k = (int)fl;
    if (k >= 0 && k < g.n_channels - 1) {
      const float w = (float)(pos - fl);
      const float l2 = (float)(dx * dx + dy * dy);
      const float* ql = q + ((size_t)v * g.n_rows + row0) * g.n_channels + k;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        if (r < n_rows_here) {
          const float val = (1.0f - w) * ql[0] + w * ql[1];
          acc[r] += val / l2;
        }
        ql += g.n_channels;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r < n_rows_here) img[((size_t)(row0 + r) * g.n_matrix + iy) * g.n_matrix + ix] = acc[r] * (float)g.dbeta;
}

// Cone-beam (FDK) back-projection for the cylindrical detector of dexct_cone_project: equiangular in the fan,
// rows equally spaced in height on the cylinder of radius SDD around the source axis.  Feldkamp's algorithm is the
// fan-beam FBP above with two changes: the projections are weighted by the cosine of the cone angle,
// cos kappa_r = SDD / sqrt(SDD^2 + (row_z[r] - src_z)^2) - a per-row constant, so it commutes with the row-wise
// filter and is applied here -, and the voxel (x, y, z) reads the detector at height
// src_z + (z - src_z) * SDD / L (L = in-plane source-voxel distance), linearly interpolated between rows.
struct FdkGeom {
  int n_views, n_channels, n_rows, n_matrix, n_slices;
  double sid, sdd, dgamma, dbeta, pixel;
  double row_z0, row_dz, src_z, z0, dz;     // detector rows: row_z0 + r*row_dz; slices: z0 + k*dz  [cm]
};

template <int R>
__global__ __launch_bounds__(kFbpBlock) void fdk_backproject_kernel(const float* __restrict__ q,
                                                                    const double* __restrict__ view_cs,
                                                                    const float* __restrict__ row_weight, FdkGeom g,
                                                                    float* __restrict__ img) {
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63);
  const int iy = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int k0 = blockIdx.z * R;
  if (ix >= g.n_matrix || iy >= g.n_matrix) return;
  const double x = (ix - 0.5 * g.n_matrix + 0.5) * g.pixel, y = (iy - 0.5 * g.n_matrix + 0.5) * g.pixel;
  const double inv_dg = 1.0 / g.dgamma, half = 0.5 * (g.n_channels - 1), inv_rdz = 1.0 / g.row_dz;
  const int n_here = g.n_slices - k0 < R ? g.n_slices - k0 : R;
  float acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.0f;
  for (int v = 0; v < g.n_views; ++v) {
    const double cb = view_cs[2 * v], sb = view_cs[2 * v + 1];
    const double dx = x - g.sid * cb, dy = y - g.sid * sb;
    const double dot = -(cb * dx + sb * dy), cross = -(cb * dy - sb * dx);
    const double pos = atan2(cross, dot) * inv_dg + half;
    const double fl = floor(pos);
    const int k = (int)fl;
    if (k < 0 || k >= g.n_channels - 1) continue;
    const float w = (float)(pos - fl);
    const double l2 = dx * dx + dy * dy;
    const float inv_l2 = (float)(1.0 / l2);
    const double mag = g.sdd / sqrt(l2);                  // magnification of heights from the voxel to the detector
    const float* qv = q + (size_t)v * g.n_rows * g.n_channels + k;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r < n_here) {
        const double z = g.z0 + (k0 + r) * g.dz;
        const double rpos = (g.src_z + (z - g.src_z) * mag - g.row_z0) * inv_rdz;
        const double rfl = floor(rpos);
        const int r0 = (int)rfl;
        if (r0 >= 0 && r0 < g.n_rows - 1) {
          const float wr = (float)(rpos - rfl);
          const float* qa = qv + (size_t)r0 * g.n_channels;
          const float* qb = qa + g.n_channels;
          const float va = ((1.0f - w) * qa[0] + w * qa[1]) * row_weight[r0];
          const float vb = ((1.0f - w) * qb[0] + w * qb[1]) * row_weight[r0 + 1];
          acc[r] += ((1.0f - wr) * va + wr * vb) * inv_l2;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (r < n_here) img[((size_t)(k0 + r) * g.n_matrix + iy) * g.n_matrix + ix] = acc[r] * (float)g.dbeta;
}

}  // namespace dexct

using namespace dexct;

// Short-scan (Parker) weights.  A scan over theta_tot < 2 pi measures some rays once and some twice: ray (beta, gamma) is
// the ray (beta + pi + 2 gamma, -gamma) seen from the other side (this build's geometry: source at angle beta, channel
// looking along beta + pi + gamma).  With Gamma' = (theta_tot - pi) / 2 (>= the half fan angle, else data are missing),
//   w = sin^2(pi/4 * beta / (Gamma' - gamma))                        0 <= beta <= 2 Gamma' - 2 gamma
//   w = 1                                                            between
//   w = sin^2(pi/4 * (pi + 2 Gamma' - beta) / (Gamma' + gamma))      pi - 2 gamma <= beta <= theta_tot
// (Parker 1982 with Silver's virtual fan angle for scans longer than the minimum) - the two weights of a ray measured
// twice add up to 1, smoothly.  The full-scan formula of dexct_fbp_filter / _backproject counts every ray twice (the
// factor 1/2 sits in its filter taps), so the sinogram is multiplied by 2 w.
__global__ __launch_bounds__(256) void parker_kernel(const float* __restrict__ sino, int n_views, int n_rows, int n_channels,
                                                     double theta_tot, double dgamma, int view_offset, int n_views_total,
                                                     float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t n = (size_t)n_views * n_rows * n_channels;
  if (i >= n) return;
  const int c = (int)(i % n_channels);
  const int v = (int)(i / ((size_t)n_rows * n_channels)) + view_offset;
  const double kPi = 3.14159265358979323846;
  const double beta = theta_tot * (double)v / (double)n_views_total;
  const double gam = ((double)c - 0.5 * (double)(n_channels - 1)) * dgamma;
  const double G = 0.5 * (theta_tot - kPi);
  double w = 1.0;
  if (beta < 2.0 * (G - gam)) {
    const double sn = sin(0.25 * kPi * beta / (G - gam));
    w = sn * sn;
  } else if (beta > kPi - 2.0 * gam) {
    const double sn = sin(0.25 * kPi * (kPi + 2.0 * G - beta) / (G + gam));
    w = sn * sn;
  }
  out[i] = (float)(2.0 * w * (double)sino[i]);
}

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

int dexct_fbp_parker(const float* sino, int32_t n_views, int32_t n_rows, int32_t n_channels, double theta_tot, double dgamma,
                      int32_t view_offset, int32_t n_views_total, float* out, void* stream) {
  if (!sino || !out || n_views < 1 || n_rows < 1 || n_channels < 1 || n_views_total < n_views || view_offset < 0 ||
      view_offset + n_views > n_views_total)
    return DEXCT_EINVAL;
  const double kPi = 3.14159265358979323846;
  const double half_fan = 0.5 * (double)(n_channels - 1) * dgamma;
  // a short scan needs pi + the full fan angle; 2 pi and beyond is not a short scan
  if (!(dgamma > 0.0) || !(theta_tot < 2.0 * kPi) || !(0.5 * (theta_tot - kPi) >= half_fan * (1.0 - 1e-12))) return DEXCT_EINVAL;
  const size_t n = (size_t)n_views * n_rows * n_channels;
  const size_t nblk = (n + 255) / 256;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  hipLaunchKernelGGL(parker_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), sino, n_views, n_rows, n_channels,
                     theta_tot, dgamma, view_offset, n_views_total, out);
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
  if (n_rows >= 8) {
    dim3 grid((n_matrix + 63) / 64, (n_matrix + 3) / 4, (n_rows + 7) / 8);
    hipLaunchKernelGGL(fbp_backproject_kernel<8>, grid, dim3(kFbpBlock), 0, as_stream(stream), q, view_cs, g, image);
  } else {
    dim3 grid((n_matrix + 63) / 64, (n_matrix + 3) / 4, n_rows);
    hipLaunchKernelGGL(fbp_backproject_kernel<1>, grid, dim3(kFbpBlock), 0, as_stream(stream), q, view_cs, g, image);
  }
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_fdk_backproject(const float* q, const double* view_cs, const float* row_weight, int32_t n_views,
                          int32_t n_channels, int32_t n_rows, double sid, double sdd, double dgamma, double dbeta,
                          double row_z0, double row_dz, double src_z, int32_t n_matrix, double fov, int32_t n_slices,
                          double z0, double dz, float* image, void* stream) {
  if (!q || !view_cs || !row_weight || !image || n_views <= 0 || n_channels < 2 || n_rows < 2 || n_matrix <= 0 ||
      n_slices <= 0)
    return DEXCT_EINVAL;
  if (!(sid > 0) || !(sdd >= sid) || !(dgamma > 0) || !(fov > 0) || !(row_dz > 0) || !(dz > 0)) return DEXCT_EINVAL;
  if ((n_slices + 3) / 4 > 65535 || (n_matrix + 3) / 4 > 65535) return DEXCT_ERANGE;
  FdkGeom g{n_views, n_channels, n_rows, n_matrix, n_slices, sid, sdd, dgamma, dbeta, fov / n_matrix,
            row_z0, row_dz, src_z, z0, dz};
  dim3 grid((n_matrix + 63) / 64, (n_matrix + 3) / 4, (n_slices + 3) / 4);
  hipLaunchKernelGGL(fdk_backproject_kernel<4>, grid, dim3(kFbpBlock), 0, as_stream(stream), q, view_cs, row_weight, g,
                     image);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

}  // extern "C"

// Virtual monoenergetic image from two basis-material images (plots.py:136-144 of the reference):
// vmi = u1 * m1 + u2 * m2, optionally in Hounsfield units against water.
namespace dexct {
__global__ __launch_bounds__(256) void vmi_kernel(const float* __restrict__ m1, const float* __restrict__ m2, int64_t n,
                                                  double u1, double u2, double u_water, int hu, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // float64 like the reference (its float64 coefficient arrays promote the float32 images, plots.py:141), same
  // operation order, no contraction (-ffp-contract=off), one rounding to float32 at the end (:144): bit-identical
  double v = u1 * (double)m1[i] + u2 * (double)m2[i];
  if (hu) v = 1000.0 * (v - u_water) / u_water;
  out[i] = (float)v;
}
}  // namespace dexct

extern "C" int dexct_vmi(const float* m1, const float* m2, int64_t n, double u1, double u2, double u_water, int32_t hu,
                         float* out, void* stream) {
  using namespace dexct;
  if (!m1 || !m2 || !out || n <= 0 || (hu && !(u_water > 0))) return DEXCT_EINVAL;
  const int64_t nblk = (n + 255) / 256;
  if (nblk > 0x7FFFFFFFll) return DEXCT_ERANGE;
  hipLaunchKernelGGL(vmi_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), m1, m2, n, u1, u2, u_water, hu, out);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

// Second-order moments of one or two images per label: the sufficient statistics of every ROI measurement the
// reference's analysis script takes (measure_roi, plots.py:146-158; the VMI RMSE and CNR sweeps, :297-303 and
// :371-393, are closed forms in these sums because a VMI is linear in the two basis images).
// out[label][6] = { count, S(m1), S(m2), S(m1^2), S(m1 m2), S(m2^2) } in float64.
namespace dexct {
constexpr int kMomBlock = 256, kMomPerThread = 16, kMomMaxLabels = 64;

__global__ __launch_bounds__(kMomBlock) void label_moments_kernel(const float* __restrict__ m1,
                                                                  const float* __restrict__ m2,
                                                                  const uint8_t* __restrict__ labels, int64_t n,
                                                                  int n_labels, double* __restrict__ out) {
  __shared__ double acc[kMomMaxLabels * 6];
  for (int i = threadIdx.x; i < n_labels * 6; i += kMomBlock) acc[i] = 0.0;
  __syncthreads();
  // a thread walks pixels 256 apart, so consecutive pixels of a thread lie in the same region most of the time:
  // sums stay in registers and go to LDS only when the label changes
  int cur = -1;
  double c = 0, s1 = 0, s2 = 0, s11 = 0, s12 = 0, s22 = 0;
  auto flush = [&]() {
    if (cur >= 0 && c > 0) {
      double* a = acc + cur * 6;
      atomicAdd(a + 0, c); atomicAdd(a + 1, s1); atomicAdd(a + 2, s2);
      atomicAdd(a + 3, s11); atomicAdd(a + 4, s12); atomicAdd(a + 5, s22);
    }
    c = s1 = s2 = s11 = s12 = s22 = 0;
  };
  const int64_t base = (int64_t)blockIdx.x * (kMomBlock * kMomPerThread) + threadIdx.x;
  for (int k = 0; k < kMomPerThread; ++k) {
    const int64_t i = base + (int64_t)k * kMomBlock;
    if (i >= n) break;
    const int l = labels ? (int)labels[i] : 0;
    if (l >= n_labels) continue;
    if (l != cur) { flush(); cur = l; }
    const double a = (double)m1[i], b = m2 ? (double)m2[i] : 0.0;
    c += 1.0; s1 += a; s2 += b; s11 += a * a; s12 += a * b; s22 += b * b;
  }
  flush();
  __syncthreads();
  for (int i = threadIdx.x; i < n_labels * 6; i += kMomBlock)
    if (acc[i] != 0.0) atomicAdd(out + i, acc[i]);
}
}  // namespace dexct

extern "C" int dexct_label_moments(const float* m1, const float* m2, const uint8_t* labels, int64_t n, int32_t n_labels,
                                   double* out, void* stream) {
  using namespace dexct;
  if (!m1 || !out || n <= 0 || n_labels <= 0) return DEXCT_EINVAL;
  if (n_labels > kMomMaxLabels) return DEXCT_ERANGE;
  const int64_t nblk = (n + kMomBlock * kMomPerThread - 1) / (kMomBlock * kMomPerThread);
  if (nblk > 0x7FFFFFFFll) return DEXCT_ERANGE;
  DEXCT_HIP_TRY(hipMemsetAsync(out, 0, sizeof(double) * 6 * n_labels, as_stream(stream)));
  hipLaunchKernelGGL(label_moments_kernel, dim3((unsigned)nblk), dim3(kMomBlock), 0, as_stream(stream), m1, m2, labels, n,
                     n_labels, out);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}
