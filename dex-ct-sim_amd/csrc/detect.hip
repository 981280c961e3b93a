// The detection pass of the material-group projection (dexct_siddon_project_grouped[_packed]): per-material path lengths from
// the accumulator planes -> counts of every spectrum.  Its own translation unit since round 5: 94 instantiations (one per number
// of table rows, lengths in registers) that used to make siddon.hip 100 s of the library's 2-minute build.
#include "common.h"
#include "noise_sample.h"
#include "siddon_detect.h"

namespace dexct {

// One thread per ray, in memory order of the chosen layout; NMAT = exact number of materials (fully unrolled:
// a version with 16 predicated material slots spent its time in scalar branches).
template <int NMAT, int R>   // R rays per thread: 4 (or, beyond 16 materials, 2) consecutive rows (layout 1, n_rows % R == 0) or 1
__global__ __launch_bounds__(256) void detect_kernel(ProjArgs a, const float* __restrict__ mu, const float* __restrict__ w,
                                                     const float* __restrict__ w2) {
  const size_t n_rays = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  const size_t ray0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * R;
  if (ray0 >= n_rays) return;
  size_t q = ray0;
  int v, c;
  if (a.layout == 0) { c = q % a.g.n_channels; q /= a.g.n_channels; v = (int)(q / a.g.n_rows); }
  else               { q /= a.g.n_rows; c = q % a.g.n_channels; v = (int)(q / a.g.n_channels); }
  float L[R][NMAT];
  size_t rays[R];
  bool valid[R];
  if (a.acc_lengths) {        // cone-beam group passes: the planes hold lengths [cm] of ALL materials; no plan, no chord
#pragma unroll
    for (int m = 0; m < NMAT; ++m) {
      const float* plane = a.acc_out + (size_t)m * n_rays + ray0;
#pragma unroll
      for (int k = 0; k < R; ++k) L[k][m] = ray0 + k < n_rays ? plane[k] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      rays[k] = ray0 + k;
      valid[k] = ray0 + k < n_rays;
    }
  } else {
    const dexct_ray_plan p = a.plan[(size_t)v * a.g.n_channels + c];    // the R rays share (view, channel)
    float others[R];
#pragma unroll
    for (int k = 0; k < R; ++k) others[k] = 0.0f;
#pragma unroll
    for (int m = 1; m < NMAT; ++m) {
      const float* plane = a.acc_out + (size_t)m * n_rays + ray0;
      if (R == 4) {
        const float4 x = *reinterpret_cast<const float4*>(plane);
        L[0][m] = x.x; L[R > 1 ? 1 : 0][m] = x.y; L[R > 2 ? 2 : 0][m] = x.z; L[R > 3 ? 3 : 0][m] = x.w;
      } else if (R == 2) {
        const float2 x = *reinterpret_cast<const float2*>(plane);
        L[0][m] = x.x; L[R > 1 ? 1 : 0][m] = x.y;
      } else {
        L[0][m] = plane[0];
      }
#pragma unroll
      for (int k = 0; k < R; ++k) others[k] += L[k][m];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
      L[k][0] = (p.chord_u - others[k]) * p.len_per_u;
#pragma unroll
      for (int m = 1; m < NMAT; ++m) L[k][m] *= p.len_per_u;
      rays[k] = ray0 + k;
      valid[k] = true;
    }
  }
  if (a.sample) {
    // quantum noise drawn here (round 6; <= 2 spectra): the variance from the same exponentials as the signal, one Philox block
    // per ray - the sample dexct_add_noise draws from the same signal and variance; a.variance (optional) receives the variance
    float res[2][R], var[2][R];
    if (a.pathlen) {
#pragma unroll
      for (int k = 0; k < R; ++k)
        if (valid[k]) {
#pragma unroll
          for (int m = 0; m < NMAT; ++m) a.pathlen[rays[k] * NMAT + m] = L[k][m];
        }
    }
    detect_store<NMAT, R, false, true>(L, a, mu, w, w2, rays, valid, BlockMasks{{~0ull, ~0ull}, false}, nullptr, &res, &var);
    const int rr0 = a.layout == 0 ? (int)((ray0 / a.g.n_channels) % a.g.n_rows) : (int)(ray0 % a.g.n_rows);     // row of the first ray
#pragma unroll
    for (int k = 0; k < R; ++k) {
      if (a.variance && valid[k]) {
#pragma unroll
        for (int sI = 0; sI < 2; ++sI)
          if (sI < a.n_spectra) a.variance[rays[k] + (size_t)sI * n_rays] = var[sI][k];
      }
      // the R rays of a thread are consecutive rows of one (view, channel) pair (layout 1) or R = 1
      float z[2];
      pixel_normals<2>((uint32_t)(a.view_begin + v), (uint32_t)(rr0 + k), (uint32_t)c, a.seed_lo, a.seed_hi, z);
      res[0][k] = noisy_count(res[0][k], var[0][k], z[0]);
      res[1][k] = noisy_count(res[1][k], var[1][k], z[1]);
    }
    store_rays<R>(a, rays, valid, res);
    return;
  }
  detect_store<NMAT, R>(L, a, mu, w, w2, rays, valid);
}

// More than 48 table rows (49 ... 256): one ray per lane, the EXPONENTS of a block of 32 energies in registers, the materials
// in a run-time loop - every material's accumulator plane is read once per block of energies (coalesced, L2 resident), its
// 32 table values arrive as scalar operands.  M x n_energies FMAs per ray like every detection, and nothing in LDS (round
// 4's first form kept the lengths of all materials in per-lane LDS columns and re-read them per energy: 242 ms for 58 rows
// on 1e8 rays, profiles/r04_ids.log; this form: the FMA count).  Same operations in the same order as detect_store_lds
// (exponent summed from material 0 up, exp2 of -p log2 e, energies in order): the same counts bit for bit.
constexpr int kDetChunk = 32;
__global__ __launch_bounds__(256) void detect_kernel_chunked(ProjArgs a, const float* __restrict__ mu,
                                                             const float* __restrict__ w, const float* __restrict__ w2) {
  const size_t n_rays = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  const size_t ray = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (ray >= n_rays) return;
  size_t q = ray;
  int v, c;
  if (a.layout == 0) { c = q % a.g.n_channels; q /= a.g.n_channels; v = (int)(q / a.g.n_rows); }
  else               { q /= a.g.n_rows; c = q % a.g.n_channels; v = (int)(q / a.g.n_channels); }
  const bool lengths = a.acc_lengths != 0;      // (cone-beam group passes: every plane holds lengths [cm]; no plan)
  dexct_ray_plan p;
  if (lengths) { p.chord_u = 0.0f; p.len_per_u = 1.0f; }
  else p = a.plan[(size_t)v * a.g.n_channels + c];
  const int n_e = a.n_energies, n_mat = a.n_materials;
  const size_t sstride = n_rays;
  // material 0 fills what the others leave of the chord (same sum, same order as detect_kernel_lds)
  float others = 0.0f;
  for (int m = 1; m < n_mat; ++m) {
    const float l = a.acc_out[(size_t)m * n_rays + ray];
    if (a.pathlen) a.pathlen[ray * n_mat + m] = l * p.len_per_u;
    others += l;
  }
  const float L0 = lengths ? a.acc_out[ray] : (p.chord_u - others) * p.len_per_u;
  if (a.pathlen) a.pathlen[ray * n_mat] = L0;
  float acc[DEXCT_MAX_SPECTRA], var[DEXCT_MAX_SPECTRA];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) acc[s] = var[s] = 0.0f;
  for (int e0 = 0; e0 < n_e; e0 += kDetChunk) {
    const int ne = min(kDetChunk, n_e - e0);                      // wave-uniform
    float pe[kDetChunk];
#pragma unroll
    for (int k = 0; k < kDetChunk; ++k) pe[k] = 0.0f;
    for (int m = 0; m < n_mat; ++m) {
      const float L = m == 0 ? L0 : a.acc_out[(size_t)m * n_rays + ray] * p.len_per_u;
      const float* __restrict__ row = mu + (size_t)m * n_e + e0;  // wave-uniform: scalar loads
      if (ne == kDetChunk) {
#pragma unroll
        for (int k = 0; k < kDetChunk; ++k) pe[k] = fmaf(row[k], L, pe[k]);
      } else {
#pragma unroll
        for (int k = 0; k < kDetChunk; ++k)
          if (k < ne) pe[k] = fmaf(row[k], L, pe[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < kDetChunk; ++k) {
      if (k < ne) {
        const float t = __builtin_amdgcn_exp2f(-pe[k] * kLog2e);
#pragma unroll
        for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
          if (s < a.n_spectra) {
            acc[s] = fmaf(w[s * n_e + e0 + k], t, acc[s]);
            if (a.variance) var[s] = fmaf(w2[s * n_e + e0 + k], t, var[s]);
          }
      }
    }
  }
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
    if (s < a.n_spectra) {
      a.counts[ray + s * sstride] = acc[s];
      if (a.sino_log) a.sino_log[ray + s * sstride] = log_ratio(a.air[s], acc[s]);
      if (a.variance) a.variance[ray + s * sstride] = var[s];
    }
}

// rays per thread: 4 up to 16 materials, 2 up to 32, 1 beyond (the lengths of all materials live in registers)
template <int NMAT>
static int launch_detect(const ProjArgs& a, const Tables& t, hipStream_t st) {
  constexpr int RMAX = NMAT <= 16 ? 4 : (NMAT <= 32 ? 2 : 1);
  const size_t n_rays = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  const bool wide = RMAX > 1 && a.layout == 1 && (a.g.n_rows % RMAX) == 0;
  const size_t n_thr = wide ? n_rays / RMAX : n_rays;
  const size_t nblk = (n_thr + 255) / 256;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  if (wide) hipLaunchKernelGGL((detect_kernel<NMAT, RMAX>), dim3((unsigned)nblk), dim3(256), 0, st, a, t.mu, t.w, t.w2);
  else hipLaunchKernelGGL((detect_kernel<NMAT, 1>), dim3((unsigned)nblk), dim3(256), 0, st, a, t.mu, t.w, t.w2);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int launch_detect_any(const ProjArgs& a, const Tables& t, hipStream_t st) {
  switch (a.n_materials) {
    case 2: return launch_detect<2>(a, t, st);
    case 3: return launch_detect<3>(a, t, st);
    case 4: return launch_detect<4>(a, t, st);
    case 5: return launch_detect<5>(a, t, st);
    case 6: return launch_detect<6>(a, t, st);
    case 7: return launch_detect<7>(a, t, st);
    case 8: return launch_detect<8>(a, t, st);
    case 9: return launch_detect<9>(a, t, st);
    case 10: return launch_detect<10>(a, t, st);
    case 11: return launch_detect<11>(a, t, st);
    case 12: return launch_detect<12>(a, t, st);
    case 13: return launch_detect<13>(a, t, st);
    case 14: return launch_detect<14>(a, t, st);
    case 15: return launch_detect<15>(a, t, st);
    case 16: return launch_detect<16>(a, t, st);
    case 17: return launch_detect<17>(a, t, st);
    case 18: return launch_detect<18>(a, t, st);
    case 19: return launch_detect<19>(a, t, st);
    case 20: return launch_detect<20>(a, t, st);
    case 21: return launch_detect<21>(a, t, st);
    case 22: return launch_detect<22>(a, t, st);
    case 23: return launch_detect<23>(a, t, st);
    case 24: return launch_detect<24>(a, t, st);
    case 25: return launch_detect<25>(a, t, st);
    case 26: return launch_detect<26>(a, t, st);
    case 27: return launch_detect<27>(a, t, st);
    case 28: return launch_detect<28>(a, t, st);
    case 29: return launch_detect<29>(a, t, st);
    case 30: return launch_detect<30>(a, t, st);
    case 31: return launch_detect<31>(a, t, st);
    case 32: return launch_detect<32>(a, t, st);
    case 33: return launch_detect<33>(a, t, st);
    case 34: return launch_detect<34>(a, t, st);
    case 35: return launch_detect<35>(a, t, st);
    case 36: return launch_detect<36>(a, t, st);
    case 37: return launch_detect<37>(a, t, st);
    case 38: return launch_detect<38>(a, t, st);
    case 39: return launch_detect<39>(a, t, st);
    case 40: return launch_detect<40>(a, t, st);
    case 41: return launch_detect<41>(a, t, st);
    case 42: return launch_detect<42>(a, t, st);
    case 43: return launch_detect<43>(a, t, st);
    case 44: return launch_detect<44>(a, t, st);
    case 45: return launch_detect<45>(a, t, st);
    case 46: return launch_detect<46>(a, t, st);
    case 47: return launch_detect<47>(a, t, st);
    case 48: return launch_detect<48>(a, t, st);
    default: break;
  }
  // 49..256 table rows: the general detection (exponents of 32 energies in registers, run-time material loop)
  const size_t n_rays = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  const size_t nblk = (n_rays + 255) / 256;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  hipLaunchKernelGGL(detect_kernel_chunked, dim3((unsigned)nblk), dim3(256), 0, st, a, t.mu, t.w, t.w2);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

}  // namespace dexct
