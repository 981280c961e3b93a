// Shared by the traversal kernels (siddon.hip, siddon_packed.hip): launch arguments, ray indexing and the
// polychromatic detection of the per-material path lengths (internal; the public surface is include/dexct.h).
#pragma once
#include "common.h"

namespace dexct {

constexpr int kBlock = 256;

// crossing-slab record of the row-parallel kernels' LDS lists
struct CrossRec {
  uint32_t offa, offb;   // column offsets of the two pieces
  float t;
  uint32_t pad;          // 16-B records (one ds_read_b128)
};
constexpr float kLog2e = 1.44269504088896340736f;

struct ProjArgs {
  dexct_fan_geom g;
  const dexct_ray_plan* plan;
  const uint8_t* vol_yx;
  const uint8_t* vol_xy;
  const uint8_t* vol_zf;
  int n_local_views;
  int n_materials, n_energies, n_spectra;
  float* counts;      // [S][ray]
  float* pathlen;     // optional [ray][M]
  float* variance;    // optional [S][ray]: variance of the detected signal (compound Poisson), needs w2
  int view_tile;      // views per locality tile of the row-parallel kernels
  int layout;         // 0: ray = (v*rows + r)*channels + c   1: ray = (v*channels + c)*rows + r
  // material-group mode of rows4_kernel (more than 4 materials): the volume holds codes 0..3 of one group of
  // three materials; raw accumulators (units of u) go to acc_out[(mat_base + code)*n_rays + ray], no detection
  float* acc_out;
  int mat_base;
};

// The attenuation and weight tables are passed as DIRECT __restrict__ kernel arguments (not inside
// ProjArgs): only then does the compiler know they are read-only and wave-uniform and fetch them with
// s_load through the scalar cache; as struct members they were fetched with 670 vector loads per wave.
struct Tables {
  const float* __restrict__ mu;   // [M][nE] linear attenuation [1/cm]
  const float* __restrict__ w;    // [S][nE]
  const float* __restrict__ w2;   // [S][nE] w * signal per photon (variance weights), or null
};

__device__ __forceinline__ size_t ray_index(const ProjArgs& a, int v, int r, int c) {
  return a.layout == 0 ? ((size_t)v * a.g.n_rows + r) * a.g.n_channels + c
                       : ((size_t)v * a.g.n_channels + c) * a.g.n_rows + r;
}

// sum_e w[s][e] * 2^(-sum_m mu[m][e] * L2[q][m]) for R rays and SLOTS spectrum slots.  Rays are handled in pairs
// (float2 -> v_pk_fma_f32, the scalar table value broadcast to both halves); each half performs exactly the
// scalar sequence of fmaf's, so the sums do not depend on the pairing.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NM, int R, int SLOTS>
__device__ __forceinline__ void detect_energies(const f32x2 (&Lp)[(R + 1) / 2][NM], const float* __restrict__ mu,
                                                const float* __restrict__ w, int n_e,
                                                const int (&srow)[DEXCT_MAX_SPECTRA],
                                                float (&acc)[DEXCT_MAX_SPECTRA][R]) {
  constexpr int P = (R + 1) / 2;                      // pairs; an odd last ray rides alone in a pair's low half
  f32x2 ap[SLOTS][P];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s)
#pragma unroll
    for (int j = 0; j < P; ++j) ap[s][j] = f32x2{0.0f, 0.0f};
  auto one_energy = [&](int e) {
    f32x2 pe[P];
#pragma unroll
    for (int j = 0; j < P; ++j) pe[j] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const float mue = mu[m * n_e + e];
#pragma unroll
      for (int j = 0; j < P; ++j) pe[j] = __builtin_elementwise_fma(f32x2{mue, mue}, Lp[j][m], pe[j]);
    }
    f32x2 te[P];
#pragma unroll
    for (int j = 0; j < P; ++j) te[j] = f32x2{__builtin_amdgcn_exp2f(-pe[j].x), __builtin_amdgcn_exp2f(-pe[j].y)};
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const float ws = w[srow[s] + e];
#pragma unroll
      for (int j = 0; j < P; ++j) ap[s][j] = __builtin_elementwise_fma(f32x2{ws, ws}, te[j], ap[s][j]);
    }
  };
  int e = 0;
  for (; e + 4 <= n_e; e += 4) {
    one_energy(e);
    one_energy(e + 1);
    one_energy(e + 2);
    one_energy(e + 3);
  }
  for (; e < n_e; ++e) one_energy(e);
#pragma unroll
  for (int s = 0; s < SLOTS; ++s)
#pragma unroll
    for (int q = 0; q < R; ++q) acc[s][q] = (q & 1) ? ap[s][q / 2].y : ap[s][q / 2].x;
}

// counts[s] = sum_e w[s][e] * exp(-sum_m mu[m][e] * L[m]) (v_exp_f32 on the log2(e)-scaled exponent)
// for R rays at once (R = 4 in rows4_kernel: one scalar table load serves 4 rays and the FMAs pair up
// into v_pk_fma_f32).  mu and w are wave-uniform (scalar loads); NM materials in registers.
template <int NM, int R>
__device__ __forceinline__ void detect_store(const float (&L)[R][NM], const ProjArgs& a, const float* __restrict__ mu,
                                             const float* __restrict__ w, const float* __restrict__ w2,
                                             const size_t (&ray)[R], const bool (&valid)[R]) {
  const int n_e = a.n_energies;
  const size_t sstride = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  if (a.pathlen) {
#pragma unroll
    for (int q = 0; q < R; ++q)
      if (valid[q]) {
#pragma unroll
        for (int m = 0; m < NM; ++m) a.pathlen[ray[q] * NM + m] = L[q][m];
      }
  }
  // exponent in base 2: scale the lengths once instead of every exponent
  float L2[R][NM];
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int m = 0; m < NM; ++m) L2[q][m] = L[q][m] * kLog2e;
  float acc[DEXCT_MAX_SPECTRA][R];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
#pragma unroll
    for (int q = 0; q < R; ++q) acc[s][q] = 0.0f;
  // Branch-free body: unused spectrum slots read row 0 of w again (their sums are never stored), so all table
  // loads of an energy are independent scalar loads the compiler can issue together; 4 energies per trip.  The
  // kernel's time is its vector-instruction count (the detection is half of it), so the body exists for 2 slots
  // (one or two spectra - the dual-energy scan) and for 4, and pairs of rays go through v_pk_fma_f32.
  int srow[DEXCT_MAX_SPECTRA];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) srow[s] = (s < a.n_spectra ? s : 0) * n_e;
  // (the pairs are formed here, not in the callee: handing the float array over by reference left it in scratch
  // memory for NM = 2 and 4)
  f32x2 Lp[(R + 1) / 2][NM];
#pragma unroll
  for (int j = 0; j < (R + 1) / 2; ++j)
#pragma unroll
    for (int m = 0; m < NM; ++m) Lp[j][m] = f32x2{L2[2 * j][m], L2[2 * j + 1 < R ? 2 * j + 1 : 2 * j][m]};
  if (a.n_spectra <= 2)
    detect_energies<NM, R, 2>(Lp, mu, w, n_e, srow, acc);
  else
    detect_energies<NM, R, DEXCT_MAX_SPECTRA>(Lp, mu, w, n_e, srow, acc);
  if (a.variance) {
    // second pass, only when noise is requested: var_s = sum_e w2[s][e] * exp(-P_e), w2 = w * (signal per photon)
    float var[DEXCT_MAX_SPECTRA][R];
#pragma unroll
    for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
#pragma unroll
      for (int q = 0; q < R; ++q) var[s][q] = 0.0f;
    for (int e = 0; e < n_e; ++e) {
      float pe[R];
#pragma unroll
      for (int q = 0; q < R; ++q) pe[q] = 0.0f;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const float mue = mu[m * n_e + e];
#pragma unroll
        for (int q = 0; q < R; ++q) pe[q] = fmaf(mue, L2[q][m], pe[q]);
      }
#pragma unroll
      for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
        if (s < a.n_spectra) {
          const float ws = w2[s * n_e + e];
#pragma unroll
          for (int q = 0; q < R; ++q) var[s][q] = fmaf(ws, __builtin_amdgcn_exp2f(-pe[q]), var[s][q]);
        }
    }
#pragma unroll
    for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
      if (s < a.n_spectra) {
#pragma unroll
        for (int q = 0; q < R; ++q)
          if (valid[q]) a.variance[ray[q] + s * sstride] = var[s][q];
      }
  }
  // 4 consecutive rays (layout 1, rows4_kernel): one 16-byte store per spectrum
  const bool vec4 = R == 4 && a.layout == 1 && (a.g.n_rows & 3) == 0 && valid[R - 1];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
    if (s < a.n_spectra) {
      if (vec4) {
        *reinterpret_cast<float4*>(a.counts + ray[0] + s * sstride) =
            make_float4(acc[s][0], acc[s][R > 1 ? 1 : 0], acc[s][R > 2 ? 2 : 0], acc[s][R > 3 ? 3 : 0]);
      } else {
#pragma unroll
        for (int q = 0; q < R; ++q)
          if (valid[q]) a.counts[ray[q] + s * sstride] = acc[s][q];
      }
    }
}

template <int NM>
__device__ __forceinline__ void detect_store1(const float (&L)[NM], const ProjArgs& a, const float* __restrict__ mu,
                                              const float* __restrict__ w, const float* __restrict__ w2, size_t ray) {
  float L1[1][NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) L1[0][m] = L[m];
  const size_t rays[1] = {ray};
  const bool valid[1] = {true};
  detect_store<NM, 1>(L1, a, mu, w, w2, rays, valid);
}


}  // namespace dexct
