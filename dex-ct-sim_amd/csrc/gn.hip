// Per-detector-pixel Newton basis-material decomposition for gfx950.
//
// Replaces optimize_sino_cpu (matdecomp.py:87-127 of the reference): every pixel is an independent
// 2-unknown minimisation of the Poisson negative log-likelihood; per iteration it needs, for both
// measurements k, the sums over energy of {1, mu0, mu1, mu0^2, mu0*mu1, mu1^2} * i0_k(e) *
// exp(-(a0*mu0(e) + a1*mu1(e))), then a closed-form 2x2 solve (:122-125).  Fixed iteration count,
// start 1e-6 (:98-99), exponent clip +-700 (:116), full Newton step incl. the (g/nu - 1) * hessian
// term (:123), no damping, exactly as the reference.
//
// Mapping: one thread per pixel, all iterations in registers.  The 14 per-energy table values
// (two attenuations + 2 x 6 products, with the reference's rounding of ssff / ssff2, :102,:105)
// are built once by gn_tables_kernel into a small workspace and read in the hot loop through the
// scalar cache: the energy index is wave-uniform, so every table value is an SGPR operand of a
// v_fma_f64 and costs neither LDS bandwidth nor VGPRs.  The only per-lane lookup is the 2048-entry
// 2^(j/2048) table of the exponential, which sits in LDS (16 KB).
// There is no dense contraction: the kernel is bound by the FP64 vector rate (FMA + exp), not by
// HBM (8 B read + 16 B written per pixel); per energy-iteration it issues 25 FP64 instructions
// (2 exponent, 2 clip, 9 exp, 12 accumulate).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace dexct {

// States kept for the exact repeated-state exit of the float64 Newton loop (cycles up to kGnHistory + 1).
constexpr int kGnHistory = 8;
constexpr int kGnRingDefault = 0;    // DEXCT_GN_RING=1: history of the lane-refill kernel as a ring (fewer vector instructions, more spills: measured equal, profiles/r03_gn_isa.md)

constexpr int kGnBlock = 256;
constexpr int kTab = 14;  // -mu0 K, -mu1 K (K = 2048/ln2), then per k: i0, i0*mu0, i0*mu1, i0*mu0^2, i0*mu0*mu1, i0*mu1^2

// Results leave with the non-temporal hint (round 3): written once, never read by the kernel - left to the default policy
// the 6.5 GB of them push the spill scratch of the resident waves out of the L2 (WRITE_SIZE 34 GB for 6.5 GB of results,
// tools/probes/gn_write.py).  The bits written are the same.
__device__ __forceinline__ void store_a(double* __restrict__ out_a, int64_t p, double a0, double a1) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  __builtin_nontemporal_store(d2{a0, a1}, reinterpret_cast<d2*>(out_a + 2 * p));
}

template <typename T>
__device__ __forceinline__ T load_g(const void* p, int is_f64, int64_t i) {
  return is_f64 ? (T) reinterpret_cast<const double*>(p)[i] : (T) reinterpret_cast<const float*>(p)[i];
}

// exp(x) for |x| <= 700, given y = x * 2048/ln2 (the tables hold -mu * 2048/ln2, so y costs the same two
// instructions as x would): y = n + f with n = rint(y) = 2048 k + j, |f| <= 1/2, and
// exp(x) = 2^k * 2^(j/2048) * e^(f ln2/2048), a cubic in f for e^r - 1 (truncation r^4/24 < 4e-17, r = f ln2/2048).
// n comes out of the low mantissa bits of y + 1.5 * 2^52 (round to nearest even, like rint); f = y - n is exact.
// About 1 ulp.
constexpr int kPowBits = 11;
constexpr int kPowN = 1 << kPowBits;
constexpr double kExpScale = 0x1.71547652b82fep+11;          // 2048 / ln 2
constexpr double kExpClip = 700.0 * kExpScale;               // the reference's clip of the exponent (matdecomp.py:116)
template <bool IEXP = false>
__device__ __forceinline__ double exp_tab(double y, const double* __restrict__ lds_pow) {
  const double kMagic = 6755399441055744.0;   // 1.5 * 2^52
  constexpr double c1 = 0x1.62e42fefa39efp-12;               // ln2 / 2048
  constexpr double c2 = c1 * c1 / 2.0, c3 = c1 * c1 * c1 / 6.0;
  const double tm = y + kMagic;
  const int ni = __double2loint(tm);
  const double f = y - (tm - kMagic);
  double q = fma(f, c3, c2);
  q = fma(f, q, c1);
  const double p = f * q;
  const double tj = lds_pow[ni & (kPowN - 1)];
  const double t = fma(tj, p, tj);
  if (!IEXP) return ldexp(t, ni >> kPowBits);
  // IEXP (A/B variant): 2^k by adding k to the exponent field - two 32-bit integer operations on the high dword
  // instead of a shift and v_ldexp_f64.  Exact here: t = tj (1 + p) lies in [1, 2 + 4e-4) and |k| <= 1010 (the
  // reference's clip at 700 bounds the exponent), so t 2^k is a normal number and ldexp performs the same exponent
  // addition; a NaN argument has ni = 0 and stays the NaN it is.
  const int hi = __double2hiint(t) + ((ni & ~(kPowN - 1)) << (20 - kPowBits));
  return __hiloint2double(hi, __double2loint(t));
}

// 1 / x by v_rcp_f64 and two Newton refinements (what a float64 division starts with, without its scaling and
// fix-up steps: the operands here - expected counts, the Hessian's determinant - are far from the subnormal
// range).  For x = 0, +-inf or NaN the refinement would turn the hardware's answer (inf, 0, NaN - what IEEE
// division gives) into NaN, so it is skipped there: a sum that overflowed during a wild transient then behaves
// as in the reference (g / inf = 0) and the pixel can still recover.
__device__ __forceinline__ double rcp_f64(double x) {
  const double r0 = __builtin_amdgcn_rcp(x);
  double r = fma(r0, fma(-x, r0, 1.0), r0);
  r = fma(r, fma(-x, r, 1.0), r);
  const double ax = fabs(x);
  return (ax > 0.0 && ax < __builtin_huge_val()) ? r : r0;
}

// Energies are sorted by gn_tables_kernel into three classes: both spectra have weight (nA), only spectrum 0
// (nB), only spectrum 1 (nC); energies no spectrum weights are dropped.  A zero weight contributes exactly
// 0 to every sum (the attenuation factor is finite thanks to the clip), so skipping those FMAs changes no
// term of the reference's sums - only their order.
template <int KSEL, bool CLIP, bool IEXP = false>   // KSEL 0: both measurements, 1: only k = 0, 2: only k = 1; CLIP: apply the +-700 clip
__device__ __forceinline__ void energy_sums_f64(const double* __restrict__ tab, const double* __restrict__ lds_pow,
                                                int e0, int e1, double a0, double a1, double (&nu)[2], double (&nuo)[2],
                                                double (&G0)[2], double (&G1)[2], double (&H00)[2], double (&H01)[2],
                                                double (&H11)[2]) {
  // The expected counts nu are summed in TWO partial sums (every other energy of a run into nuo, joined by the
  // caller): the residual g / nu - 1 cancels to rounding level at the solution, so the rounding of nu is what moves the
  // iterate in its last bits, and halving the length of each running sum shortens that wandering - the exact
  // repeated-state exit then fires after 28.6 instead of 31.8 iterations on average (benchmark sinograms), at no
  // instruction per energy.  (A compensated sum of nu was tried: no further gain, 25.3 vs 24.7 with the residual below.)
  auto one = [&](int e, double (&nuk)[2]) {
    const double* __restrict__ t = tab + e * kTab;   // wave-uniform: scalar loads
    double y = fma(a1, t[1], a0 * t[0]);               // t[0], t[1] = -mu0, -mu1 times 2048/ln2
    if (CLIP) y = fmin(fmax(y, -kExpClip), kExpClip);
    const double at = exp_tab<IEXP>(y, lds_pow);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if ((KSEL == 1 && k == 1) || (KSEL == 2 && k == 0)) continue;
      const double* __restrict__ tk = t + 2 + 6 * k;
      nuk[k] = fma(tk[0], at, nuk[k]);
      G0[k] = fma(tk[1], at, G0[k]);
      G1[k] = fma(tk[2], at, G1[k]);
      H00[k] = fma(tk[3], at, H00[k]);
      H01[k] = fma(tk[4], at, H01[k]);
      H11[k] = fma(tk[5], at, H11[k]);
    }
  };
  int e = e0;
  for (; e + 2 <= e1; e += 2) {
    one(e, nu);
    one(e + 1, nuo);
  }
  if (e < e1) one(e, nu);
}

// Each class is stored as [energies that always need the clip (large mu) | energies whose exponent is provably
// within +-700 whenever |a0| * mu0_free_max + |a1| * mu1_free_max <= 699.9 (the margin covers the rounding of the
// scaled tables)]: for the second part the clip of matdecomp.py:116 is the identity and is skipped (two FP64
// instructions per energy), bit for bit the same result.
struct EnergyClasses { int nA, nAc, nB, nBc, nC, nCc; double m0_free, m1_free; };

template <bool IEXP = false>
__device__ __forceinline__ void newton_step_f64(const double* __restrict__ tab, const double* __restrict__ lds_pow,
                                                EnergyClasses ec, double g0, double g1, double& a0, double& a1) {
  double nu[2] = {0, 0}, nuo[2] = {0, 0}, G0[2] = {0, 0}, G1[2] = {0, 0}, H00[2] = {0, 0}, H01[2] = {0, 0}, H11[2] = {0, 0};
  const int bA = 0, bB = ec.nA, bC = ec.nA + ec.nB;
  // the always-clipped heads of the three classes
  energy_sums_f64<0, true, IEXP>(tab, lds_pow, bA, bA + ec.nAc, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
  energy_sums_f64<1, true, IEXP>(tab, lds_pow, bB, bB + ec.nBc, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
  energy_sums_f64<2, true, IEXP>(tab, lds_pow, bC, bC + ec.nCc, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
  // the tails: clip-free when the bound holds for this pixel (NaN compares false -> clipped path)
  if (fma(fabs(a1), ec.m1_free, fabs(a0) * ec.m0_free) <= 699.9) {
    energy_sums_f64<0, false, IEXP>(tab, lds_pow, bA + ec.nAc, bA + ec.nA, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
    energy_sums_f64<1, false, IEXP>(tab, lds_pow, bB + ec.nBc, bB + ec.nB, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
    energy_sums_f64<2, false, IEXP>(tab, lds_pow, bC + ec.nCc, bC + ec.nC, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
  } else {
    energy_sums_f64<0, true, IEXP>(tab, lds_pow, bA + ec.nAc, bA + ec.nA, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
    energy_sums_f64<1, true, IEXP>(tab, lds_pow, bB + ec.nBc, bB + ec.nB, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
    energy_sums_f64<2, true, IEXP>(tab, lds_pow, bC + ec.nCc, bC + ec.nC, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
  }
  nu[0] += nuo[0];
  nu[1] += nuo[1];
  const double g[2] = {g0, g1};
  double c[2], q[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double inv = rcp_f64(nu[k]), ratio = g[k] * inv;
    // g / nu - 1 (matdecomp.py:122) as (g - nu) / nu: near the solution the subtraction is exact, so the residual
    // carries the rounding of nu only, not an extra half ulp of 1 from the quotient - the last-bit wandering of the
    // iterate is shorter again (mean executed iterations 28.6 -> 24.7).  An overflowed nu (inf) keeps the reference's
    // value: g / inf - 1 = -1.
    c[k] = fabs(nu[k]) < __builtin_huge_val() ? (g[k] - nu[k]) * inv : ratio - 1.0;
    q[k] = ratio * inv;                                        // g / nu^2 (matdecomp.py:123)
  }
  // the sums over the two measurements, first term + second term (a running sum started at 0 costs an addition of 0
  // per sum that the compiler may not drop: 0 + (-0) is +0)
  const double dF0 = c[0] * G0[0] + c[1] * G0[1];
  const double dF1 = c[0] * G1[0] + c[1] * G1[1];
  const double h00 = (q[0] * (G0[0] * G0[0]) - c[0] * H00[0]) + (q[1] * (G0[1] * G0[1]) - c[1] * H00[1]);
  const double h01 = (q[0] * (G0[0] * G1[0]) - c[0] * H01[0]) + (q[1] * (G0[1] * G1[1]) - c[1] * H01[1]);
  const double h11 = (q[0] * (G1[0] * G1[0]) - c[0] * H11[0]) + (q[1] * (G1[1] * G1[1]) - c[1] * H11[1]);
  const double inv_det = rcp_f64(h00 * h11 - h01 * h01);
  a0 -= (h11 * dF0 - h01 * dF1) * inv_det;
  a1 -= (h00 * dF1 - h01 * dF0) * inv_det;
}

// float32 table layout per energy: [mu0*log2e, mu1*log2e, then for c in {1, mu0, mu1, mu0^2, mu0mu1, mu1^2}:
// (i0_0 * c, i0_1 * c)] - the two measurements of one product adjacent, so that each SGPR pair feeds one
// v_pk_fma_f32 directly.
template <int KSEL>
__device__ __forceinline__ void energy_sums_f32(const float* __restrict__ tab, int e0, int e1, float a0, float a1,
                                                float (&acc)[6][2]) {
#pragma unroll 2
  for (int e = e0; e < e1; ++e) {
    const float* __restrict__ t = tab + e * kTab;
    float x = -fmaf(a1, t[1], a0 * t[0]);
    x = fminf(fmaxf(x, -120.0f), 120.0f);
    const float at = __builtin_amdgcn_exp2f(x);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      if (KSEL != 2) acc[c][0] = fmaf(t[2 + 2 * c], at, acc[c][0]);
      if (KSEL != 1) acc[c][1] = fmaf(t[3 + 2 * c], at, acc[c][1]);
    }
  }
}

__device__ __forceinline__ void newton_step_f32(const float* __restrict__ tab, EnergyClasses ec, float g0, float g1,
                                                float& a0, float& a1) {
  float acc[6][2];
#pragma unroll
  for (int c = 0; c < 6; ++c) acc[c][0] = acc[c][1] = 0.0f;
  energy_sums_f32<0>(tab, 0, ec.nA, a0, a1, acc);
  energy_sums_f32<1>(tab, ec.nA, ec.nA + ec.nB, a0, a1, acc);
  energy_sums_f32<2>(tab, ec.nA + ec.nB, ec.nA + ec.nB + ec.nC, a0, a1, acc);
  const float g[2] = {g0, g1};
  float dF0 = 0, dF1 = 0, h00 = 0, h01 = 0, h11 = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float nu = acc[0][k], G0 = acc[1][k], G1 = acc[2][k];
    const float c = g[k] / nu - 1.0f, q = g[k] / (nu * nu);
    dF0 += c * G0;
    dF1 += c * G1;
    h00 += q * (G0 * G0) - c * acc[3][k];
    h01 += q * (G0 * G1) - c * acc[4][k];
    h11 += q * (G1 * G1) - c * acc[5][k];
  }
  const float det = h00 * h11 - h01 * h01;
  a0 -= (h11 * dF0 - h01 * dF1) / det;
  a1 -= (h00 * dF1 - h01 * dF0) / det;
}

// Workspace layout (doubles): [0] = scale of the float32 tables, [1..3] = nA, nB, nC (energy classes), [4..6] =
// how many of each class come first and always need the clip, [7..8] = max mu0 / mu1 over the clip-free parts,
// [9] = (uint64, diagnostic) pixel-iterations the last gn_refill_kernel launch on this workspace executed,
// [10] = (uint64, progress) pixels that launch has finished so far (added wave by wave while it runs; with the run queue:
// pixels handed out so far), [11] = (uint64) head of the run queue,
// pad to 16, then [n_e][14] float64, then [n_e][14] float32, then n_e ints (the permutation).
constexpr int kWsHeader = 16;

// One block per spectrum row ("bin": 1 for the channel-independent case, else one per detector channel).
// i0 is [2][n_bins][n_e].  With several bins the energies are NOT sorted into classes (they may differ
// per bin): class A then simply holds all n_e energies.
__global__ __launch_bounds__(256) void gn_tables_kernel(const double* __restrict__ i0, const double* __restrict__ mus,
                                                        int n_e, int n_bins, double* __restrict__ ws) {
  // float32 tables are scaled by one power of two common to both measurements (the Newton step is
  // invariant under a common scaling of counts and spectra) so that sums stay near 1.
  __shared__ double s_scale;
  __shared__ int s_n[3], s_nc[3];
  const int bin = blockIdx.x;
  const double* __restrict__ i00 = i0 + (size_t)bin * n_e;                       // k = 0
  const double* __restrict__ i01 = i0 + ((size_t)n_bins + bin) * n_e;            // k = 1
  double* tab = ws + kWsHeader + (size_t)bin * n_e * kTab;
  float* tab32 = reinterpret_cast<float*>(ws + kWsHeader + (size_t)n_bins * n_e * kTab);   // only for n_bins == 1
  int* perm = reinterpret_cast<int*>(tab32 + (size_t)n_e * kTab) + (size_t)bin * n_e;
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int e = 0; e < n_e; ++e) { s0 += i00[e]; s1 += i01[e]; }
    int ex = 0;
    frexp(fmax(s0, s1), &ex);
    s_scale = ldexp(1.0, -ex);
    int n = 0;
    double m0f = 0.0, m1f = 0.0;
    if (n_bins == 1) {
      // stable partition of the energies into the classes A (both), B (only 0), C (only 1), each split into
      // [large attenuation: clip always | the rest]; kMuFree = 4 cm^2/g keeps the bound valid up to
      // |a0| + |a1| = 175 g/cm^2, far beyond any physical ray
      const double kMuFree = 4.0;
      for (int cls = 0; cls < 3; ++cls) {
        for (int part = 0; part < 2; ++part) {
          int cnt = 0;
          for (int e = 0; e < n_e; ++e) {
            const bool z0 = i00[e] == 0.0, z1 = i01[e] == 0.0;
            const int c = (!z0 && !z1) ? 0 : (!z0 ? 1 : (!z1 ? 2 : 3));
            const bool big = !(fmax(fabs(mus[e]), fabs(mus[n_e + e])) <= kMuFree);
            if (c == cls && big == (part == 0)) {
              perm[n++] = e;
              ++cnt;
              if (!big) { m0f = fmax(m0f, fabs(mus[e])); m1f = fmax(m1f, fabs(mus[n_e + e])); }
            }
          }
          if (part == 0) s_nc[cls] = cnt; else s_n[cls] = s_nc[cls] + cnt;
        }
      }
    } else {
      for (int e = 0; e < n_e; ++e) perm[e] = e;
      s_n[0] = n_e; s_n[1] = 0; s_n[2] = 0;
      s_nc[0] = n_e; s_nc[1] = 0; s_nc[2] = 0;      // channel-dependent spectra: everything clipped
    }
    if (bin == 0) {
      ws[0] = s_scale;
      ws[1] = (double)s_n[0];
      ws[2] = (double)s_n[1];
      ws[3] = (double)s_n[2];
      ws[4] = (double)s_nc[0];
      ws[5] = (double)s_nc[1];
      ws[6] = (double)s_nc[2];
      ws[7] = m0f;
      ws[8] = m1f;
      reinterpret_cast<unsigned long long*>(ws)[9] = 0ull;      // executed pixel-iterations, counted by gn_refill_kernel
      reinterpret_cast<unsigned long long*>(ws)[10] = 0ull;     // finished pixels (progress of the running launch)
      reinterpret_cast<unsigned long long*>(ws)[11] = 0ull;     // head of the pixel-run queue of gn_refill_kernel
    }
  }
  __syncthreads();
  const double scale = s_scale;
  const int n_used = s_n[0] + s_n[1] + s_n[2];
  for (int j = threadIdx.x; j < n_used; j += blockDim.x) {
    const int e = perm[j];
    const double m0 = mus[e], m1 = mus[n_e + e];
    double* t = tab + j * kTab;
    t[0] = -m0 * kExpScale;                              // exponent in units of ln2/2048, see exp_tab
    t[1] = -m1 * kExpScale;
    const double m00 = m0 * m0, m01 = m0 * m1, m11 = m1 * m1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double w = k == 0 ? i00[e] : i01[e];
      double* tk = t + 2 + 6 * k;
      tk[0] = w;
      tk[1] = w * m0;
      tk[2] = w * m1;
      tk[3] = w * m00;
      tk[4] = w * m01;
      tk[5] = w * m11;
    }
    if (n_bins == 1) {
      float* f = tab32 + j * kTab;
      f[0] = (float)(m0 * 1.4426950408889634);
      f[1] = (float)(m1 * 1.4426950408889634);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        f[2 + 2 * c] = (float)(t[2 + c] * scale);
        f[3 + 2 * c] = (float)(t[8 + c] * scale);
      }
    }
  }
}

// MIXED: n_iters - n_polish iterations in float32, then n_polish in float64.
// PER_BIN: pixel p uses the tables of bin (p / bin_div) % n_bins (channel-dependent spectra, the general
// signature of optimize_sino_cpu); the table pointer is then per lane and the values arrive by vector loads.
template <bool MIXED, bool PER_BIN>
__global__ __launch_bounds__(kGnBlock) void gn_kernel(const void* __restrict__ g1, const void* __restrict__ g2,
                                                      int g_is_f64, int64_t n_pix, const double* __restrict__ ws,
                                                      int n_e, int n_iters, int n_polish, int n_bins, int bin_div,
                                                      const double* __restrict__ mask_max, double mask_frac,
                                                      int exact_exit, double* __restrict__ out_a) {
  __shared__ double lds_pow[kPowN];     // 2^(j/2048), 16 KB
  for (int j = threadIdx.x; j < kPowN; j += kGnBlock) lds_pow[j] = exp2((double)j * (1.0 / kPowN));
  __syncthreads();
  const float* __restrict__ tab32 = reinterpret_cast<const float*>(ws + kWsHeader + (size_t)n_bins * n_e * kTab);
  const EnergyClasses ec{(int)ws[1], (int)ws[4], (int)ws[2], (int)ws[5], (int)ws[3], (int)ws[6], ws[7], ws[8]};
  const int64_t p = (int64_t)blockIdx.x * kGnBlock + threadIdx.x;
  if (p >= n_pix) return;
  const double* __restrict__ tab = ws + kWsHeader;
  if (PER_BIN) tab += (size_t)((p / bin_div) % n_bins) * n_e * kTab;
  const double gd0 = load_g<double>(g1, g_is_f64, p), gd1 = load_g<double>(g2, g_is_f64, p);
  // Fused air mask of get_basismat_sinos (matdecomp.py:195-196, :204-205): a pixel with g1 >= frac * max is set
  // to 0 afterwards whatever the iteration produced, so its iterations are not run at all.
  if (mask_max && gd0 >= mask_frac * mask_max[0]) {
    store_a(out_a, p, 0.0, 0.0);
    return;
  }
  double a0 = 1e-6, a1 = 1e-6;
  int it = 0;
  if (MIXED) {
    const double scale = ws[0];
    float fa0 = 1e-6f, fa1 = 1e-6f;
    const float fg0 = (float)(gd0 * scale), fg1 = (float)(gd1 * scale);
    const int n_bulk = n_iters > n_polish ? n_iters - n_polish : 0;
    // float32 bulk with the same exact repeated-state exit as the float64 loop: a state is the pair of floats,
    // compared as one 64-bit pattern; the state the cycle holds at iteration n_bulk is what all n_bulk
    // iterations would have produced.
    {
      auto pack = [](float x, float y) {
        return ((unsigned long long)__float_as_uint(y) << 32) | (unsigned long long)__float_as_uint(x);
      };
      unsigned long long hs[kGnHistory];
#pragma unroll
      for (int k = 0; k < kGnHistory; ++k) hs[k] = 0ull;
      int hit = -2;
      for (; it < n_bulk; ++it) {
        float n0 = fa0, n1 = fa1;
        newton_step_f32(tab32, ec, fg0, fg1, n0, n1);
        if (exact_exit) {
          const unsigned long long b = pack(n0, n1), c = pack(fa0, fa1);
          if (b == c) hit = -1;
#pragma unroll
          for (int k = kGnHistory - 1; k >= 0; --k)
            if (k < it && b == hs[k] && hit != -1) hit = k;
          if (hit != -2) break;
#pragma unroll
          for (int k = kGnHistory - 1; k > 0; --k) hs[k] = hs[k - 1];
          hs[0] = c;
        }
        fa0 = n0;
        fa1 = n1;
      }
      if (hit >= 0) {
        const int base = it - 1 - hit, period = hit + 2;
        const int slot = hit - (n_bulk - base) % period;        // -1 selects the current state
#pragma unroll
        for (int k = 0; k < kGnHistory; ++k)
          if (slot == k) { fa0 = __uint_as_float((unsigned)hs[k]); fa1 = __uint_as_float((unsigned)(hs[k] >> 32)); }
      }
      it = n_bulk;
    }
    if (n_bulk > 0) { a0 = (double)fa0; a1 = (double)fa1; }
    // float64 polish; a pixel whose float32 trajectory did not arrive (non-finite, or the polish
    // steps are still moving it) is redone from the start in float64, i.e. in the reference's
    // arithmetic.  Divergent, but rare on well-posed data.
    double p0 = a0, p1 = a1;
    for (; it < n_iters; ++it) {
      p0 = a0; p1 = a1;
      newton_step_f64(tab, lds_pow, ec, gd0, gd1, a0, a1);
    }
    const double moved = fmax(fabs(a0 - p0), fabs(a1 - p1));
    const double size = fmax(fmax(fabs(a0), fabs(a1)), 1.0);
    if (n_bulk > 0 && !(moved <= 1e-9 * size)) {
      a0 = 1e-6; a1 = 1e-6;
      for (it = 0; it < n_iters; ++it) newton_step_f64(tab, lds_pow, ec, gd0, gd1, a0, a1);
    }
  }
  // The Newton update is a deterministic function of the two doubles (a0, a1).  Once an iterate repeats bit for
  // bit - a fixed point, or a short cycle in the last ulps - every later iterate is known without computing
  // it, so the loop may stop there and still return exactly what n_iters iterations would have produced.
  // hist[k] holds the state s_{it-1-k}; a match s_{it+1} == s_{it-1-k} makes s_m periodic with period k + 2 for
  // m >= it-1-k, and the answer s_{n_iters} is read from the history.  (An entry is only compared once it has
  // been written; NaN states repeat or not like any other bit pattern, which is still exact.)
  long long h0[kGnHistory], h1[kGnHistory];
#pragma unroll
  for (int k = 0; k < kGnHistory; ++k) { h0[k] = 0; h1[k] = 0; }
  int hit = -2;                                              // -2 none, -1 fixed point, k >= 0 history slot
  for (; it < n_iters; ++it) {
    double n0 = a0, n1 = a1;
    newton_step_f64(tab, lds_pow, ec, gd0, gd1, n0, n1);
    if (exact_exit) {
      const long long b0 = __double_as_longlong(n0), b1 = __double_as_longlong(n1);
      if (b0 == __double_as_longlong(a0) && b1 == __double_as_longlong(a1)) hit = -1;
#pragma unroll
      for (int k = kGnHistory - 1; k >= 0; --k)              // descending, so the smallest period wins
        if (k < it && b0 == h0[k] && b1 == h1[k] && hit != -1) hit = k;
      if (hit != -2) break;
#pragma unroll
      for (int k = kGnHistory - 1; k > 0; --k) { h0[k] = h0[k - 1]; h1[k] = h1[k - 1]; }
      h0[0] = __double_as_longlong(a0);
      h1[0] = __double_as_longlong(a1);
    }
    a0 = n0;
    a1 = n1;
  }
  if (hit >= 0) {
    // s_m = s_{base + ((m - base) mod period)} for m >= base = it-1-hit; s_{base+j} is hist[hit-j], s_it is (a0, a1)
    const int base = it - 1 - hit, period = hit + 2;
    const int slot = hit - (n_iters - base) % period;        // -1 selects the current state
#pragma unroll
    for (int k = 0; k < kGnHistory; ++k)
      if (slot == k) { a0 = __longlong_as_double(h0[k]); a1 = __longlong_as_double(h1[k]); }
  }
  store_a(out_a, p, a0, a1);
}


// The history of the repeated-state exit as a RING whose write position is wave-uniform.  In the lane-refill kernel all
// lanes of a wave take their Newton steps together, so "the state k + 1 steps ago" sits in the same ring slot for every
// lane - (pos - 1 - k) mod 8 with pos = the wave's step counter mod 8 - whatever iteration each lane's own pixel is at
// (entries older than the lane's pixel are excluded by k < it, as before).  With pos a template parameter every index
// is static: the 32 selects that moved the history down one place per step (and the 32 copies back) become one
// 4-register write.  Same states compared, same slot chosen: bit-identical results.
template <int POS>
struct GnRing {
  static constexpr int slot_of(int k) { return ((POS - 1 - k) % kGnHistory + kGnHistory) % kGnHistory; }
  // -2: no repeat, -1: fixed point, k >= 0: equal to the state k + 1 steps before the current one (smallest k)
  static __device__ __forceinline__ int hit(const long long (&h0)[kGnHistory], const long long (&h1)[kGnHistory],
                                            long long b0, long long b1, int it, bool fixed) {
    int hit = fixed ? -1 : -2;
#pragma unroll
    for (int k = kGnHistory - 1; k >= 0; --k)
      if (k < it && b0 == h0[slot_of(k)] && b1 == h1[slot_of(k)] && hit != -1) hit = k;
    return hit;
  }
  static __device__ __forceinline__ void pick(const long long (&h0)[kGnHistory], const long long (&h1)[kGnHistory], int slot,
                                              double& f0, double& f1) {
#pragma unroll
    for (int k = 0; k < kGnHistory; ++k)
      if (slot == k) { f0 = __longlong_as_double(h0[slot_of(k)]); f1 = __longlong_as_double(h1[slot_of(k)]); }
  }
  static __device__ __forceinline__ void push(long long (&h0)[kGnHistory], long long (&h1)[kGnHistory], double a0, double a1) {
    h0[POS] = __double_as_longlong(a0);
    h1[POS] = __double_as_longlong(a1);
  }
};

// float64, one shared spectrum - the benchmark's path - with lane refill.  The repeated-state exit ends pixels
// at very different iterations (from 15 to all of n_iters), and a wave is as slow as its slowest lane.  Here a
// wave owns a contiguous run of 64 * chunk pixels and every lane whose pixel has finished takes the next one of
// the run, so all 64 lanes keep iterating until the run is used up.  The energy loops stay wave-uniform (scalar
// table loads) because the tables do not depend on the pixel.  Results are bit-identical to gn_kernel's.
// HLDS (A/B variant): the four older states of the history live in an LDS ring (one 16-B slot per lane and state, written
// when a state leaves the registers) instead of 16 VGPRs.
template <int MINW, bool IEXP, bool HLDS = false, int HIST = kGnHistory, bool RING = false>      // RING: history as a ring with a wave-uniform write position (GnRing); HIST: states kept for the repeated-state exit; MINW: minimum waves per SIMD the register allocation must allow (5: 96 VGPRs, 4: 128)
__global__ __launch_bounds__(kGnBlock, MINW) void gn_refill_kernel(const void* __restrict__ g1, const void* __restrict__ g2,
                                                             int g_is_f64, int64_t n_pix, const double* __restrict__ ws,
                                                             int n_e, int n_iters, int chunk,
                                                             const double* __restrict__ mask_max, double mask_frac,
                                                             int exact_exit, double stop_tol,
                                                             double* __restrict__ out_a,
                                                             unsigned long long* __restrict__ executed,
                                                             unsigned long long* __restrict__ queue) {
  __shared__ double lds_pow[kPowN];
  __shared__ longlong2 lds_hist[HLDS ? 4 : 1][HLDS ? kGnBlock : 1];
  static_assert(!HLDS || HIST == kGnHistory, "the LDS ring holds the 4 older of 8 states");
  static_assert(!RING || (HIST == kGnHistory && !HLDS), "the register ring has kGnHistory slots");
  constexpr int kR = HLDS ? 4 : HIST;       // states kept in registers
  for (int j = threadIdx.x; j < kPowN; j += kGnBlock) lds_pow[j] = exp2((double)j * (1.0 / kPowN));
  __syncthreads();                        // the only barrier: waves leave the loop below independently
  const EnergyClasses ec{(int)ws[1], (int)ws[4], (int)ws[2], (int)ws[5], (int)ws[3], (int)ws[6], ws[7], ws[8]};
  const double* __restrict__ tab = ws + kWsHeader;
  const int64_t run = (int64_t)kWave * chunk;
  // Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8).  A sinogram is periodic in the view
  // length (air at both ends of every fan: pixels that end at once), so with "block b takes run b" an unlucky run length
  // hands some XCDs nothing but air and others nothing but object: 2.67 instead of 2.0 ns/pixel on the 2000 x 1024 scan
  // at 64 pixels per lane, +10 % on the benchmark scan at 20 or 40 (profiles/r03_kernels.md).  Each XCD therefore works
  // through a CONTIGUOUS eighth of the runs: the same mix of views for every XCD, whatever the run length.
  const uint32_t nblk_x = gridDim.x, per_x = nblk_x >> 3;
  const uint32_t lblk = (blockIdx.x < (per_x << 3)) ? (blockIdx.x & 7u) * per_x + (blockIdx.x >> 3) : blockIdx.x;
  int64_t next = ((int64_t)lblk * (kGnBlock / kWave) + (threadIdx.x >> 6)) * run;   // wave-uniform
  int64_t end = next + run < n_pix ? next + run : n_pix;
  // QUEUE (the default, `queue` != null): runs are not assigned but FETCHED - a wave whose run is used up takes the next
  // one from a global counter at once, while its other lanes are still iterating.  No lane waits for the slowest pixel of
  // "its" run any more (lanes idle only when the whole sinogram is used up), the load balances itself over CUs and XCDs,
  // and the run length stops mattering.  Pixels are independent problems: bit-identical results in any order.
  bool exhausted = false;
  if (queue) { next = 0; end = 0; }
  const bool has_mask = mask_max != nullptr;
  const double thresh = has_mask ? mask_frac * mask_max[0] : 0.0;

  int64_t p = -1;                         // this lane's pixel, -1: none
  unsigned n_exec = 0;                    // Newton steps this wave executed (< 2^32: 64 lanes x chunk x n_iters)
  double a0 = 1e-6, a1 = 1e-6, gd0 = 1.0, gd1 = 1.0;
  int it = 0;
  int ring_pos = 0;                       // RING: write position of this step, wave-uniform
  long long h0[kR], h1[kR];
#pragma unroll
  for (int k = 0; k < kR; ++k) { h0[k] = 0; h1[k] = 0; }

  for (;;) {
    unsigned long long want = __ballot(p < 0);
    while (want != 0ull) {
      if (next >= end) {
        if (!queue || exhausted) break;
        long long base = 0;
        if ((threadIdx.x & 63) == 0) base = (long long)atomicAdd(queue, (unsigned long long)run);
        base = ((long long)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
               (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_pix) { exhausted = true; break; }
        next = base;
        end = base + run < n_pix ? base + run : n_pix;
        if (executed && (threadIdx.x & 63) == 0) atomicAdd(executed + 1, (unsigned long long)(end - base));   // progress: handed out
      }
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
      const int64_t np = next + rank;
      if (p < 0 && np < end) {
        gd0 = load_g<double>(g1, g_is_f64, np);
        gd1 = load_g<double>(g2, g_is_f64, np);
        if (has_mask && gd0 >= thresh) {           // air (matdecomp.py:195-196, :204-205): 0, not iterated
          store_a(out_a, np, 0.0, 0.0);
        } else if (n_iters <= 0) {
          store_a(out_a, np, 1e-6, 1e-6);
        } else {
          p = np; a0 = 1e-6; a1 = 1e-6; it = 0;
        }
      }
      next += __popcll(want);
      if (!queue) break;                           // static runs: one serving per step, as in rounds 1-2
      want = __ballot(p < 0);                      // air pixels leave their lane wanting: it is served again at once
    }
    const unsigned long long busy = __ballot(p >= 0);
    if (busy == 0ull) {
      if (next >= end && (!queue || exhausted)) break;
      continue;
    }
    n_exec += (unsigned)__popcll(busy);                        // wave-uniform (scalar) count of Newton steps run
    double n0 = a0, n1 = a1;
    newton_step_f64<IEXP>(tab, lds_pow, ec, gd0, gd1, n0, n1);      // idle lanes repeat their last pixel's step; unused
    if constexpr (RING) {
      // ---- exit logic on the ring: one copy per write position, chosen by a scalar branch
      const long long b0 = __double_as_longlong(n0), b1 = __double_as_longlong(n1);
      const bool fixed = exact_exit && b0 == __double_as_longlong(a0) && b1 == __double_as_longlong(a1);
      bool converged = false;
      if (stop_tol > 0.0) {
        const double size = fmax(fmax(fabs(n0), fabs(n1)), 1.0);
        converged = fmax(fabs(n0 - a0), fabs(n1 - a1)) <= stop_tol * size;      // NaN compares false
      }
      int hit = fixed ? -1 : -2;
      double f0 = converged ? n0 : a0, f1 = converged ? n1 : a1;
      auto on_ring = [&](auto pos_tag) {
        using R = GnRing<decltype(pos_tag)::value>;
        if (exact_exit) hit = R::hit(h0, h1, b0, b1, it, fixed);
        if (__ballot(hit >= 0) != 0ull) {
          // the state a cycle holds at iteration n_iters (see the non-ring form below)
          int slot = -1;
          if (hit >= 0) {
            const int period = hit + 2, x = n_iters - it - 1;
            int r;
            if (n_iters < (1 << 22)) {
              const int q = (int)((float)x * __builtin_amdgcn_rcpf((float)period));
              r = x - q * period;
              r += r < 0 ? period : 0;
              r -= r >= period ? period : 0;
            } else {
              r = x % period;
            }
            slot = hit - r;
          }
          R::pick(h0, h1, slot, f0, f1);
        }
        R::push(h0, h1, a0, a1);              // unconditional: a lane that ends its pixel here never reads the history again
      };
      switch (ring_pos) {
        case 0: on_ring(std::integral_constant<int, 0>{}); break;
        case 1: on_ring(std::integral_constant<int, 1>{}); break;
        case 2: on_ring(std::integral_constant<int, 2>{}); break;
        case 3: on_ring(std::integral_constant<int, 3>{}); break;
        case 4: on_ring(std::integral_constant<int, 4>{}); break;
        case 5: on_ring(std::integral_constant<int, 5>{}); break;
        case 6: on_ring(std::integral_constant<int, 6>{}); break;
        default: on_ring(std::integral_constant<int, 7>{}); break;
      }
      ring_pos = (ring_pos + 1) & (kGnHistory - 1);
      const bool advance = hit == -2 && !converged;
      a0 = advance ? n0 : f0;
      a1 = advance ? n1 : f1;
      it += advance ? 1 : 0;
      if (p >= 0 && (!advance || it >= n_iters)) {
        store_a(out_a, p, a0, a1);
        p = -1;
      }
      continue;
    }
    // same exit rule as gn_kernel: s_{it+1} equal to s_it (fixed point) or to hist[k] = s_{it-1-k} (cycle of
    // k + 2 states) determines every later iterate.  Written with selects instead of branches; idle lanes run
    // through it too and are ignored.
    int hit = -2;
    if (exact_exit) {
      const long long b0 = __double_as_longlong(n0), b1 = __double_as_longlong(n1);
      if (b0 == __double_as_longlong(a0) && b1 == __double_as_longlong(a1)) hit = -1;
      if (HLDS) {
        // hist[k] = s_{it-1-k} for k >= 4 sits in ring slot (it - 1 - k) & 3 (written when it left the registers)
#pragma unroll
        for (int k = kGnHistory - 1; k >= kR; --k) {
          const longlong2 hs = lds_hist[(it - 1 - k) & 3][threadIdx.x];
          if (k < it && b0 == hs.x && b1 == hs.y && hit != -1) hit = k;
        }
      }
#pragma unroll
      for (int k = kR - 1; k >= 0; --k)
        if (k < it && b0 == h0[k] && b1 == h1[k] && hit != -1) hit = k;
    }
    // opt-in (DEXCT_GN_STOP_TOL, off by default): also stop when the step no longer moves the pixel by more than
    // stop_tol relative to max(|a|, 1) - not the reference's fixed count any more, but within stop_tol of it
    bool converged = false;
    if (stop_tol > 0.0) {
      const double size = fmax(fmax(fabs(n0), fabs(n1)), 1.0);
      converged = fmax(fabs(n0 - a0), fabs(n1 - a1)) <= stop_tol * size;      // NaN compares false
    }
    const bool advance = hit == -2 && !converged;
    // the state a cycle holds at iteration n_iters: s_m = s_{base + (m - base) mod period} for m >= base = it-1-hit,
    // and (n_iters - base) = (n_iters - it - 1) mod period; s_{base+j} is hist[hit-j], s_it the current state (slot -1)
    int slot = -1;
    if (hit >= 0) {
      const int period = hit + 2, x = n_iters - it - 1;
      int r;
      if (n_iters < (1 << 22)) {            // wave-uniform; small x: quotient by a float reciprocal, then corrected
        const int q = (int)((float)x * __builtin_amdgcn_rcpf((float)period));
        r = x - q * period;
        r += r < 0 ? period : 0;
        r -= r >= period ? period : 0;
      } else {
        r = x % period;
      }
      slot = hit - r;
    }
    double f0 = converged ? n0 : a0, f1 = converged ? n1 : a1;
    if (__ballot(slot >= 0) != 0ull) {
#pragma unroll
      for (int k = 0; k < kR; ++k)
        if (slot == k) { f0 = __longlong_as_double(h0[k]); f1 = __longlong_as_double(h1[k]); }
      if (HLDS && slot >= kR) {
        const longlong2 hs = lds_hist[(it - 1 - slot) & 3][threadIdx.x];
        f0 = __longlong_as_double(hs.x);
        f1 = __longlong_as_double(hs.y);
      }
    }
    if (HLDS && advance) lds_hist[(it - 4) & 3][threadIdx.x] = longlong2{h0[kR - 1], h1[kR - 1]};   // s_{it-4} leaves the registers
#pragma unroll
    for (int k = kR - 1; k > 0; --k) {
      h0[k] = advance ? h0[k - 1] : h0[k];
      h1[k] = advance ? h1[k - 1] : h1[k];
    }
    h0[0] = advance ? __double_as_longlong(a0) : h0[0];
    h1[0] = advance ? __double_as_longlong(a1) : h1[0];
    a0 = advance ? n0 : f0;
    a1 = advance ? n1 : f1;
    it += advance ? 1 : 0;
    if (p >= 0 && (!advance || it >= n_iters)) {
      store_a(out_a, p, a0, a1);
      p = -1;
    }
  }
  if (executed && (threadIdx.x & 63) == 0) {
    atomicAdd(executed, (unsigned long long)n_exec);            // one atomic per wave
    const int64_t first = ((int64_t)lblk * (kGnBlock / kWave) + (threadIdx.x >> 6)) * run;
    if (!queue && end > first) atomicAdd(executed + 1, (unsigned long long)(end - first));    // this wave's run of pixels is done
  }
}

__global__ __launch_bounds__(256) void mask_kernel(const void* __restrict__ g1, int g_is_f64, int64_t n_pix,
                                                   double thresh, double* __restrict__ out_a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pix) return;
  if (load_g<double>(g1, g_is_f64, p) >= thresh) {
    store_a(out_a, p, 0.0, 0.0);
  }
}

__global__ void max_init_kernel(double* out) { *out = -__builtin_huge_val(); }

// NaN-propagating, like np.max (matdecomp.py:195-196): one NaN count makes the maximum NaN, every comparison
// `g >= thresh * NaN` is then false and NO pixel is masked - exactly what the reference does with such a sinogram.
__device__ __forceinline__ double nanmax(double m, double v) {
  return (v != v || m != m) ? __builtin_nan("") : fmax(m, v);
}

__global__ __launch_bounds__(256) void max_kernel(const void* __restrict__ g1, int g_is_f64, int64_t n_pix,
                                                  double* __restrict__ out) {
  __shared__ double part[4];
  double m = -__builtin_huge_val();
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * 256)
    m = nanmax(m, load_g<double>(g1, g_is_f64, p));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = nanmax(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = nanmax(nanmax(part[0], part[1]), nanmax(part[2], part[3]));
    unsigned long long* addr = reinterpret_cast<unsigned long long*>(out);
    unsigned long long old = *addr;
    for (;;) {
      const double cur = __longlong_as_double((long long)old);
      if (cur != cur || !(m != m || cur < m)) break;        // a published NaN stays; otherwise publish NaN or a larger value
      const unsigned long long seen = atomicCAS(addr, old, (unsigned long long)__double_as_longlong(m));
      if (seen == old) break;
      old = seen;
    }
  }
}

}  // namespace dexct

using namespace dexct;

extern "C" {

int64_t dexct_gn_workspace_bytes(int32_t n_energies, int32_t n_bins) {
  if (n_energies <= 0 || n_bins <= 0) return 0;
  return (int64_t)sizeof(double) * kWsHeader + (int64_t)n_bins * n_energies * kTab * sizeof(double) +
         (int64_t)n_energies * kTab * sizeof(float) + (int64_t)n_bins * n_energies * sizeof(int) + 16;
}

int dexct_gn_decompose(const void* g1, const void* g2, int32_t g_is_f64, int64_t n_pix, const double* i0,
                       const double* mus, int32_t n_energies, int32_t n_bins, int32_t bin_div, int32_t n_iters,
                       int32_t precision, int32_t n_polish, const double* mask_max, double mask_frac, double* out_a,
                       void* workspace, void* stream) {
  if (!g1 || !g2 || !i0 || !mus || !out_a || !workspace || n_pix <= 0 || n_energies <= 0 || n_iters < 0)
    return DEXCT_EINVAL;
  if (n_bins < 1 || bin_div < 1 || n_bins > 65535) return DEXCT_EINVAL;
  if (precision != 0 && precision != 1) return DEXCT_EINVAL;
  if (precision == 1 && n_bins > 1) return DEXCT_EINVAL;   // mixed precision only with one shared spectrum
  if (n_polish < 0) return DEXCT_EINVAL;
  if (reinterpret_cast<uintptr_t>(out_a) & 15u) return DEXCT_EINVAL;      // a pixel's two doubles leave as one 16-byte store
  if (n_energies > 4096) return DEXCT_ERANGE;
  const int64_t nblk = (n_pix + kGnBlock - 1) / kGnBlock;
  if (nblk > 0x7FFFFFFFll) return DEXCT_ERANGE;
  hipStream_t st = as_stream(stream);
  double* ws = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(gn_tables_kernel, dim3(n_bins), dim3(256), 0, st, i0, mus, n_energies, n_bins, ws);
  DEXCT_LAUNCH_CHECK();
  // DEXCT_GN_FULL_LOOP=1 runs every iteration (to check that the repeated-state exit changes no bit)
  const char* full = getenv("DEXCT_GN_FULL_LOOP");
  const int exact_exit = (full && full[0] == '1') ? 0 : 1;
  const dim3 grid((unsigned)nblk), block(kGnBlock);
  if (n_bins > 1) {
    hipLaunchKernelGGL((gn_kernel<false, true>), grid, block, 0, st, g1, g2, g_is_f64, n_pix, (const double*)ws,
                       n_energies, n_iters, 0, n_bins, bin_div, mask_max, mask_frac, exact_exit, out_a);
  } else if (precision == 0) {
    // lane refill: each wave works through a run of 64 * chunk pixels.  Measured optimum (8e5 ... 1e9 pixels; round 3, after
    // the XCD-contiguous run assignment, tools/bench_gn.py GN_VARIANTS=chunk): up to 6 pixels per lane while that still
    // leaves ~12 500 waves (2.4 x the 5 120 resident ones), beyond that as many as keep the grid near 133 000 waves, at
    // most 32 - 5.1e7 pixels (one GPU's share of an 8-GPU scan): 4 / 6 / 8 / 12 / 16 = 108 / 107 / 111 / 108 / 110 ms;
    // 1.0e8: 8 / 12 / 16 / 24 = 216 / 209 / 211 / 211; 2.0e8: 16 / 24 / 32 = 413 / 406 / 410; 4.1e8: 24 / 32 / 48 / 64 = 802 / 801
    // / 804 / 801; 1.05e9: 24 / 32 / 64 = 2036 / 2026 / 2058.  Small inputs degenerate to one pixel per lane.
    const char* ce = getenv("DEXCT_GN_CHUNK");
    int64_t chunk = n_pix / (kWave * 12500ll);
    if (chunk > 6) chunk = 6;
    if (n_pix / (kWave * 133000ll) > chunk) chunk = n_pix / (kWave * 133000ll);
    if (ce) chunk = atoll(ce);
    chunk = chunk < 1 ? 1 : (chunk > (ce ? 1024 : 32) ? (ce ? 1024 : 32) : chunk);
    // run queue (default; DEXCT_GN_QUEUE=0 restores the static runs above): 2 pixels per lane and fetch (benchmark
    // sinograms: 1 / 2 / 4 / 8 / 16 = 766 / 764 / 769 / 771 / 773 ms against 811 ms with static runs; an 8-GPU share, 5.1e7
    // pixels: 98 / 97 / 99 / 102 / 106 against 108), and no more workgroups than could ever be resident
    const char* qe = getenv("DEXCT_GN_QUEUE");
    const bool use_queue = !(qe && atoi(qe) == 0);
    // Below ~1e8 pixels a fetch of 64 pixels is better (round 3, tools/probes/gn_small2.py: the reference's own 1200 x 800
    // single-row sinogram 4.33 -> 3.61 ms, configs[1]'s 360 x 512: 2.06 -> 1.59, 5.1e6 pixels 11.6 -> 11.0, 2.6e7: 51.8 -> 51.5):
    // the tail of a small launch is the last runs' slowest pixels, and halving a run halves what a wave can be left with.
    if (use_queue && !ce) chunk = n_pix < 100000000ll ? 1 : 2;
    int64_t n_waves = (n_pix + kWave * chunk - 1) / (kWave * chunk);
    int64_t nb = (n_waves + kGnBlock / kWave - 1) / (kGnBlock / kWave);
    if (use_queue) {
      static const int n_cu = [] {             // queried once per process (one GPU per process)
        int dev_id = 0, n = 0;
        if (hipGetDevice(&dev_id) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess || n <= 0)
          n = 256;
        return n;
      }();
      const char* be = getenv("DEXCT_GN_BLOCKS_PER_CU");          // tuning knob
      const int64_t cap = (int64_t)n_cu * (be && atoi(be) > 0 ? atoi(be) : 8);       // 8 waves per SIMD is the hardware's most: whatever this instantiation's
                                                   // occupancy, every resident slot gets a workgroup; the rest find the queue empty
      if (nb > cap) nb = cap;
    }
    unsigned long long* queue = use_queue ? reinterpret_cast<unsigned long long*>(ws) + 11 : nullptr;
    const char* te = getenv("DEXCT_GN_STOP_TOL");
    const double stop_tol = te ? atof(te) : 0.0;
    // tuning knobs for A/B runs (defaults are the measured optimum): DEXCT_GN_MINW=4 trades occupancy
    // for a spill-free register allocation, DEXCT_GN_IEXP=1 scales by 2^k with integer adds instead of v_ldexp_f64
    const char* ve = getenv("DEXCT_GN_MINW");
    const char* ie = getenv("DEXCT_GN_IEXP");
    const char* he = getenv("DEXCT_GN_HLDS");
    const int hlds = (he && atoi(he) == 1) ? 1 : 0;
    const int minw = (ve && atoi(ve) == 4) ? 4 : 5;
    const int iexp = (ie && atoi(ie) == 1) ? 1 : 0;
    unsigned long long* stat = reinterpret_cast<unsigned long long*>(ws) + 9;
    const double tol = stop_tol > 0.0 ? stop_tol : 0.0;
#define DEXCT_GN_LAUNCH(MW, IE)                                                                                         \
  hipLaunchKernelGGL((gn_refill_kernel<MW, IE>), dim3((unsigned)nb), block, 0, st, g1, g2, g_is_f64, n_pix,            \
                     (const double*)ws, n_energies, n_iters, (int)chunk, mask_max, mask_frac, exact_exit, tol, out_a, stat, queue)
#define DEXCT_GN_LAUNCH_H(MW, H)                                                                                        \
  hipLaunchKernelGGL((gn_refill_kernel<MW, false, false, H>), dim3((unsigned)nb), block, 0, st, g1, g2, g_is_f64, n_pix, \
                     (const double*)ws, n_energies, n_iters, (int)chunk, mask_max, mask_frac, exact_exit, tol, out_a, stat, queue)
    const char* hs = getenv("DEXCT_GN_HIST");
    const int hist = hs ? atoi(hs) : kGnHistory;
    const int mw = ve ? atoi(ve) : 5;
    const char* re = getenv("DEXCT_GN_RING");
    const int ring = re ? atoi(re) : kGnRingDefault;
    if (ring && !hlds && hist == kGnHistory && !iexp && mw == 5)
      hipLaunchKernelGGL((gn_refill_kernel<5, false, false, kGnHistory, true>), dim3((unsigned)nb), block, 0, st, g1, g2, g_is_f64,
                         n_pix, (const double*)ws, n_energies, n_iters, (int)chunk, mask_max, mask_frac, exact_exit, tol, out_a, stat, queue);
    else if (ring && !hlds && hist == kGnHistory && !iexp && mw == 4)
      hipLaunchKernelGGL((gn_refill_kernel<4, false, false, kGnHistory, true>), dim3((unsigned)nb), block, 0, st, g1, g2, g_is_f64,
                         n_pix, (const double*)ws, n_energies, n_iters, (int)chunk, mask_max, mask_frac, exact_exit, tol, out_a, stat, queue);
    else if (hlds)
      hipLaunchKernelGGL((gn_refill_kernel<5, false, true>), dim3((unsigned)nb), block, 0, st, g1, g2, g_is_f64, n_pix,
                         (const double*)ws, n_energies, n_iters, (int)chunk, mask_max, mask_frac, exact_exit, tol, out_a, stat, queue);
    else if (hist == 4 && mw == 6) DEXCT_GN_LAUNCH_H(6, 4);
    else if (hist == 6 && mw == 6) DEXCT_GN_LAUNCH_H(6, 6);
    else if (hist == 4) DEXCT_GN_LAUNCH_H(5, 4);
    else if (hist == 5) DEXCT_GN_LAUNCH_H(5, 5);
    else if (hist == 6) DEXCT_GN_LAUNCH_H(5, 6);
    else if (hist == 7) DEXCT_GN_LAUNCH_H(5, 7);
    else if (hist == 10) DEXCT_GN_LAUNCH_H(5, 10);
    else if (hist == 12) DEXCT_GN_LAUNCH_H(5, 12);
    else if (minw == 4 && iexp) DEXCT_GN_LAUNCH(4, true);
    else if (minw == 4) DEXCT_GN_LAUNCH(4, false);
    else if (iexp) DEXCT_GN_LAUNCH(5, true);
    else DEXCT_GN_LAUNCH(5, false);
#undef DEXCT_GN_LAUNCH_H
#undef DEXCT_GN_LAUNCH
  } else {
    hipLaunchKernelGGL((gn_kernel<true, false>), grid, block, 0, st, g1, g2, g_is_f64, n_pix, (const double*)ws,
                       n_energies, n_iters, n_polish, 1, 1, mask_max, mask_frac, exact_exit, out_a);
  }
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_gn_apply_mask(const void* g1, int32_t g_is_f64, int64_t n_pix, double thresh_value, double* out_a,
                        void* stream) {
  if (!g1 || !out_a || n_pix <= 0) return DEXCT_EINVAL;
  const int64_t nblk = (n_pix + 255) / 256;
  if (nblk > 0x7FFFFFFFll) return DEXCT_ERANGE;
  hipLaunchKernelGGL(mask_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), g1, g_is_f64, n_pix,
                     thresh_value, out_a);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_reduce_max(const void* g1, int32_t g_is_f64, int64_t n_pix, double* out_max, void* stream) {
  if (!g1 || !out_max || n_pix <= 0) return DEXCT_EINVAL;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(max_init_kernel, dim3(1), dim3(1), 0, st, out_max);
  DEXCT_LAUNCH_CHECK();
  int64_t nblk = (n_pix + 255) / 256;
  if (nblk > 2048) nblk = 2048;
  hipLaunchKernelGGL(max_kernel, dim3((unsigned)nblk), dim3(256), 0, st, g1, g_is_f64, n_pix, out_max);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

}  // extern "C"
