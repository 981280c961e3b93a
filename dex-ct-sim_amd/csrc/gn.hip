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
#ifndef DEXCT_GN_HISTORY
#define DEXCT_GN_HISTORY 8
#endif
constexpr int kGnHistory = DEXCT_GN_HISTORY;

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
// 2^k without an instruction of its own (round 6): the table entry j holds 2^(j/2048) with j << 9 SUBTRACTED from its high word
// (pow_entry), so that adding n << 9 = (k << 20) + (j << 9) to the high word of what was loaded - one v_lshl_add_u32, no mask -
// gives 2^k 2^(j/2048) before the last FMA; scaling by a power of two is exact, so the bits are those of ldexp(fma(tj, p, tj), k)
// (rounds 3-5: a shift and v_ldexp_f64) wherever that is a normal number - and it is for |x| <= 700 (e^-700 = 1e-304), which every
// caller guarantees: the clip of matdecomp.py:116, the clip-free bound of newton_sums_f64, float32-normal arguments of log_pos.
// NaN stays NaN (f is NaN then, whatever the table entry became).
constexpr int kPowBits = 11;
constexpr int kPowN = 1 << kPowBits;
constexpr double kExpScale = 0x1.71547652b82fep+11;          // 2048 / ln 2
constexpr double kExpClip = 700.0 * kExpScale;               // the reference's clip of the exponent (matdecomp.py:116)
__device__ __forceinline__ double pow_entry(int j) {
  const double v = exp2((double)j * (1.0 / kPowN));
  return __hiloint2double(__double2hiint(v) - (j << (20 - kPowBits)), __double2loint(v));
}
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
  const double tr = lds_pow[ni & (kPowN - 1)];
  const double tj = __hiloint2double((int)((unsigned)__double2hiint(tr) + ((unsigned)ni << (20 - kPowBits))), __double2loint(tr));
  return fma(tj, p, tj);
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
#ifndef DEXCT_GN_ENERGY_UNROLL
#define DEXCT_GN_ENERGY_UNROLL 4
#endif
template <int KSEL, bool CLIP, int SUMS = 2>       // KSEL 0: both measurements, 1: only k = 0, 2: only k = 1; CLIP: apply the +-700 clip;
                                                   // SUMS 2: all six sums per measurement (a Newton step), 0: the expected counts nu
                                                   // alone (the chord step of the short cut: 2 of the 12 accumulations per energy)
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
    const double at = exp_tab(y, lds_pow);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if ((KSEL == 1 && k == 1) || (KSEL == 2 && k == 0)) continue;
      const double* __restrict__ tk = t + 2 + 6 * k;
      nuk[k] = fma(tk[0], at, nuk[k]);
      if (SUMS >= 1) {
        G0[k] = fma(tk[1], at, G0[k]);
        G1[k] = fma(tk[2], at, G1[k]);
      }
      if (SUMS >= 2) {
        H00[k] = fma(tk[3], at, H00[k]);
        H01[k] = fma(tk[4], at, H01[k]);
        H11[k] = fma(tk[5], at, H11[k]);
      }
    }
  };
  int e = e0;
  // (4 pairs per trip: eight table rows' scalar loads and eight LDS reads in flight - 27.2 -> 26.4 ms for the chord launch of the
  // benchmark, 756 -> 747 ms for the exact count; 8 pairs: no further gain.  tools/probes/gn_ab.py, build_variant.sh)
#pragma unroll DEXCT_GN_ENERGY_UNROLL
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

// The 12 sums over energy a Newton step needs.  NPARTS = 1: all energies (the one-lane-per-pixel kernels).  NPARTS > 1:
// share `part` of them - every class range is cut into NPARTS contiguous pieces - for the cooperative kernel, whose waves
// split the energy loop of the same 64 pixels.
struct GnSums { double nu[2], G0[2], G1[2], H00[2], H01[2], H11[2]; };

template <int NPARTS, int SUMS = 2>
__device__ __forceinline__ void newton_sums_f64(const double* __restrict__ tab, const double* __restrict__ lds_pow,
                                                EnergyClasses ec, int part, double a0, double a1, GnSums& s) {
  double nu[2] = {0, 0}, nuo[2] = {0, 0}, G0[2] = {0, 0}, G1[2] = {0, 0}, H00[2] = {0, 0}, H01[2] = {0, 0}, H11[2] = {0, 0};
  const int bA = 0, bB = ec.nA, bC = ec.nA + ec.nB;
  auto lo = [&](int b, int e) { return NPARTS == 1 ? b : b + (e - b) * part / NPARTS; };
  auto hi = [&](int b, int e) { return NPARTS == 1 ? e : b + (e - b) * (part + 1) / NPARTS; };
  // the always-clipped heads of the three classes
  energy_sums_f64<0, true, SUMS>(tab, lds_pow, lo(bA, bA + ec.nAc), hi(bA, bA + ec.nAc), a0, a1, nu, nuo, G0, G1, H00, H01, H11);
  energy_sums_f64<1, true, SUMS>(tab, lds_pow, lo(bB, bB + ec.nBc), hi(bB, bB + ec.nBc), a0, a1, nu, nuo, G0, G1, H00, H01, H11);
  energy_sums_f64<2, true, SUMS>(tab, lds_pow, lo(bC, bC + ec.nCc), hi(bC, bC + ec.nCc), a0, a1, nu, nuo, G0, G1, H00, H01, H11);
  // the tails: clip-free when the bound holds for this pixel (NaN compares false -> clipped path)
  const int tA0 = lo(bA + ec.nAc, bA + ec.nA), tA1 = hi(bA + ec.nAc, bA + ec.nA);
  const int tB0 = lo(bB + ec.nBc, bB + ec.nB), tB1 = hi(bB + ec.nBc, bB + ec.nB);
  const int tC0 = lo(bC + ec.nCc, bC + ec.nC), tC1 = hi(bC + ec.nCc, bC + ec.nC);
  if (fma(fabs(a1), ec.m1_free, fabs(a0) * ec.m0_free) <= 699.9) {
    energy_sums_f64<0, false, SUMS>(tab, lds_pow, tA0, tA1, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
    energy_sums_f64<1, false, SUMS>(tab, lds_pow, tB0, tB1, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
    energy_sums_f64<2, false, SUMS>(tab, lds_pow, tC0, tC1, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
  } else {
    energy_sums_f64<0, true, SUMS>(tab, lds_pow, tA0, tA1, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
    energy_sums_f64<1, true, SUMS>(tab, lds_pow, tB0, tB1, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
    energy_sums_f64<2, true, SUMS>(tab, lds_pow, tC0, tC1, a0, a1, nu, nuo, G0, G1, H00, H01, H11);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    s.nu[k] = nu[k] + nuo[k];
    s.G0[k] = G0[k]; s.G1[k] = G1[k]; s.H00[k] = H00[k]; s.H01[k] = H01[k]; s.H11[k] = H11[k];
  }
}

// The step from the sums: residuals, gradient, Hessian incl. the (g/nu - 1) * hessian term (matdecomp.py:122-123), closed-form
// 2x2 solve (:125).
__device__ __forceinline__ void newton_solve_f64(const GnSums& s, double g0, double g1, double& a0, double& a1) {
  const double g[2] = {g0, g1};
  double c[2], q[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const double inv = rcp_f64(s.nu[k]), ratio = g[k] * inv;
    // g / nu - 1 (matdecomp.py:122) as (g - nu) / nu: near the solution the subtraction is exact, so the residual
    // carries the rounding of nu only, not an extra half ulp of 1 from the quotient - the last-bit wandering of the
    // iterate is shorter again (mean executed iterations 28.6 -> 24.7).  An overflowed nu (inf) keeps the reference's
    // value: g / inf - 1 = -1.
    c[k] = fabs(s.nu[k]) < __builtin_huge_val() ? (g[k] - s.nu[k]) * inv : ratio - 1.0;
    q[k] = ratio * inv;                                        // g / nu^2 (matdecomp.py:123)
  }
  // the sums over the two measurements, first term + second term (a running sum started at 0 costs an addition of 0
  // per sum that the compiler may not drop: 0 + (-0) is +0)
  const double dF0 = c[0] * s.G0[0] + c[1] * s.G0[1];
  const double dF1 = c[0] * s.G1[0] + c[1] * s.G1[1];
  const double h00 = (q[0] * (s.G0[0] * s.G0[0]) - c[0] * s.H00[0]) + (q[1] * (s.G0[1] * s.G0[1]) - c[1] * s.H00[1]);
  const double h01 = (q[0] * (s.G0[0] * s.G1[0]) - c[0] * s.H01[0]) + (q[1] * (s.G0[1] * s.G1[1]) - c[1] * s.H01[1]);
  const double h11 = (q[0] * (s.G1[0] * s.G1[0]) - c[0] * s.H11[0]) + (q[1] * (s.G1[1] * s.G1[1]) - c[1] * s.H11[1]);
  const double inv_det = rcp_f64(h00 * h11 - h01 * h01);
  a0 -= (h11 * dF0 - h01 * dF1) * inv_det;
  a1 -= (h00 * dF1 - h01 * dF0) * inv_det;
}

__device__ __forceinline__ void newton_step_f64(const double* __restrict__ tab, const double* __restrict__ lds_pow,
                                                EnergyClasses ec, double g0, double g1, double& a0, double& a1) {
  GnSums s;
  newton_sums_f64<1>(tab, lds_pow, ec, 0, a0, a1, s);
  newton_solve_f64(s, g0, g1, a0, a1);
}

// THE CHORD STEP of the short cut (gn_shortcut_kernel<1>; round 6 - round 5 took a Gauss-Newton step here: 6 of the 12 sums).
// From a start value s within 1e-10 of |a| of the fixed point a*, what a step has to get right is the RESIDUAL - the relative
// misfit of the counts c_k = g_k / nu_k(s) - 1, which needs the full energy sum of nu to its last bits - not the Jacobian it is
// multiplied with: a relative error eps of the Jacobian leaves eps e0 of the distance.  And the inverse of the model's
// log-Jacobian, B = (d ln nu / d a)^-1, is already in the table: it is the GRADIENT of the tabulated fixed points with respect to
// the (logarithms of the) counts - gn_start<DERIV> forms it from the same 36 loads - so the energy loop carries 2 accumulations per energy
// (nu of both measurements) instead of 6: with c(a) = g / nu(a) - 1, Dc(a*) = -L*, the step m = s + Bt c(s) leaves
//     m - a* = (I - Bt L*) e0 + 1/2 Bt D2c(xi) [e0, e0],      |e1| <= eps |e0| + kappa |e0|^2,
// eps >= |I - Bt L*| (what the derivative of the interpolant leaves: measured by the host at every cell's corners - where the
// derivative of an even-order interpolant is worst - and centre, against the exact Jacobians there; per cell, with a safety factor)
// and kappa = 1/2 max_i sum_j |B_ij| sum_pq |D2c_jpq| (quadrature.chord_tables; both per cell, as the pair (kappa, eps)).
// The residuals only:
__device__ __forceinline__ void chord_residuals_f64(const double* __restrict__ tab, const double* __restrict__ lds_pow,
                                                    EnergyClasses ec, double g0, double g1, double a0, double a1, double (&c)[2]) {
  GnSums s;
  newton_sums_f64<1, 0>(tab, lds_pow, ec, 0, a0, a1, s);
  const double g[2] = {g0, g1};
#pragma unroll
  for (int k = 0; k < 2; ++k) {                                  // (g - nu) / nu as in newton_solve_f64
    const double inv = rcp_f64(s.nu[k]);
    c[k] = fabs(s.nu[k]) < __builtin_huge_val() ? (g[k] - s.nu[k]) * inv : g[k] * inv - 1.0;
  }
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
// [10] = (uint64, progress) pixels that launch has handed to its waves so far (added tile by tile while it runs),
// [11] = (uint64) head of the tile queue, [12] = (uint64, diagnostic) lane-steps spent without a pixel because all of a
// wave's result slots were waiting for stragglers,
// pad to 16, then [n_e][14] float64, then [n_e][14] float32, then n_e ints (the permutation), then (16-byte aligned) the
// tile-order region of gn_tile_* below: kSortBuckets ints (histogram), kMaxSortTiles ints (the order), kMaxSortTiles
// uint16 (keys).
constexpr int kWsHeader = 16;
constexpr int kSortBuckets = 2048;         // sign + exponent + 2 mantissa bits of a positive float32
constexpr int kMaxSortTiles = 32768;       // 2.1e6 pixels: beyond that the order of the hand-out does not matter (profiles/r04_gn.md)

__host__ __device__ inline size_t gn_ws_tables_bytes(int n_e, int n_bins) {
  const size_t b = sizeof(double) * kWsHeader + (size_t)n_bins * n_e * kTab * sizeof(double) + (size_t)n_e * kTab * sizeof(float) +
                   (size_t)n_bins * n_e * sizeof(int);
  return (b + 15) & ~(size_t)15;
}

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
      reinterpret_cast<unsigned long long*>(ws)[11] = 0ull;     // head of the tile queue of gn_refill_kernel
      reinterpret_cast<unsigned long long*>(ws)[12] = 0ull;     // lane-steps stalled on result slots
    }
  }
  if (bin == 0) {                       // the histogram of the tile sort starts at zero
    int* hist = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + gn_ws_tables_bytes(n_e, n_bins));
    for (int j = threadIdx.x; j < kSortBuckets; j += blockDim.x) hist[j] = 0;
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

// ---- pixel tiles, the run queue and the order of the results ---------------------------------------------------------
// The lane-refill kernel hands pixels out in TILES of 64 (one per lane of a wave) that waves fetch from a global counter.
// A tile is also the unit in which results leave: a wave collects the 64 (a0, a1) pairs of a tile in LDS and, when the last
// one has arrived, writes the tile with one wave-wide store of whole 128-byte lines (round 3 stored every pair by itself
// the moment its pixel ended: WRITE_SIZE 1.9 x the results, tools/probes/gn_write.py).
//   plain:       tile t = pixels [64 t, 64 t + 64), results in the order of the pixels.
//   transposed:  the sinograms are [view][channel][row] (row fastest - what the stacked-fan projection writes, layout 1 of
//                dexct_siddon_project) and the results go out as [view][row][channel] (the reference's order,
//                matdecomp.py:200-201): a tile is 4 channels x 16 rows of one view - on the input side 4 runs of 16
//                consecutive floats (64 bytes), on the output side 16 runs of 4 consecutive (a0, a1) pairs = 64 bytes
//                each.  This replaces the separate transpose pass over the 6.5 GB of results (2.3 ms and one more round
//                trip through HBM in round 3).  Tile shape measured on the benchmark sinograms, exact mode, same box
//                (profiles/r04_gn.md): 8 x 8: 770 ms, 4 x 16: 763, 2 x 32: 762, 1 x 64: 766; the plain order: 763.
constexpr int kTilePix = kWave;
#ifndef DEXCT_GN_SLOTS
#define DEXCT_GN_SLOTS 3
#endif
#ifndef DEXCT_GN_TILE_CLOG2
#define DEXCT_GN_TILE_CLOG2 2
#endif
constexpr int kSlots = DEXCT_GN_SLOTS;   // tiles a wave may have in flight: 3 x 64 x 16 B = 3 KB of LDS per wave (with 4 the 32 KB of a
                                         // workgroup leave room for 4 workgroups per CU instead of 5: +2.5 %, profiles/r04_gn.md)
constexpr int kTileCLog2 = DEXCT_GN_TILE_CLOG2, kTileC = 1 << kTileCLog2, kTileR = kTilePix / kTileC;   // transposed tiles: channels x rows

struct GnTiling {
  long long n_tiles;
  int transposed;                    // 0 / 1
  int rows, channels;                // transposed: R, C
  int tiles_r, tiles_c;              // ceil(R / kTileR), ceil(C / kTileC)
  unsigned mul_r, sh_r, mul_c, sh_c; // n / tiles_r and n / tiles_c as (mulhi(n, mul) + n) >> sh (gn_magic): a tile is decoded per 64
                                     // pixels, and on the short cut - one step per pixel - two 30-instruction divisions showed
};

// Division of a 32-bit n by an invariant d >= 1 (Granlund & Montgomery, the round-up form): s = ceil(log2 d),
// m = floor(2^32 (2^s - d) / d) + 1, n / d = (mulhi(n, m) + n) >> s, the sum in 64 bits.  Exact for every n < 2^32.
inline void gn_magic(unsigned d, unsigned* mul, unsigned* shift) {
  unsigned s = 0;
  while ((1ull << s) < (unsigned long long)d) ++s;
  *mul = (unsigned)((((1ull << s) - d) << 32) / d + 1ull);
  *shift = s;
}

__device__ __forceinline__ unsigned gn_fastdiv(unsigned n, unsigned mul, unsigned shift) {
  return (unsigned)(((unsigned long long)__umulhi(n, mul) + n) >> shift);
}

struct GnTile {                      // wave-uniform description of one tile
  long long in_base, out_base;       // pixel index of the tile's first pixel in the input / output order
  int nr, nc;                        // valid rows / channels (edge tiles); plain: nr = 1, nc = valid pixels
};

__device__ __forceinline__ GnTile gn_decode_tile(const GnTiling& tl, long long n_pix, long long t) {
  GnTile d;
  if (!tl.transposed) {
    d.in_base = d.out_base = t * kTilePix;
    const long long left = n_pix - d.in_base;
    d.nr = 1;
    d.nc = left < kTilePix ? (int)left : kTilePix;
  } else {
    const unsigned tu = (unsigned)t;                                     // n_tiles < 2^31 (checked by the host)
    const unsigned u = gn_fastdiv(tu, tl.mul_r, tl.sh_r), rb = tu - u * (unsigned)tl.tiles_r;
    const unsigned v = gn_fastdiv(u, tl.mul_c, tl.sh_c), cb = u - v * (unsigned)tl.tiles_c;
    const long long R = tl.rows, C = tl.channels;
    d.in_base = ((long long)v * C + (long long)kTileC * cb) * R + (long long)kTileR * rb;
    d.out_base = ((long long)v * R + (long long)kTileR * rb) * C + (long long)kTileC * cb;
    d.nr = tl.rows - kTileR * (int)rb < kTileR ? tl.rows - kTileR * (int)rb : kTileR;
    d.nc = tl.channels - kTileC * (int)cb < kTileC ? tl.channels - kTileC * (int)cb : kTileC;
  }
  return d;
}

// where pixel p of the input order goes in the output order (the kernels that store pixel by pixel)
__device__ __forceinline__ long long gn_out_index(const GnTiling& tl, long long p) {
  if (!tl.transposed) return p;
  const long long R = tl.rows, C = tl.channels;
  const long long u = p / R, r = p - u * R;
  const long long v = u / C, c = u - v * C;
  return (v * R + r) * C + c;
}

// The tolerance stop (the default since round 4; stop_tol = 0 keeps the reference's fixed count bit for bit).  After a step of
// size d_k that followed one of size d_{k-1} > d_k, a sequence whose contraction does not get worse than r = d_k / d_{k-1}
// still has at most d_k r / (1 - r) to go (Newton's contraction only improves near the solution, so this is generous).  The
// pixel ends - with the state AFTER the step - when that remainder is at most stop_tol / 4 * max(|a|, 1):
//     d_k^2 <= (stop_tol / 4) * max(|a|, 1) * (d_{k-1} - d_k).
// For r <= 1/2 this is implied by d_k <= stop_tol / 4 * size (the plain "step below tolerance" rule); for a quadratically
// converging pixel it fires one iteration before that rule would - at the step whose successor would be below the
// tolerance - and that iteration is what it saves (17.4 -> 16.4 on the benchmark sinograms).  A pixel that creeps (r near 1: a
// nearly singular Hessian, a wild transient far from the solution where steps are small only relative to a huge |a|) or
// whose steps grow is NOT stopped and runs on to the exact repeated-state exit or to n_iters, as in the reference.  prev0 /
// prev1: the state before (a0, a1) (history slot 0), valid when it >= 1; no stop after the very first step.
__device__ __forceinline__ bool gn_rule(double stop_tol, double a0, double a1, double n0, double n1, double prev0, double prev1) {
  const double dk = fmax(fabs(n0 - a0), fabs(n1 - a1));
  const double size = fmax(fmax(fabs(n0), fabs(n1)), 1.0);
  const double dprev = fmax(fabs(a0 - prev0), fabs(a1 - prev1));
  return dk < dprev && dk * dk <= (0.25 * stop_tol) * size * (dprev - dk) && dprev < __builtin_huge_val();   // NaN compares false
}

// prev2: the state before prev (history slot 1), valid when it >= 2.  `confirm`: the walk from the reference's start value ends
// only when the step BEFORE contracted as well (d_(k-1) < d_(k-2): three steps in decreasing order).  A wandering pixel -
// photon-starved counts that no thicknesses reproduce, drifting towards |a| ~ 1e3 - takes a step now and then that happens
// to be tiny next to its predecessor and was stopped there, where the reference walks on (to NaN, mostly): 481 + 48 of the
// 4.7e7 pixels of a noisy 2e4-photon scan differed from the exact count (profiles/r04_gn.md section 1).  A converging pixel's
// steps decrease anyway.  (Asking the previous step to meet the whole rule would: a quadratically
// converging pixel meets it once, at its last step above the rounding floor.)  Not asked of a pixel that starts next to a
// tabulated, isolated fixed point (the short cut): its first step has no predecessor.
__device__ __forceinline__ bool gn_converged(double stop_tol, double a0, double a1, double n0, double n1, double prev0,
                                             double prev1, int it, bool confirm = false, double prev2_0 = 0.0, double prev2_1 = 0.0) {
  if (!(stop_tol > 0.0) || it < 1) return false;
  bool ok = gn_rule(stop_tol, a0, a1, n0, n1, prev0, prev1);
  if (confirm) {
    // ... and the distance still to go is estimated with the WORSE of the last two contraction ratios (a creeping pixel's
    // ratio fluctuates: one good step does not make a geometric series): d_k r / (1 - r) <= stop_tol / 4 * size with
    // r = d_(k-1) / d_(k-2) as well, i.e. d_k d_(k-1) <= stop_tol / 4 * size * (d_(k-2) - d_(k-1))
    const double dk = fmax(fabs(n0 - a0), fabs(n1 - a1));
    const double size = fmax(fmax(fabs(n0), fabs(n1)), 1.0);
    const double dprev = fmax(fabs(a0 - prev0), fabs(a1 - prev1));
    const double dprev2 = fmax(fabs(prev0 - prev2_0), fabs(prev1 - prev2_1));
    ok = ok && it >= 2 && dprev < dprev2 && dk * dprev <= (0.25 * stop_tol) * size * (dprev2 - dprev);
  }
  return ok;
}

// MIXED: n_iters - n_polish iterations in float32, then n_polish in float64.
// PER_BIN: pixel p uses the tables of bin (p / bin_div) % n_bins (channel-dependent spectra, the general
// signature of optimize_sino_cpu); the table pointer is then per lane and the values arrive by vector loads.
template <bool MIXED, bool PER_BIN>
__global__ __launch_bounds__(kGnBlock) void gn_kernel(const void* __restrict__ g1, const void* __restrict__ g2,
                                                      int g_is_f64, int64_t n_pix, const double* __restrict__ ws,
                                                      int n_e, int n_iters, int n_polish, int n_bins, int bin_div,
                                                      const double* __restrict__ mask_max, double mask_frac,
                                                      int exact_exit, double stop_tol, GnTiling tl,
                                                      double* __restrict__ out_a) {
  __shared__ double lds_pow[kPowN];     // 2^(j/2048) in pow_entry's form, 16 KB
  for (int j = threadIdx.x; j < kPowN; j += kGnBlock) lds_pow[j] = pow_entry(j);
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
  const long long po = gn_out_index(tl, p);              // where the result goes (a scattered 16-byte store either way)
  if (mask_max && gd0 >= mask_frac * mask_max[0]) {
    store_a(out_a, po, 0.0, 0.0);
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
    if (exact_exit & 1) {
      const long long b0 = __double_as_longlong(n0), b1 = __double_as_longlong(n1);
      if (b0 == __double_as_longlong(a0) && b1 == __double_as_longlong(a1)) hit = -1;
#pragma unroll
      for (int k = kGnHistory - 1; k >= 0; --k)              // descending, so the smallest period wins
        if (k < it && b0 == h0[k] && b1 == h1[k] && hit != -1) hit = k;
      if (hit != -2) break;
      // the tolerance stop (float64 loop only; see gn_converged)
      if (gn_converged(stop_tol, a0, a1, n0, n1, __longlong_as_double(h0[0]), __longlong_as_double(h1[0]), it, (exact_exit & 4) != 0,
                       __longlong_as_double(h0[1]), __longlong_as_double(h1[1]))) {
        a0 = n0;
        a1 = n1;
        break;
      }
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
  store_a(out_a, po, a0, a1);
}


__device__ __forceinline__ const int* gn_tile_order(const unsigned long long* counters, int n_e) {      // counters = workspace word 9; shared-spectrum kernels: n_bins = 1
  return reinterpret_cast<const int*>(reinterpret_cast<const char*>(counters - 9) + gn_ws_tables_bytes(n_e, 1)) + kSortBuckets;
}

// ---- order of the hand-out for SMALL sinograms: longest tiles first ---------------------------------------------------
// The number of Newton iterations of a pixel grows with the attenuation along its ray (from the start value 1e-6 the
// iteration walks about one unit of exponent per step before it converges quadratically: 7 iterations for a thin ray, 26
// through the centre of the benchmark phantom), so the counts themselves say how long a tile will take.  When the sinogram
// is so small that every lane sees only a few pixels (the reference's own scan: 9.6e5 pixels for 3.3e5 resident lanes), the
// launch ends with whatever was handed out last; handing the thick tiles out FIRST leaves the thin ones for the end and
// keeps the pixels a wave works on at any time alike: 2.50 -> 2.06 ms on the 1200 x 800 scan (tools/probes/gn_order.py).
// A counting sort of the tiles by the smallest unmasked count of sinogram 1 in the tile (2048 buckets = the top bits of the
// float32 value; all-air tiles last), three tiny kernels (keys + histogram, scan, scatter); up to kMaxSortTiles tiles.  The order only decides WHEN a pixel
// is solved: results do not depend on it.
// One atomic per wave and distinct key instead of one per lane (air tiles and the thick centre share few buckets: per-lane
// atomics on those few addresses serialise - measured 2 ms at 6e4 tiles).  Returns this lane's rank among the lanes of the
// wave with the same key, and *group_base = what the leader's atomicAdd(counter + key, group size) returned.
__device__ __forceinline__ int wave_grouped_add(int* __restrict__ counter, int key, bool valid, int* group_base) {
  const int lane = threadIdx.x & 63;
  unsigned long long rem = __ballot(valid);
  int rank = 0, base = 0;
  while (rem != 0ull) {
    const int leader = __builtin_ctzll(rem);
    const int k = __shfl(key, leader, 64);
    const unsigned long long m = __ballot(valid && key == k);
    int b = 0;
    if (lane == leader) b = atomicAdd(counter + k, (int)__popcll(m));
    b = __shfl(b, leader, 64);
    if (valid && key == k) {
      base = b;
      rank = (int)__popcll(m & ((1ull << lane) - 1ull));
    }
    rem &= ~m;
  }
  *group_base = base;
  return rank;
}

// 64 tiles per workgroup: a wave reads a tile with one coalesced load per lane (16 tiles per wave, loads independent of
// each other), reduces to the smallest unmasked count, and the first wave then adds the 64 keys to the histogram.
__global__ __launch_bounds__(256) void gn_tile_key_kernel(const void* __restrict__ g1, int g_is_f64, long long n_pix, GnTiling tl,
                                                          const double* __restrict__ mask_max, double mask_frac,
                                                          int* __restrict__ hist, unsigned short* __restrict__ keys) {
  __shared__ int skey[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int in_stride = tl.transposed ? tl.rows : 1;
  const bool has_mask = mask_max != nullptr;
  const float thresh = has_mask ? (float)(mask_frac * mask_max[0]) : 0.0f;
  const int c_off = tl.transposed ? lane / kTileR : lane, r_off = tl.transposed ? lane % kTileR : 0;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int local = w * 16 + i;
    const long long t = (long long)blockIdx.x * 64 + local;              // wave-uniform
    if (t >= tl.n_tiles) continue;
    const GnTile d = gn_decode_tile(tl, n_pix, t);
    float v = __builtin_huge_valf();
    if (c_off < d.nc && r_off < d.nr) {
      const float x = load_g<float>(g1, g_is_f64, d.in_base + (long long)c_off * in_stride + r_off);
      if (!(has_mask && x >= thresh)) v = (x == x) ? x : 0.0f;          // air pixels do not count; a NaN count sorts first
    }
    const bool any = __ballot(v < __builtin_huge_valf()) != 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    int key = kSortBuckets - 1;                                          // all air: last
    if (any) {
      key = (v > 0.0f) ? (int)(__float_as_uint(v) >> 21) : 0;            // ascending counts = descending attenuation; <= 0: first
      key = key > kSortBuckets - 2 ? kSortBuckets - 2 : key;
    }
    if (lane == 0) {
      skey[local] = key;
      keys[t] = (unsigned short)key;
    }
  }
  __syncthreads();
  if (w == 0) {
    const long long t = (long long)blockIdx.x * 64 + lane;
    const bool valid = t < tl.n_tiles;
    int unused;
    wave_grouped_add(hist, valid ? skey[lane] : 0, valid, &unused);
  }
}

// exclusive scan of the bucket counts, in place (one workgroup of 1024: 2 buckets per thread)
__global__ __launch_bounds__(1024) void gn_tile_scan_kernel(int* __restrict__ hist) {
  __shared__ int wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c0 = hist[2 * tid], c1 = hist[2 * tid + 1];
  int x = c0 + c1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o, 64);
    if (lane >= o) x += y;
  }
  if (lane == 63) wave_tot[w] = x;
  __syncthreads();
  int base = 0;
  for (int k = 0; k < w; ++k) base += wave_tot[k];
  const int excl = base + x - (c0 + c1);
  hist[2 * tid] = excl;
  hist[2 * tid + 1] = excl + c0;
}

__global__ __launch_bounds__(256) void gn_tile_scatter_kernel(int n_tiles, int* __restrict__ offs,
                                                              const unsigned short* __restrict__ keys, int* __restrict__ order) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const bool valid = t < n_tiles;
  const int key = valid ? (int)keys[t] : 0;
  int base;
  const int rank = wave_grouped_add(offs, key, valid, &base);
  if (valid) order[base + rank] = t;
}

// The exit logic of one Newton step, shared by the lane-refill and the cooperative kernel: given the state before the step
// (a0, a1), the state after it (n0, n1), the history and the lane's iteration counter, either advance (returns true) or end
// the pixel with (a0, a1) = what all n_iters iterations would return (repeated state) / the converged state (tolerance).
__device__ __forceinline__ bool gn_exit_or_advance(double n0, double n1, int n_iters, int exact_exit, double stop_tol,
                                                   double& a0, double& a1, int& it, long long (&h0)[kGnHistory],
                                                   long long (&h1)[kGnHistory], bool* by_rule = nullptr, bool confirm = false) {
  int hit = -2;
  if (exact_exit) {
    const long long b0 = __double_as_longlong(n0), b1 = __double_as_longlong(n1);
    if (b0 == __double_as_longlong(a0) && b1 == __double_as_longlong(a1)) hit = -1;
#pragma unroll
    for (int k = kGnHistory - 1; k >= 0; --k)
      if (k < it && b0 == h0[k] && b1 == h1[k] && hit != -1) hit = k;
  }
  const bool converged = gn_converged(stop_tol, a0, a1, n0, n1, __longlong_as_double(h0[0]), __longlong_as_double(h1[0]), it, confirm,
                                      __longlong_as_double(h0[1]), __longlong_as_double(h1[1]));
  if (by_rule) *by_rule = converged;
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
    for (int k = 0; k < kGnHistory; ++k)
      if (slot == k) { f0 = __longlong_as_double(h0[k]); f1 = __longlong_as_double(h1[k]); }
  }
#pragma unroll
  for (int k = kGnHistory - 1; k > 0; --k) {
    h0[k] = advance ? h0[k - 1] : h0[k];
    h1[k] = advance ? h1[k - 1] : h1[k];
  }
  h0[0] = advance ? __double_as_longlong(a0) : h0[0];
  h1[0] = advance ? __double_as_longlong(a1) : h1[0];
  a0 = advance ? n0 : f0;
  a1 = advance ? n1 : f1;
  it += advance ? 1 : 0;
  return advance;
}

// ---- THE SHORT CUT past the reference's walk from 1e-6 (dexct_gn_options.pass = DEXCT_GN_PASS_SHORTCUT, .start) ---------------
//
// THE GATE.  What the reference returns is a function of a pixel's two counts alone: the state its iteration reaches from
// 1e-6 in n_iters steps - the fixed point only if it gets there in time, and, where noisy counts admit several fixed
// points, the one ITS walk ends at.  So the short cut is laid out in DATA space.  When the tables of a pair of spectra are
// prepared, the host runs the reference's iteration (this library's single launch, full tables, from 1e-6, counting steps:
// DEXCT_GN_PASS_COUNT) on the counts at the corners of a cell grid over (ln u0, u1 / u0), u_k = ln(air_k / g_k) / log_range,
// and records where it ends and after how many steps.  The start array carries, per corner, that fixed point; per cell, the
// number of steps a pixel needs (the largest count among the corners of the cell and of its eight neighbours, plus a margin;
// infinity where a corner did not end by the tolerance rule, or where the corners' fixed points do not vary smoothly - a
// boundary between two basins crosses the cell) and an acceptance radius (a hundredth of the spread of the corners' fixed
// points).  A pixel takes the short cut iff its counts fall in a cell with n_iters >= that number; it starts from the
// sextic (6 x 6 Lagrange) interpolant s of the corners' fixed points - a point on the REFERENCE'S branch, 1e-10 of |a| from the pixel's own
// fixed point - and its result is accepted only within the radius of s.  Every other pixel (few steps asked for, counts
// outside the grid, another fixed point, NaN) is solved the reference's way.
// Layout: [0],[1] unattenuated signals; [2] 1 / log_range; [3] cells per axis n; [4] ln of the smallest u0 of the grid;
// [5] cells per unit of ln u0; [6] smallest ratio u1 / u0 of the grid; [7] cells per unit of the ratio; [8],[9] ln of [0],[1];
// [10] 2 if the tables of the one-step acceptance follow the cells (below), else 0; [11] reserved; then the corners' fixed points
// as pairs (a0, a1)[(n+1)^2] (row = index along ln u0), then per cell the pair (need, radius)[n^2], then - [10] = 2 - per cell the
// pair (kappa, eps)[n^2]; the array is 16-byte aligned (pairs are read with one load).
//
// ONE STEP INSTEAD OF TWO (DEXCT_GN_FLAG_ONE_STEP).  A step from a start value s at distance e0 of the fixed point a* that
// provably leaves  e1 <= eps e0 + kappa e0^2  (the chord step above; round 5: a Gauss-Newton step with eps = 0) measures e0 by its
// own length, d1 = |m - s| >= e0 - e1.  The host tabulates (kappa, eps) per cell - kappa from the second derivatives of the
// misfit at the tabulated fixed points (2.5 x the largest value at the corners of the cell and of the eight around it), eps
// from the table's own gradient against the exact Jacobians at the cell's corners and centre (4 x the largest of the cell and the
// eight around it); infinity where a corner or centre does not count.  A pixel whose step satisfies
//     (kappa d1 + eps) d1 <= stop_tol / 4 * max(min(|a0|, |a1|), 1)
// has what the tolerance rule asks of two steps - a bound on the distance it still has to go, below stop_tol / 4 of its size
// (here even of its SMALLER component: the one-step results are compared with the exact count per component) -
// from one, and ends there; every other pixel goes on with full Newton steps and the rule, as before.  With the sextic
// interpolant d1 is 1e-11 of |a| at the median point of the data plane (5e-10 at the 90th percentile, 1e-8 at the 99th: the thick
// end at the edges of the physical ratios), kappa |a| is 100 on average, eps 2e-8 at the median point (2e-7 / 1e-5 at the 90th /
// 99th percentile: it shrinks and grows with d1): ~99 % of the plane passes with orders to spare, the rest takes full Newton steps.
constexpr int kStartHeader = 12;
#ifndef DEXCT_GN_INTERP_UNROLL
#define DEXCT_GN_INTERP_UNROLL 2      // rows of the 6 x 6 interpolation per loop trip (A/B: tools/probes/build_variant.sh)
#endif

// ln(x) for a positive, normal double: the hardware's float32 logarithm as a first guess y0 (|error| < 1e-5), then one
// Newton step on the table-driven exponential above: ln x = y0 + ln(x e^-y0) = y0 + r - r^2 / 2 with r = x e^-y0 - 1, |r| <
// 1e-5 (r^3 / 3 < 1e-16).  About 22 instructions against the 90 of the library logarithm; absolute error ~2e-16 + 1 ulp.
// Zero, negative, subnormal-in-float32, infinite or NaN arguments return NaN or +-inf (the gate then closes: every comparison
// with them is false).
__device__ __forceinline__ double log_pos(double x, const double* __restrict__ lds_pow) {
#ifdef DEXCT_GN_LIBM_LOG
  return log(x);
#else
  const double y0 = (double)(__builtin_amdgcn_logf((float)x) * 0.693147180559945309f);
  const double r = fma(x, exp_tab(-y0 * kExpScale, lds_pow), -1.0);
  return y0 + fma(-0.5 * r, r, r);
#endif
}

// What the chord step needs of the pixel's cell: the one-step pair and the inverse log-Jacobian at the pixel's place, as
// Bx_p = d a_p / d x * (-1 / (log_range u0)), Bt_p = d a_p / d t * (-1 / (log_range u0)) (x = ln u0, t = u1 / u0 - see gn_start)
struct GnCell { int idx; double t, Bx[2], Bt[2]; };      // idx: the cell, for its pairs (need, radius) and (kappa, eps) AFTER the energy loop

// DERIV (the chord step, round 6): besides the interpolant also ITS DERIVATIVES along the two coordinates, from the same 36
// loads - the fixed point as a function of the data-plane coordinates IS the inverse of the forward model, so the gradient of
// the table is the inverse Jacobian the chord step multiplies its residuals with: d a* / d ln g_k = (d a* / d (x, t)) (d (x, t) /
// d ln g_k), the first factor from the derivative weights of the Lagrange basis, the second analytic (u_k = (ln air_k - ln g_k) /
// log_range, x = ln u0, t = u1 / u0:  dx / d ln g0 = -1 / (log_range u0),  dt / d ln g0 = t / (log_range u0),  dt / d ln g1 =
// -1 / (log_range u0)).  No table of Jacobians, no load; its error shrinks with the interpolant's own (eps ~ e0 / cell x cond):
// 2e-8 at the median point of the plane, where a float32 table of the inverse stopped at 1e-5 (rounding x cond(L) = 200).
template <bool DERIV = false>
__device__ __forceinline__ bool gn_start(const double* __restrict__ start, const double* __restrict__ lds_pow, int n_iters,
                                         double g0, double g1, double& s0, double& s1, double& radius, GnCell* cell_out = nullptr) {
#ifdef DEXCT_GN_LIBM_LOG          // (round 4's arithmetic, kept for one bit-for-bit comparison of the two kernels)
  const double u0 = log(start[0] / g0) * start[2], u1 = log(start[1] / g1) * start[2];
  const double ru0 = 1.0 / u0;
  const double t = u1 / u0;
#else
  const double u0 = (start[8] - log_pos(g0, lds_pow)) * start[2], u1 = (start[9] - log_pos(g1, lds_pow)) * start[2];
  const double ru0 = rcp_f64(u0);
  const double t = u1 * ru0;
#endif
  const int n = (int)start[3];
  const double fx = (log_pos(u0, lds_pow) - start[4]) * start[5], fy = (t - start[6]) * start[7];
  bool ok = fx >= 0.0 && fy >= 0.0 && fx < (double)n && fy < (double)n;        // (NaN compares false)
  const int i = ok ? (int)fx : 0, j = ok ? (int)fy : 0;
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2* __restrict__ roots = reinterpret_cast<const d2*>(start + kStartHeader);       // (a0, a1) per corner
  const d2* __restrict__ cells = roots + (n + 1) * (n + 1);                               // (need, radius) per cell
  const d2 cell = cells[i * n + j];
  ok = ok && (double)n_iters >= cell.x;
  // 6 x 6 Lagrange interpolation of the corners' fixed points (the corners of the 5 x 5 cells the step table vouches for; cells
  // within two of the border of the grid are closed by the host).  The fixed point is an analytic function of (ln u0, u1 / u0);
  // at 256 cells per axis the sextic interpolant is within 1e-11 of |a| of the pixel's own fixed point at the median point of the
  // plane, 5e-10 at the 90th percentile (the Catmull-Rom interpolant of round 4, third order: 2e-6; tools/probes/gn_interp_cpu.py)
  // - close enough for ONE step to land at rounding level with a proven bound, and it costs 36 loads and 100 - 250 FMAs against a
  // step's 2 500.
  const double wx = fx - (double)i, wy = fy - (double)j;
  auto weights = [](double t_, double (&w)[6], double (&dw)[6]) {
    // nodes -2 .. 3: w_a = prod_(b != a) (t - x_b) / (x_a - x_b), by prefix and suffix products of p_b = t - x_b; DERIV: and
    // dw_a / dt by the product rule on the same recurrences (l_(a+1) = l_a p_a: l'_(a+1) = l'_a p_a + l_a)
    const double p0 = t_ + 2.0, p1 = t_ + 1.0, p2 = t_, p3 = t_ - 1.0, p4 = t_ - 2.0, p5 = t_ - 3.0;
    const double l1 = p0, l2 = l1 * p1, l3 = l2 * p2, l4 = l3 * p3, l5 = l4 * p4;          // prod_(b < a) p_b
    const double r4 = p5, r3 = r4 * p4, r2 = r3 * p3, r1 = r2 * p2, r0 = r1 * p1;          // prod_(b > a) p_b
    w[0] = r0 * (-1.0 / 120.0);
    w[1] = l1 * r1 * (1.0 / 24.0);
    w[2] = l2 * r2 * (-1.0 / 12.0);
    w[3] = l3 * r3 * (1.0 / 12.0);
    w[4] = l4 * r4 * (-1.0 / 24.0);
    w[5] = l5 * (1.0 / 120.0);
    if (DERIV) {
      const double dl1 = 1.0, dl2 = fma(dl1, p1, l1), dl3 = fma(dl2, p2, l2), dl4 = fma(dl3, p3, l3), dl5 = fma(dl4, p4, l4);
      const double dr4 = 1.0, dr3 = fma(dr4, p4, r4), dr2 = fma(dr3, p3, r3), dr1 = fma(dr2, p2, r2), dr0 = fma(dr1, p1, r1);
      dw[0] = dr0 * (-1.0 / 120.0);
      dw[1] = fma(dl1, r1, l1 * dr1) * (1.0 / 24.0);
      dw[2] = fma(dl2, r2, l2 * dr2) * (-1.0 / 12.0);
      dw[3] = fma(dl3, r3, l3 * dr3) * (1.0 / 12.0);
      dw[4] = fma(dl4, r4, l4 * dr4) * (-1.0 / 24.0);
      dw[5] = dl5 * (1.0 / 120.0);
    }
  };
  double cx[6], cy[6], dx[6], dy[6];
  weights(wx, cx, dx);
  weights(wy, cy, dy);
  const int i0c = i > 2 ? i - 2 : 0, j0c = j > 2 ? j - 2 : 0;               // (closed border cells never get here with ok)
  const int base = (i0c <= n - 5 ? i0c : n - 5) * (n + 1) + (j0c <= n - 5 ? j0c : n - 5);
  s0 = 0.0;
  s1 = 0.0;
  double ax0 = 0.0, ax1 = 0.0, at0 = 0.0, at1 = 0.0;                         // DERIV: d s / d fx, d s / d fy (cell units)
  // (rows per trip: 2 - twelve loads in flight, not thirty-six: registers at the kernel's tightest spot; 1 with the derivatives,
  // whose four more row sums and twelve more weights spilled 8 registers at 2)
#pragma unroll DERIV ? 1 : DEXCT_GN_INTERP_UNROLL
  for (int p = 0; p < 6; ++p) {
    double ra = 0.0, rb = 0.0, da = 0.0, db = 0.0;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const d2 r = roots[base + p * (n + 1) + q];
      ra = fma(cy[q], r.x, ra);
      rb = fma(cy[q], r.y, rb);
      if (DERIV) {
        da = fma(dy[q], r.x, da);
        db = fma(dy[q], r.y, db);
      }
    }
    s0 = fma(cx[p], ra, s0);
    s1 = fma(cx[p], rb, s1);
    if (DERIV) {
      ax0 = fma(dx[p], ra, ax0);
      ax1 = fma(dx[p], rb, ax1);
      at0 = fma(cx[p], da, at0);
      at1 = fma(cx[p], db, at1);
    }
  }
  radius = cell.y;
  if (DERIV) {
    const double kx = -(start[5] * start[2]) * ru0, kt = -(start[7] * start[2]) * ru0;     // cells per unit x (t) x d x (t) / d ln g
    *cell_out = GnCell{i * n + j, t, {ax0 * kx, ax1 * kx}, {at0 * kt, at1 * kt}};
  }
  return ok;
}

// (radius, kappa, eps) of cell idx: read AFTER the energy loop of the chord step (registers), not carried through it
__device__ __forceinline__ void gn_cell_pairs(const double* __restrict__ start, int idx, double& radius, double& kappa, double& eps) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  const int n = (int)start[3];
  const d2* __restrict__ cells = reinterpret_cast<const d2*>(start + kStartHeader) + (n + 1) * (n + 1);
  const d2 ke = start[10] == 2.0 ? (cells + n * n)[idx] : d2{__builtin_huge_val(), __builtin_huge_val()};
  radius = cells[idx].y;
  kappa = ke.x;
  eps = ke.y;
}

// float64, one shared spectrum - the single launch from the reference's start value - with lane refill.  The exits end
// pixels at very different iterations (from 10 to all of n_iters), and a wave is as slow as its slowest lane.  Here every
// lane whose pixel has ended takes the next pixel of the wave's current tile, and a wave whose tile is handed out fetches the
// next tile from a global counter at once, while its other lanes still iterate: all 64 lanes keep iterating until the
// sinogram is used up, the load balances itself over CUs and XCDs.  The energy loops stay wave-uniform (scalar table loads)
// because the tables do not depend on the pixel.  Pixels are independent problems: results are bit-identical to gn_kernel's
// in any order.  Register allocation: 4 waves per SIMD, 110 VGPRs, no scratch.
//
// COUNT (dexct_gn_options.pass = DEXCT_GN_PASS_COUNT): the same iteration, counting steps: besides the result, `iters` receives
// per pixel (result order) the number of steps after which the tolerance rule ended it, or 255 when it ended any other way
// (n_iters reached, repeated state, NaN).  Run by the host on a grid of counts to tabulate where the reference's walk ends
// (the gate of the short cut, above).
template <bool COUNT>
__global__ __launch_bounds__(kGnBlock, 4) void gn_refill_kernel(const void* __restrict__ g1, const void* __restrict__ g2,
                                                             int g_is_f64, long long n_pix, const double* __restrict__ ws,
                                                             int n_e, int n_iters, GnTiling tl,
                                                             const double* __restrict__ mask_max, double mask_frac,
                                                             int flags, double stop_tol,      // flags: bit 0 exact repeated-state exit, bit 1 sorted hand-out, bit 2 the rule asks for two contracting steps, bits 8..19: tiles per queue reservation
                                                             double* __restrict__ out_a,
                                                             unsigned long long* __restrict__ counters,
                                                             unsigned char* __restrict__ iters) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  // `counters` = the workspace words 9.. (executed, progress, queue head, stalls): ONE pointer, and the tile order is found
  // from it too (the table pointer `ws` stays read-only for the compiler: its loads are scalar loads)
  auto counter = [&](int k) { return counters + (k - 9); };
  __shared__ double lds_pow[kPowN];                                      // 16 KB
  __shared__ d2 lds_out[kGnBlock / kWave][kSlots * kTilePix];            // 12 KB: with the table 28 KB = 5 workgroups per CU
  __shared__ unsigned char lds_it[COUNT ? kGnBlock / kWave : 1][COUNT ? kSlots * kTilePix : 1];
  unsigned char* __restrict__ my_it = lds_it[COUNT ? (threadIdx.x >> 6) : 0];
  for (int j = threadIdx.x; j < kPowN; j += kGnBlock) lds_pow[j] = pow_entry(j);
  __syncthreads();                        // the only barrier: waves leave the loop below independently
  const EnergyClasses ec{(int)ws[1], (int)ws[4], (int)ws[2], (int)ws[5], (int)ws[3], (int)ws[6], ws[7], ws[8]};
  const double* __restrict__ tab = ws + kWsHeader;
  d2* __restrict__ my_out = lds_out[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  const bool has_mask = mask_max != nullptr;
  const double thresh = has_mask ? mask_frac * mask_max[0] : 0.0;
  const int in_stride = tl.transposed ? tl.rows : 1;                     // input distance of neighbouring channels of a tile
  const int out_stride = tl.transposed ? tl.channels : 0;                // output distance of neighbouring rows of a tile

  // wave-uniform state of the tiles in flight
  int tid[kSlots];                        // tile held by a slot, -1: free
#pragma unroll
  for (int k = 0; k < kSlots; ++k) tid[k] = -1;
  unsigned pend = 0u;                     // results a slot still waits for, one byte per slot (<= 64)
  int cur = 0, next_j = kTilePix;         // the slot being handed out and its next rank; next_j >= 64: fetch a tile first
  GnTile ct{0, 0, 0, 0};                  // the tile being handed out
  bool exhausted = false;
  unsigned n_exec = 0, n_stall = 0;       // Newton steps executed; lane-steps spent waiting for a free slot (diagnostic)
  const int batch = ((flags >> 8) & 0xFFF) > 0 ? ((flags >> 8) & 0xFFF) : 1;      // queue positions reserved per atomic (flags bits 8..19)
  int q_next = 0, q_end = 0;              // the reserved positions not yet handed out (n_tiles < 2^31)
  unsigned handed = 0u;                   // pixels of the tiles handed out since the last reservation (progress word)

  int ent = -1;                           // slot * 64 + place of this lane's result in the slot, -1: no pixel
  double a0 = 1e-6, a1 = 1e-6, gd0 = 1.0, gd1 = 1.0;
  int it = 0;
  long long h0[kGnHistory], h1[kGnHistory];
#pragma unroll
  for (int k = 0; k < kGnHistory; ++k) { h0[k] = 0; h1[k] = 0; }

  // `done_lanes` have just put their result into LDS (my_slot: the slot of this lane's result): count them off their
  // slots; a slot whose last result has arrived is written - one wave-wide store of whole lines - and freed
  auto settle = [&](unsigned long long done_lanes, int my_slot) {
    unsigned complete = 0u;
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
      const unsigned long long m = __ballot(my_slot == k) & done_lanes;
      pend -= (unsigned)__popcll(m) << (8 * k);
      complete |= (tid[k] >= 0 && ((pend >> (8 * k)) & 0xFFu) == 0u) ? (1u << k) : 0u;
    }
    while (complete != 0u) {                          // (wave-uniform; one copy of the store code for all slots)
      const int k = __builtin_ctz(complete);
      complete &= complete - 1u;
      int tile = tid[0];
#pragma unroll
      for (int q = 1; q < kSlots; ++q) tile = k == q ? tid[q] : tile;
#pragma unroll
      for (int q = 0; q < kSlots; ++q) tid[q] = k == q ? -1 : tid[q];
      if (k == cur) next_j = kTilePix;                // every valid pixel of the tile was handed out: the rest of its ranks is padding
      __builtin_amdgcn_wave_barrier();
      const GnTile d = gn_decode_tile(tl, n_pix, (long long)tile);
      const int c_off = tl.transposed ? (lane & (kTileC - 1)) : lane, r_off = tl.transposed ? (lane >> kTileCLog2) : 0;
      if (c_off < d.nc && r_off < d.nr) {
        const d2 v = my_out[k * kTilePix + lane];
        __builtin_nontemporal_store(v, reinterpret_cast<d2*>(out_a) + (d.out_base + (long long)r_off * out_stride + c_off));
        if (COUNT) iters[d.out_base + (long long)r_off * out_stride + c_off] = my_it[k * kTilePix + lane];
      }
      __builtin_amdgcn_wave_barrier();
    }
  };

  for (;;) {
    unsigned long long want = __ballot(ent < 0);
    while (want != 0ull) {
      if (next_j >= kTilePix) {
        if (exhausted) break;
        int s = -1;
#pragma unroll
        for (int k = kSlots - 1; k >= 0; --k) s = tid[k] < 0 ? k : s;
        if (s < 0) { n_stall += (unsigned)__popcll(want); break; }     // every slot still waits for a straggler
        if (q_next >= q_end) {
          // reserve the next `batch` queue positions: ONE atomic on the queue head for `batch` tiles, and one for the progress
          // word (the pixels of the tiles handed out since the last reservation)
          long long t = 0;
          if (lane == 0) {
            t = (long long)atomicAdd(counter(11), (unsigned long long)batch);
            if (handed) atomicAdd(counter(10), (unsigned long long)handed);        // progress: handed to a wave
          }
          handed = 0u;
          t = ((long long)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) |
              (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)t);
          if (t >= tl.n_tiles) { exhausted = true; break; }
          q_next = (int)t;
          q_end = t + batch < tl.n_tiles ? (int)t + batch : (int)tl.n_tiles;
        }
        long long t = q_next++;
        if (flags & 2) t = gn_tile_order(counters, n_e)[t];     // queue position -> tile (thick tiles first, see gn_tile_key_kernel)
        ct = gn_decode_tile(tl, n_pix, t);
        const int n_valid = ct.nr * ct.nc;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) tid[k] = k == s ? (int)t : tid[k];
        pend += (unsigned)n_valid << (8 * s);
        cur = s;
        next_j = 0;
        handed += (unsigned)n_valid;
      }
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
      const int j = next_j + rank;
      bool done_now = false;
      if (ent < 0 && j < kTilePix) {
        // consecutive lanes take consecutive rows of a channel (transposed) / consecutive pixels (plain)
        const int c_off = tl.transposed ? j / kTileR : j, r_off = tl.transposed ? j % kTileR : 0;
        if (c_off < ct.nc && r_off < ct.nr) {
          const long long np = ct.in_base + (long long)c_off * in_stride + r_off;
          const int place = cur * kTilePix + (tl.transposed ? r_off * kTileC + c_off : j);
          gd0 = load_g<double>(g1, g_is_f64, np);
          gd1 = load_g<double>(g2, g_is_f64, np);
          if (has_mask && gd0 >= thresh) {             // air (matdecomp.py:195-196, :204-205): 0, not iterated
            my_out[place] = d2{0.0, 0.0};
            if (COUNT) my_it[place] = 255;
            done_now = true;
          } else if (n_iters <= 0) {
            my_out[place] = d2{1e-6, 1e-6};
            if (COUNT) my_it[place] = 255;
            done_now = true;
          } else {
            ent = place; a0 = 1e-6; a1 = 1e-6; it = 0;
          }
        }
      }
      next_j += __popcll(want);
      const unsigned long long dn = __ballot(done_now);
      if (dn != 0ull) settle(dn, cur);
      want = __ballot(ent < 0);                      // air pixels leave their lane wanting: it is served again at once
    }
    const unsigned long long busy = __ballot(ent >= 0);
    if (busy == 0ull) {
      if (exhausted) break;                          // (no busy lane and a tile to hand out or to fetch: around again)
      continue;
    }
    n_exec += (unsigned)__popcll(busy);              // wave-uniform (scalar) count of Newton steps run
    double n0 = a0, n1 = a1;
    newton_step_f64(tab, lds_pow, ec, gd0, gd1, n0, n1);      // idle lanes repeat their last pixel's step; unused
    // same exit rule as gn_kernel: s_{it+1} equal to s_it (fixed point) or to hist[k] = s_{it-1-k} (cycle of
    // k + 2 states) determines every later iterate.  Written with selects instead of branches; idle lanes run
    // through it too and are ignored.
    bool by_rule = false;
    const bool advance = gn_exit_or_advance(n0, n1, n_iters, flags & 1, stop_tol, a0, a1, it, h0, h1,
                                            COUNT ? &by_rule : nullptr, (flags & 4) != 0);
    const bool fin = ent >= 0 && (!advance || it >= n_iters);
    const unsigned long long fb = __ballot(fin);
    if (fb != 0ull) {
      if (COUNT && fin) my_it[ent] = (unsigned char)(by_rule ? (it + 1 < 255 ? it + 1 : 254) : 255);
      if (fin) my_out[ent] = d2{a0, a1};
      const int my_slot = ent >> 6;                  // (-1 for lanes without a pixel: matches no slot)
      if (fin) ent = -1;
      settle(fb, my_slot);
    }
  }
  if (lane == 0) {
    atomicAdd(counter(9), (unsigned long long)n_exec);            // one atomic per wave
    if (n_stall) atomicAdd(counter(12), (unsigned long long)n_stall);
    if (handed) atomicAdd(counter(10), (unsigned long long)handed);
  }
}

// THE SHORT-CUT KERNEL (dexct_gn_options.pass = DEXCT_GN_PASS_SHORTCUT; the default of get_basismat_sinos since round 4, this
// form since round 5).  On the short cut every pixel takes the same one or two steps, so there is nothing to balance and the
// machinery of the refill kernel (result slots, hand-out loop, repeated-state history: a fifth of round 4's launch) is in the
// way.  Here a wave works through tiles in lock step:
//   FAST PATH, straight-line code for the 64 pixels of a tile at once: counts in, air mask, gate + start value (gn_start),
//     STEPS steps on the full tables, the evidence of convergence - STEPS = 2: the tolerance rule on the second step (the FULL
//     model has converged to stop_tol); STEPS = 1 (DEXCT_GN_FLAG_ONE_STEP): the cell's tabulated kappa, see kStartHeader - the
//     acceptance radius, results out through LDS as whole 64-byte runs in the reference's order.
//   STASH: a pixel the fast path does not finish - closed cell (walk from the reference's start value 1e-6 with all n_iters
//     steps), no evidence yet (it continues where it is), result outside the radius (walk) - is put aside in LDS (96 entries
//     per wave) and the wave moves on.
//   DRAIN: when 32 entries have gathered (and at the end) the wave solves them with the general iteration - one lane per
//     entry, refilled from the stash, every exit of gn_exit_or_advance, the acceptance test for continued pixels - and stores
//     those results pixel by pixel.
// STEPS = 2: per pixel the sequence of states and the exit taken are those of round 4's refill kernel with start values
// (bit-identical results, tools/probes/gn_shortcut_ab.py); only who computes them when has changed.
constexpr int kStashCap = 96, kStashDrain = 32;
constexpr long long kStashCont = 1ll << 62;          // entry: the pixel continues from the stored states (else: walks from 1e-6)
constexpr long long kStashOne = 1ll << 61;           // ... after ONE step (states: sp = start value s, sa = after the step); else after two

template <int STEPS>
__global__ __launch_bounds__(kGnBlock, 4) void gn_shortcut_kernel(const void* __restrict__ g1, const void* __restrict__ g2,
                                                               int g_is_f64, long long n_pix, const double* __restrict__ ws,
                                                               int n_e, int n_iters, GnTiling tl,
                                                               const double* __restrict__ mask_max, double mask_frac,
                                                               int flags, double stop_tol, double* __restrict__ out_a,     // flags: as gn_refill_kernel
                                                               unsigned long long* __restrict__ counters,
                                                               const double* __restrict__ start) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  auto counter = [&](int k) { return counters + (k - 9); };
  __shared__ double lds_pow[kPowN];                                      // 16 KB
  __shared__ d2 lds_out[kGnBlock / kWave][kTilePix];                     // 4 KB: one tile of results per wave
  __shared__ long long lds_snp[kGnBlock / kWave][kStashCap];             // 3 KB: the stash - input index of the pixel | kind,
  __shared__ d2 lds_sa[kGnBlock / kWave][kStashCap];                     // 6 KB: its state,
  __shared__ d2 lds_sp[kGnBlock / kWave][kStashCap];                     // 6 KB: the state before,
  __shared__ double lds_srad[kGnBlock / kWave][kStashCap];               // 3 KB: its acceptance radius  (38 KB: 4 workgroups per CU)
  for (int j = threadIdx.x; j < kPowN; j += kGnBlock) lds_pow[j] = pow_entry(j);
  __syncthreads();                        // the only barrier: waves work independently from here on
  const EnergyClasses ec{(int)ws[1], (int)ws[4], (int)ws[2], (int)ws[5], (int)ws[3], (int)ws[6], ws[7], ws[8]};
  const double* __restrict__ tab = ws + kWsHeader;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  d2* __restrict__ my_out = lds_out[wv];
  long long* __restrict__ snp = lds_snp[wv];
  d2* __restrict__ sa = lds_sa[wv];
  d2* __restrict__ sp = lds_sp[wv];
  double* __restrict__ srad = lds_srad[wv];
  const bool has_mask = mask_max != nullptr;
  const double thresh = has_mask ? mask_frac * mask_max[0] : 0.0;
  const int in_stride = tl.transposed ? tl.rows : 1, out_stride = tl.transposed ? tl.channels : 0;
  const bool exact_exit = (flags & 1) != 0, confirm_walk = (flags & 4) != 0;
  const int batch = ((flags >> 8) & 0xFFF) > 0 ? ((flags >> 8) & 0xFFF) : 1;
  int q_next = 0, q_end = 0;
  bool exhausted = false;
  unsigned handed = 0u;
  unsigned long long n_exec = 0ull;
  int n_stash = 0;                        // wave-uniform

  // the next tile of this wave, -1 when the queue is used up (one atomic per `batch` tiles, as in gn_refill_kernel)
  auto next_tile = [&]() -> long long {
    if (exhausted) return -1;
    if (q_next >= q_end) {
      long long t = 0;
      if (lane == 0) {
        t = (long long)atomicAdd(counter(11), (unsigned long long)batch);
        if (handed) atomicAdd(counter(10), (unsigned long long)handed);
      }
      handed = 0u;
      t = ((long long)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) |
          (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)t);
      if (t >= tl.n_tiles) { exhausted = true; return -1; }
      q_next = (int)t;
      q_end = t + batch < tl.n_tiles ? (int)t + batch : (int)tl.n_tiles;
    }
    return (long long)q_next++;
  };

  // DRAIN: the general iteration on the entries of the stash.  A continued pixel goes on from its state sa with sp as the
  // state before.  After ONE step (sp = the start value s, sa = n): exactly the refill kernel's pixel at it = 1, accepted within
  // the radius of s.  After TWO (sp = n, sa = m; s not kept): it = 2 with the budget of n_iters; accepted within srad = radius -
  // |m - s| of m (hence within the radius of s); s is not entered into the history - a cycle through it is found one period
  // later, with the same result.  (Neither s nor the table is touched here: registers.)
  auto drain = [&]() {
    int head = 0;
    long long po = -1;                    // where this lane's result goes, -1: no pixel
    double a0 = 1e-6, a1 = 1e-6, gd0 = 1.0, gd1 = 1.0;
    int it = 0, warm = -1;                // warm >= 0: a continued pixel: its entry, + 256 if it continues after ONE step
    long long h0[kGnHistory], h1[kGnHistory];
#pragma unroll
    for (int k = 0; k < kGnHistory; ++k) { h0[k] = 0; h1[k] = 0; }
    for (;;) {
      const unsigned long long want = __ballot(po < 0);
      if (head < n_stash && want != 0ull) {
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
        const int idx = head + rank;
        if (po < 0 && idx < n_stash) {
          const long long e = snp[idx];
          const long long np = e & ~(kStashCont | kStashOne);
          gd0 = load_g<double>(g1, g_is_f64, np);
          gd1 = load_g<double>(g2, g_is_f64, np);
          po = gn_out_index(tl, np);
          it = 0; a0 = 1e-6; a1 = 1e-6; warm = -1;
          if (e & kStashCont) {
            const d2 va = sa[idx], vp = sp[idx];
            a0 = va.x; a1 = va.y;
            h0[0] = __double_as_longlong(vp.x); h1[0] = __double_as_longlong(vp.y);
            it = 1;                       // (two steps taken: as if sa followed sp directly, one iteration less allowed)
            warm = idx | ((e & kStashOne) ? 256 : 0);
          }
        }
        const int taken = (int)__popcll(want);
        head = head + taken < n_stash ? head + taken : n_stash;
      }
      const unsigned long long busy = __ballot(po >= 0);
      if (busy == 0ull) break;            // (head == n_stash here: an idle wave with entries left takes them above)
      n_exec += (unsigned long long)__popcll(busy);
      double n0 = a0, n1 = a1;
      newton_step_f64(tab, lds_pow, ec, gd0, gd1, n0, n1);
      const int bud = (warm >= 0 && (warm & 256) == 0) ? n_iters - 1 : n_iters;     // iterations allowed, counted like `it`
      const bool advance = gn_exit_or_advance(n0, n1, bud, exact_exit ? 1 : 0, stop_tol, a0, a1, it, h0, h1, nullptr, confirm_walk && warm < 0);
      bool fin = po >= 0 && (!advance || it >= bud);
      if (fin && warm >= 0) {
        // a continued pixel that used up its budget, ran into NaN or ended away from the reference's branch: the reference's
        // own solve instead
        const d2 vc = (warm & 256) ? sp[warm & 255] : sa[warm & 255];
        const bool redo = advance || !(fmax(fabs(a0 - vc.x), fabs(a1 - vc.y)) <= srad[warm & 255]);
        if (redo) { a0 = 1e-6; a1 = 1e-6; it = 0; fin = false; }
        warm = -1;
      }
      if (fin) {
        store_a(out_a, po, a0, a1);
        po = -1;
      }
    }
    n_stash = 0;
  };

  // (No software prefetch of the next tile's counts: the registers it holds across the arithmetic cost more - spills - than the
  // three other waves of the SIMD leave uncovered of one load's latency.)
  for (long long t = next_tile(); t >= 0; t = next_tile()) {
    const GnTile d = gn_decode_tile(tl, n_pix, t);
    handed += (unsigned)(d.nr * d.nc);
    // this lane's pixel of the tile on the input side (consecutive lanes: consecutive rows of a channel / consecutive pixels);
    // (these lane constants are formed where they are used - a few shifts per tile - rather than held in registers)
    const int ci = tl.transposed ? lane / kTileR : lane, ri = tl.transposed ? lane % kTileR : 0;
    const bool valid = ci < d.nc && ri < d.nr;
    const long long np = d.in_base + (long long)ci * in_stride + ri;
    const double gd0 = valid ? load_g<double>(g1, g_is_f64, np) : 1.0, gd1 = valid ? load_g<double>(g2, g_is_f64, np) : 1.0;
    const bool air = has_mask && gd0 >= thresh;             // (matdecomp.py:195-196, :204-205): 0, not iterated
    const bool act = valid && !air && n_iters > 0;
    d2 res = air ? d2{0.0, 0.0} : d2{1e-6, 1e-6};
    bool done = valid && !act;
    bool to_walk = false, to_cont = false;
    d2 st_a = d2{1e-6, 1e-6}, st_p = d2{0.0, 0.0};
    double st_rad = 0.0;
    if (__ballot(act) != 0ull) {
      double s0, s1, rad;
      GnCell cl{0, 0.0, {0.0, 0.0}, {0.0, 0.0}};
      double kap1 = __builtin_huge_val(), eps1 = __builtin_huge_val();
      const bool open = gn_start<STEPS == 1>(start, lds_pow, n_iters, gd0, gd1, s0, s1, rad, &cl) && n_iters >= 3;
      const bool fast = act && open;
      to_walk = act && !open;
      const unsigned long long fm = __ballot(fast);
      if (fm != 0ull) {
        n_exec += (unsigned long long)(STEPS * (int)__popcll(fm));
        double n0 = s0, n1 = s1, m0 = s0, m1 = s1;
        if (STEPS == 1) {                                                    // from s to m: the chord step (chord_residuals_f64)
          double c[2];
          chord_residuals_f64(tab, lds_pow, ec, gd0, gd1, s0, s1, c);
          // m = s + B c with B_p0 = Bx_p - t Bt_p, B_p1 = Bt_p (gn_start<DERIV>)
          const double ct = fma(-cl.t, c[0], c[1]);
          m0 = s0 + fma(cl.Bx[0], c[0], cl.Bt[0] * ct);
          m1 = s1 + fma(cl.Bx[1], c[0], cl.Bt[1] * ct);
          gn_cell_pairs(start, cl.idx, rad, kap1, eps1);
        } else {
#pragma nounroll
          for (int k = 0; k < STEPS; ++k) {                                  // step 1: from s to n, step 2: from n to m
            n0 = m0; n1 = m1;
            newton_step_f64(tab, lds_pow, ec, gd0, gd1, m0, m1);
          }
        }
        bool ended;
        double f0, f1;
        if (STEPS == 1) {
          // one step, from s to m: a fixed point, or the cell's (kappa, eps) vouch that what is left is below the tolerance (gn_start)
          const bool fixed1 = exact_exit && __double_as_longlong(m0) == __double_as_longlong(s0) &&
                              __double_as_longlong(m1) == __double_as_longlong(s1);
          // (the size that counts is the SMALLER component's, at least 1: what the bound leaves is then below stop_tol / 4 of
          // every component by itself - a ray through 37 cm of water has a second component of -0.6, and the comparisons
          // with the exact count hold 1e-12 per component)
          const double d1 = fmax(fabs(m0 - s0), fabs(m1 - s1)), size = fmax(fmin(fabs(m0), fabs(m1)), 1.0);
          const bool conv1 = fma(kap1, d1, eps1) * d1 <= (0.25 * stop_tol) * size;      // (NaN, inf: no)
          ended = fixed1 || conv1;
          f0 = fixed1 ? s0 : m0; f1 = fixed1 ? s1 : m1;
        } else {
          // the exits of gn_exit_or_advance for these two steps (no history yet: a fixed point at either step, the tolerance
          // rule at the second); anything else - the rule not met, a cycle through s - continues in the drain
          const bool fixed1 = exact_exit && __double_as_longlong(n0) == __double_as_longlong(s0) &&
                              __double_as_longlong(n1) == __double_as_longlong(s1);
          const bool fixed2 = exact_exit && __double_as_longlong(m0) == __double_as_longlong(n0) &&
                              __double_as_longlong(m1) == __double_as_longlong(n1);
          const bool cycle2 = exact_exit && __double_as_longlong(m0) == __double_as_longlong(s0) &&
                              __double_as_longlong(m1) == __double_as_longlong(s1);
          const bool conv = gn_converged(stop_tol, n0, n1, m0, m1, s0, s1, 1);
          ended = fixed1 || ((fixed2 || conv) && !cycle2);
          f0 = fixed1 ? s0 : (conv ? m0 : n0); f1 = fixed1 ? s1 : (conv ? m1 : n1);
        }
        const bool accept = fmax(fabs(f0 - s0), fabs(f1 - s1)) <= rad;        // (NaN: no)
        if (fast) {
          if (ended && accept) { res = d2{f0, f1}; done = true; }
          else if (ended) to_walk = true;                                     // another fixed point: the reference's own solve
          else if (STEPS == 1) { to_cont = true; st_a = d2{m0, m1}; st_p = d2{s0, s1}; st_rad = rad; }
          else {                                                              // goes on from m if m is well inside the radius of s
            st_rad = rad - fmax(fabs(m0 - s0), fabs(m1 - s1));
            if (st_rad > 0.0) { to_cont = true; st_a = d2{m0, m1}; st_p = d2{n0, n1}; }
            else to_walk = true;                                              // (NaN included)
          }
        }
      }
    }
    // the stash takes what is left
    const unsigned long long pm = __ballot(to_walk || to_cont);
    if (pm != 0ull) {
      const int idx = n_stash + __builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0u));
      if (to_walk || to_cont) {
        snp[idx] = np | (to_cont ? (kStashCont | (STEPS == 1 ? kStashOne : 0ll)) : 0ll);
        sa[idx] = st_a;
        sp[idx] = st_p;
        srad[idx] = st_rad;
      }
      n_stash += (int)__popcll(pm);
    }
    // results of the tile: through LDS into the order of the output, whole 64-byte runs
    const unsigned long long dm = __ballot(done);
    if (dm != 0ull) {
      // where this lane's result goes in the tile's LDS image, and the pixel this lane WRITES (consecutive lanes: consecutive
      // channels), computed by the input-side lane lane_in_of_out
      const int place = tl.transposed ? (lane % kTileR) * kTileC + lane / kTileR : lane;
      const int co = tl.transposed ? (lane & (kTileC - 1)) : lane, ro = tl.transposed ? (lane >> kTileCLog2) : 0;
      const int lane_in_of_out = tl.transposed ? co * kTileR + ro : lane;
      if (done) my_out[place] = res;
      __builtin_amdgcn_wave_barrier();
      if (co < d.nc && ro < d.nr && ((dm >> lane_in_of_out) & 1ull) != 0ull) {
        const d2 v = my_out[lane];
        __builtin_nontemporal_store(v, reinterpret_cast<d2*>(out_a) + (d.out_base + (long long)ro * out_stride + co));
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (n_stash > kStashDrain) drain();
  }
  if (n_stash > 0) drain();
  if (lane == 0) {
    atomicAdd(counter(9), n_exec);                                // one atomic per wave
    if (handed) atomicAdd(counter(10), (unsigned long long)handed);
  }
}

// COOPERATIVE Newton kernel for sinograms too small to fill the chip with one pixel per lane (the reference's own scan:
// 1200 x 800 x 1 row = 9.6e5 pixels, input/params.txt:21-22, for 3.3e5 resident lanes - the launch then lasts as long as the
// few pixels that need all n_iters iterations, each of them one lane's 3 900 instructions per iteration).  Here the kCoopWaves
// waves of a workgroup work on the SAME 64 pixels: every wave keeps the full state of all 64 (replicated, deterministic),
// sums its own share of the energies (wave-uniform energy index: the tables still arrive as scalar operands), the partial
// sums meet in LDS, and every wave adds them in the same fixed order and takes the same step: 2.6 x less latency per
// iteration for 1.5 x the instructions.  Lanes are refilled from the tile queue exactly as in gn_refill_kernel; a tile id is
// fetched by wave 0 and handed to the others through LDS.  The energy sums are formed in another order than in the
// one-lane kernel (per wave, then ((w0 + w1) + (w2 + w3))): results agree with it to rounding (and with the reference's
// goldens to 1e-9), not bit for bit; within this kernel the update is still a pure function of the state, so the exact
// repeated-state exit holds.
constexpr int kCoopWaves = 4;
constexpr long long kCoopBelowDefault = 100000;      // pixels; measured crossover (1.5e4: 0.31 vs 0.55 ms, 5e4: 0.37 vs 0.55,
                                                     // 1e5: 0.57 vs 0.58, 1.8e5: 0.75 vs 0.74, 9.6e5: 2.6 vs 2.2; profiles/r04_gn.md)

__global__ __launch_bounds__(kCoopWaves * kWave, 3) void gn_coop_kernel(const void* __restrict__ g1, const void* __restrict__ g2,
                                                                     int g_is_f64, long long n_pix, const double* __restrict__ ws,
                                                                     int n_e, int n_iters, GnTiling tl,
                                                                     const double* __restrict__ mask_max, double mask_frac,
                                                                     int flags, double stop_tol, double* __restrict__ out_a,
                                                                     unsigned long long* __restrict__ counters) {
  auto counter = [&](int k) { return counters + (k - 9); };
  __shared__ double lds_pow[kPowN];                                      // 16 KB
  __shared__ double lds_part[kCoopWaves][12][kWave];                     // 24 KB: the partial sums of one step
  __shared__ long long lds_tile[2];
  for (int j = threadIdx.x; j < kPowN; j += kCoopWaves * kWave) lds_pow[j] = pow_entry(j);
  __syncthreads();
  const EnergyClasses ec{(int)ws[1], (int)ws[4], (int)ws[2], (int)ws[5], (int)ws[3], (int)ws[6], ws[7], ws[8]};
  const double* __restrict__ tab = ws + kWsHeader;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const bool writer = wave == 0;                                         // wave 0 talks to memory (results, queue, counters)
  const bool has_mask = mask_max != nullptr;
  const double thresh = has_mask ? mask_frac * mask_max[0] : 0.0;
  const int in_stride = tl.transposed ? tl.rows : 1, out_stride = tl.transposed ? tl.channels : 0;

  // every wave holds the same values in all of these
  GnTile ct{0, 0, 0, 0};
  int next_j = kTilePix, fetches = 0;
  bool exhausted = false;
  unsigned n_exec = 0;
  long long pout = -1;                    // where this lane's result goes, -1: no pixel
  double a0 = 1e-6, a1 = 1e-6, gd0 = 1.0, gd1 = 1.0;
  int it = 0;
  long long h0[kGnHistory], h1[kGnHistory];
#pragma unroll
  for (int k = 0; k < kGnHistory; ++k) { h0[k] = 0; h1[k] = 0; }

  for (;;) {
    unsigned long long want = __ballot(pout < 0);
    while (want != 0ull) {
      if (next_j >= kTilePix) {
        if (exhausted) break;
        // one tile id for the whole workgroup: fetched by wave 0, read by all (two LDS words used in turn: a wave that runs
        // ahead to the NEXT fetch writes the other word, and cannot reach the one after before everybody has passed this barrier)
        if (writer && lane == 0) lds_tile[fetches & 1] = (long long)atomicAdd(counter(11), 1ull);
        __syncthreads();
        long long t = lds_tile[fetches & 1];
        t = ((long long)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) |
            (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)t);
        ++fetches;
        if (t >= tl.n_tiles) { exhausted = true; break; }
        if (flags & 2) t = gn_tile_order(counters, n_e)[t];
        ct = gn_decode_tile(tl, n_pix, t);
        next_j = 0;
        if (writer && lane == 0) atomicAdd(counter(10), (unsigned long long)(ct.nr * ct.nc));
      }
      const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(want >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)want, 0u));
      const int j = next_j + rank;
      if (pout < 0 && j < kTilePix) {
        const int c_off = tl.transposed ? j / kTileR : j, r_off = tl.transposed ? j % kTileR : 0;
        if (c_off < ct.nc && r_off < ct.nr) {
          const long long np = ct.in_base + (long long)c_off * in_stride + r_off;
          const long long po = ct.out_base + (long long)r_off * out_stride + c_off;
          gd0 = load_g<double>(g1, g_is_f64, np);
          gd1 = load_g<double>(g2, g_is_f64, np);
          if (has_mask && gd0 >= thresh) {             // air (matdecomp.py:195-196, :204-205): 0, not iterated
            if (writer) store_a(out_a, po, 0.0, 0.0);
          } else if (n_iters <= 0) {
            if (writer) store_a(out_a, po, 1e-6, 1e-6);
          } else {
            pout = po; a0 = 1e-6; a1 = 1e-6; it = 0;
          }
        }
      }
      next_j += __popcll(want);
      want = __ballot(pout < 0);
    }
    const unsigned long long busy = __ballot(pout >= 0);
    if (busy == 0ull) {
      if (exhausted) break;
      continue;
    }
    n_exec += (unsigned)__popcll(busy);
    // this wave's share of the energies
    GnSums ps;
    newton_sums_f64<kCoopWaves>(tab, lds_pow, ec, wave, a0, a1, ps);
    {
      double (*mine)[kWave] = lds_part[wave];
      mine[0][lane] = ps.nu[0];  mine[1][lane] = ps.nu[1];
      mine[2][lane] = ps.G0[0];  mine[3][lane] = ps.G0[1];
      mine[4][lane] = ps.G1[0];  mine[5][lane] = ps.G1[1];
      mine[6][lane] = ps.H00[0]; mine[7][lane] = ps.H00[1];
      mine[8][lane] = ps.H01[0]; mine[9][lane] = ps.H01[1];
      mine[10][lane] = ps.H11[0]; mine[11][lane] = ps.H11[1];
    }
    __syncthreads();
    double tot[12];
#pragma unroll
    for (int q = 0; q < 12; ++q)                    // the same order in every wave: identical bits everywhere
      tot[q] = (lds_part[0][q][lane] + lds_part[1][q][lane]) + (lds_part[2][q][lane] + lds_part[3][q][lane]);
    __syncthreads();                                // the partial sums may be overwritten from here on
    const GnSums s{{tot[0], tot[1]}, {tot[2], tot[3]}, {tot[4], tot[5]}, {tot[6], tot[7]}, {tot[8], tot[9]}, {tot[10], tot[11]}};
    double n0 = a0, n1 = a1;
    newton_solve_f64(s, gd0, gd1, n0, n1);
    const bool advance = gn_exit_or_advance(n0, n1, n_iters, flags & 1, stop_tol, a0, a1, it, h0, h1, nullptr, (flags & 4) != 0);
    if (pout >= 0 && (!advance || it >= n_iters)) {
      if (writer) store_a(out_a, pout, a0, a1);
      pout = -1;
    }
  }
  if (writer && lane == 0) atomicAdd(counter(9), (unsigned long long)n_exec);
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

// The environment is read ONCE per process (first call): tuning and checking knobs only - what a call computes is decided by its
// arguments (dexct_gn_options), never by a variable that changes between two calls.
struct GnEnv {
  int full_loop;            // DEXCT_GN_FULL_LOOP=1: every call executes every iteration
  double default_tol;       // options->stop_tol < 0: DEXCT_GN_EXACT=1 -> 0, DEXCT_GN_STOP_TOL=<t>, else DEXCT_GN_DEFAULT_STOP_TOL
  int blocks_per_cu;        // DEXCT_GN_BLOCKS_PER_CU=<n>: workgroups per CU of the queue kernels (0: what is resident)
  long long coop_below;     // DEXCT_GN_COOP_BELOW=<pixels>
  int sort;                 // DEXCT_GN_SORT=0: natural order of the hand-out on small sinograms
  int tiles_per_fetch;      // DEXCT_GN_TILES_PER_FETCH=<n>: queue positions per reservation (0: by size)
};

static const GnEnv& gn_env() {
  static const GnEnv env = [] {
    GnEnv e{0, DEXCT_GN_DEFAULT_STOP_TOL, 0, kCoopBelowDefault, 1, 0};
    auto flag = [](const char* name, char c) { const char* v = getenv(name); return v && v[0] == c; };
    auto number = [](const char* name, long long lo, long long hi, long long dflt) {
      const char* v = getenv(name);
      if (!v || !*v) return dflt;
      char* end = nullptr;
      const long long x = strtoll(v, &end, 10);
      return (end && *end == 0 && x >= lo && x <= hi) ? x : dflt;
    };
    e.full_loop = flag("DEXCT_GN_FULL_LOOP", '1');
    if (flag("DEXCT_GN_EXACT", '1')) {
      e.default_tol = 0.0;
    } else if (const char* t = getenv("DEXCT_GN_STOP_TOL")) {
      char* end = nullptr;
      const double x = strtod(t, &end);
      if (end && end != t && *end == 0 && x >= 0.0) e.default_tol = x;      // anything else: ignored (the default stays)
    }
    e.blocks_per_cu = (int)number("DEXCT_GN_BLOCKS_PER_CU", 1, 64, 0);
    e.coop_below = number("DEXCT_GN_COOP_BELOW", 0, 1ll << 62, kCoopBelowDefault);
    e.sort = !flag("DEXCT_GN_SORT", '0');
    e.tiles_per_fetch = (int)number("DEXCT_GN_TILES_PER_FETCH", 1, 1024, 0);
    return e;
  }();
  return env;
}

}  // namespace dexct

using namespace dexct;

// The energy sums of the forward model at n states (the table assembly of the short cut: quadrature._model_sums - thresholds and
// bounds, no bit of a result): per state nu[k] = sum_e i0[k][e] att_e, G[k][m] = sum_e i0[k][e] mu[m][e] live_e and, when asked,
// S[k][m][p] = sum_e i0[k][e] mu[m][e] mu[p][e] live_e, with att_e = exp(clip(-(a0 mu0[e] + a1 mu1[e]), +-700)) (matdecomp.py:116)
// and live_e = att_e where the clip is not active, else 0 (the clipped exponent has no slope).  One lane per state, the tables by
// scalar loads, the library exponential.  (Round 6: these ran as torch float64 kernels on the tables' device; the calibration of
// the soak's 1 - 3 energy tables then aborted the process once in a few runs of the GPU suite, from a thread of the runtime,
// with the host in exactly these passes - the product's own arithmetic does not go through another library any more.)
__global__ __launch_bounds__(256) void gn_model_sums_kernel(const double* __restrict__ a, long long n, const double* __restrict__ i0,
                                                             const double* __restrict__ mus, int n_e, double* __restrict__ nu_out,
                                                             double* __restrict__ g_out, double* __restrict__ s_out) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const double a0 = a[2 * p], a1 = a[2 * p + 1];
  double nu[2] = {0.0, 0.0}, G[2][2] = {{0.0, 0.0}, {0.0, 0.0}}, S[2][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  for (int e = 0; e < n_e; ++e) {
    const double m0 = mus[e], m1 = mus[n_e + e];
    const double x = -(a0 * m0 + a1 * m1);
    const double att = exp(fmin(fmax(x, -700.0), 700.0));
    const double live = fabs(x) < 700.0 ? att : 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const double w = i0[k * n_e + e];
      nu[k] = fma(w, att, nu[k]);
      const double wl = w * live;
      G[k][0] = fma(wl, m0, G[k][0]);
      G[k][1] = fma(wl, m1, G[k][1]);
      if (s_out) {
        S[k][0] = fma(wl * m0, m0, S[k][0]);
        S[k][1] = fma(wl * m0, m1, S[k][1]);
        S[k][2] = fma(wl * m1, m1, S[k][2]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    nu_out[2 * p + k] = nu[k];
    g_out[4 * p + 2 * k] = G[k][0];
    g_out[4 * p + 2 * k + 1] = G[k][1];
    if (s_out) {
      double* sp = s_out + 8 * p + 4 * k;          // [k][m][p], symmetric in (m, p)
      sp[0] = S[k][0]; sp[1] = S[k][1]; sp[2] = S[k][1]; sp[3] = S[k][2];
    }
  }
}

extern "C" {

int64_t dexct_gn_workspace_bytes(int32_t n_energies, int32_t n_bins) {
  if (n_energies <= 0 || n_bins <= 0) return 0;
  return (int64_t)gn_ws_tables_bytes(n_energies, n_bins) + (int64_t)kSortBuckets * sizeof(int) +
         (int64_t)kMaxSortTiles * (sizeof(int) + sizeof(unsigned short));
}

int dexct_gn_decompose(const void* g1, const void* g2, int32_t g_is_f64, int64_t n_pix, const double* i0,
                       const double* mus, int32_t n_energies, int32_t n_bins, int32_t bin_div, int32_t n_iters,
                       int32_t precision, int32_t n_polish, const double* mask_max, double mask_frac, double* out_a,
                       const dexct_gn_options* options, void* workspace, void* stream) {
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
  // order of the results: the pixels' own, or [..][row][channel] for pixels given as [..][channel][row]
  GnTiling tl{(n_pix + kTilePix - 1) / kTilePix, 0, 1, 1, 1, 1, 1u, 0u, 1u, 0u};
  if (options && (options->out_rows != 0 || options->out_channels != 0)) {
    const int64_t R = options->out_rows, C = options->out_channels;
    if (R < 1 || C < 1 || n_pix % (R * C) != 0) return DEXCT_EINVAL;
    tl.transposed = 1;
    tl.rows = (int)R;
    tl.channels = (int)C;
    tl.tiles_r = (int)((R + kTileR - 1) / kTileR);
    tl.tiles_c = (int)((C + kTileC - 1) / kTileC);
    tl.n_tiles = (n_pix / (R * C)) * tl.tiles_r * tl.tiles_c;
    gn_magic((unsigned)tl.tiles_r, &tl.mul_r, &tl.sh_r);
    gn_magic((unsigned)tl.tiles_c, &tl.mul_c, &tl.sh_c);
  }
  if (tl.n_tiles > 0x7FFFFFFFll) return DEXCT_ERANGE;
  if (options && (options->kernel < 0 || options->kernel > 2)) return DEXCT_EINVAL;
  // the step-counting launch and the short cut (see gn_refill_kernel<true>, gn_shortcut_kernel): lane kernels, one shared
  // spectrum, float64, step counts in a byte
  const int pass = options ? options->pass : 0;
  if (pass < 0 || pass > 2) return DEXCT_EINVAL;
  if (pass != 0 && (n_bins > 1 || precision != 0 || n_iters > 254 || options->kernel == 2)) return DEXCT_EINVAL;
  if (pass == DEXCT_GN_PASS_COUNT && (!options->iterations || options->start)) return DEXCT_EINVAL;
  if (pass == DEXCT_GN_PASS_SHORTCUT && (!options->start || options->iterations)) return DEXCT_EINVAL;
  const double* start = (pass == DEXCT_GN_PASS_SHORTCUT) ? options->start : nullptr;
  if (start && (reinterpret_cast<uintptr_t>(start) & 15u)) return DEXCT_EINVAL;   // its pairs are read with 16-byte loads
  if (options && (options->flags & ~(DEXCT_GN_FLAG_FULL_LOOP | DEXCT_GN_FLAG_NATURAL_ORDER | DEXCT_GN_FLAG_ONE_STEP))) return DEXCT_EINVAL;
  if (options && (options->flags & DEXCT_GN_FLAG_ONE_STEP) && pass != DEXCT_GN_PASS_SHORTCUT) return DEXCT_EINVAL;
  if (options && options->blocks_per_cu < 0) return DEXCT_EINVAL;
  hipStream_t st = as_stream(stream);
  double* ws = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(gn_tables_kernel, dim3(n_bins), dim3(256), 0, st, i0, mus, n_energies, n_bins, ws);
  DEXCT_LAUNCH_CHECK();
  const GnEnv& env = gn_env();
  const int oflags = options ? options->flags : 0;
  // DEXCT_GN_FLAG_FULL_LOOP (or DEXCT_GN_FULL_LOOP=1 at process start) runs every iteration: the check that the repeated-state
  // exit changes no bit
  const int exact_exit = ((oflags & DEXCT_GN_FLAG_FULL_LOOP) || env.full_loop) ? 0 : 1;
  // The tolerance stop.  options->stop_tol >= 0 is taken as given (0 = the reference's fixed count, bit for bit);
  // negative or no options = the library default: 1e-12, or DEXCT_GN_STOP_TOL, or 0 with DEXCT_GN_EXACT=1 (read once).
  double tol = options ? options->stop_tol : -1.0;
  if (!(tol >= 0.0)) tol = env.default_tol;
  if (!exact_exit) tol = 0.0;                         // the full loop is the full loop
  // a walk from the reference's start value ends by the tolerance rule only when the step before contracted too (gn_converged)
  const int confirm_flag = 4;
  if (pass != 0 && !(tol > 0.0)) return DEXCT_EINVAL;  // both passes end pixels by the tolerance rule
  const dim3 grid((unsigned)nblk), block(kGnBlock);
  if (n_bins > 1) {
    hipLaunchKernelGGL((gn_kernel<false, true>), grid, block, 0, st, g1, g2, g_is_f64, n_pix, (const double*)ws,
                       n_energies, n_iters, 0, n_bins, bin_div, mask_max, mask_frac, exact_exit | confirm_flag, tol, tl, out_a);
  } else if (precision == 0) {
    // Lane refill from a queue of 64-pixel tiles; no more workgroups than can be resident: 28 - 35 KB of LDS and the registers
    // allow 4 per CU (those beyond would start when the queue is already empty).  Round 3 measured the fetch size: 64 / 128 /
    // 256 / 512 / 1024 pixels = 766 / 764 / 769 / 771 / 773 ms on the benchmark sinograms and 64 ahead below 1e8 pixels
    // (tools/probes/gn_small2.py): the tail of a launch is the last fetches' slowest pixels.
    static const int n_cu = [] {             // queried once per process (one GPU per process)
      int dev_id = 0, n = 0;
      if (hipGetDevice(&dev_id) != hipSuccess ||
          hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess || n <= 0)
        n = 256;
      return n;
    }();
    const int per_cu = (options && options->blocks_per_cu > 0) ? options->blocks_per_cu : env.blocks_per_cu;
    const int64_t cap = (int64_t)n_cu * (per_cu > 0 ? per_cu : 4);
    int64_t nb = (tl.n_tiles + kGnBlock / kWave - 1) / (kGnBlock / kWave);
    if (nb > cap) nb = cap;
    unsigned long long* counters = reinterpret_cast<unsigned long long*>(ws) + 9;
    // the cooperative kernel below DEXCT_GN_COOP_BELOW pixels (options->kernel: 1 / 2 force one or the other)
    const int which = options ? options->kernel : 0;
    // small sinograms: thick tiles first (DEXCT_GN_FLAG_NATURAL_ORDER / DEXCT_GN_SORT=0 keep the natural order)
    const int* order = nullptr;
    if (pass == 0 && tl.n_tiles <= kMaxSortTiles && tl.n_tiles > 1 && env.sort && !(oflags & DEXCT_GN_FLAG_NATURAL_ORDER)) {
      char* base = reinterpret_cast<char*>(workspace) + gn_ws_tables_bytes(n_energies, n_bins);
      int* hist = reinterpret_cast<int*>(base);
      int* ord = hist + kSortBuckets;
      unsigned short* keys = reinterpret_cast<unsigned short*>(ord + kMaxSortTiles);
      hipLaunchKernelGGL(gn_tile_key_kernel, dim3((unsigned)((tl.n_tiles + 63) / 64)), dim3(256), 0, st, g1, g_is_f64,
                         (long long)n_pix, tl, mask_max, mask_frac, hist, keys);
      DEXCT_LAUNCH_CHECK();
      hipLaunchKernelGGL(gn_tile_scan_kernel, dim3(1), dim3(1024), 0, st, hist);
      DEXCT_LAUNCH_CHECK();
      hipLaunchKernelGGL(gn_tile_scatter_kernel, dim3((unsigned)((tl.n_tiles + 255) / 256)), dim3(256), 0, st, (int)tl.n_tiles, hist,
                         (const unsigned short*)keys, ord);
      DEXCT_LAUNCH_CHECK();
      order = ord;
    }
    // tiles a wave reserves per atomic on the queue head: 1 for the single launch (measured in round 3: larger fetches gain
    // nothing at ~17 steps per pixel and lengthen the tail).  On the short cut (two steps per pixel) the one word all
    // waves of the chip queue for is the limit (12 ns per atomic): 2 / 4 / 8 tiles per reservation once a wave gets 3 / 8 / 128
    // tiles on average (tools/probes/gn_tpf_small.py, 1 / 2 / 4 / 8 tiles: 1200 x 800: 0.69 / 0.57 / 0.60 / 0.74 ms; 2.6e6 pixels:
    // 1.29 / 0.85 / 0.82 / 0.84; 1.2e7: 4.75 / 2.62 / 2.49 / 2.55; 250 views of the benchmark: 38.7 / 20.4 / 17.7 / 17.6);
    // DEXCT_GN_TILES_PER_FETCH overrides
    const int64_t tiles_per_wave = tl.n_tiles / (nb * (kGnBlock / kWave));
    int tiles_per_fetch = pass != DEXCT_GN_PASS_SHORTCUT ? 1 : (tiles_per_wave >= 128 ? 8 : tiles_per_wave >= 8 ? 4 : tiles_per_wave >= 3 ? 2 : 1);
    if (env.tiles_per_fetch >= 1) tiles_per_fetch = env.tiles_per_fetch;
    const int kflags = exact_exit | (order ? 2 : 0) | confirm_flag | (tiles_per_fetch << 8);
    if (pass == DEXCT_GN_PASS_COUNT)
      hipLaunchKernelGGL((gn_refill_kernel<true>), dim3((unsigned)nb), block, 0, st, g1, g2, g_is_f64, (long long)n_pix,
                         (const double*)ws, n_energies, n_iters, tl, mask_max, mask_frac, kflags, tol, out_a, counters, options->iterations);
    else if (pass == DEXCT_GN_PASS_SHORTCUT && (oflags & DEXCT_GN_FLAG_ONE_STEP))
      hipLaunchKernelGGL(gn_shortcut_kernel<1>, dim3((unsigned)nb), block, 0, st, g1, g2, g_is_f64, (long long)n_pix,
                         (const double*)ws, n_energies, n_iters, tl, mask_max, mask_frac, kflags, tol, out_a, counters, start);
    else if (pass == DEXCT_GN_PASS_SHORTCUT)
      hipLaunchKernelGGL(gn_shortcut_kernel<2>, dim3((unsigned)nb), block, 0, st, g1, g2, g_is_f64, (long long)n_pix,
                         (const double*)ws, n_energies, n_iters, tl, mask_max, mask_frac, kflags, tol, out_a, counters, start);
    else if (which == 2 || (which == 0 && n_pix < env.coop_below)) {
      int64_t ncb = tl.n_tiles;
      const int64_t ccap = (int64_t)n_cu * (per_cu > 0 ? per_cu : 3);
      if (ncb > ccap) ncb = ccap;
      hipLaunchKernelGGL(gn_coop_kernel, dim3((unsigned)ncb), dim3(kCoopWaves * kWave), 0, st, g1, g2, g_is_f64, (long long)n_pix,
                         (const double*)ws, n_energies, n_iters, tl, mask_max, mask_frac, exact_exit | (order ? 2 : 0) | confirm_flag, tol, out_a, counters);
    } else
      hipLaunchKernelGGL((gn_refill_kernel<false>), dim3((unsigned)nb), block, 0, st, g1, g2, g_is_f64, (long long)n_pix,
                         (const double*)ws, n_energies, n_iters, tl, mask_max, mask_frac, kflags, tol, out_a, counters, nullptr);
  } else {
    hipLaunchKernelGGL((gn_kernel<true, false>), grid, block, 0, st, g1, g2, g_is_f64, n_pix, (const double*)ws,
                       n_energies, n_iters, n_polish, 1, 1, mask_max, mask_frac, exact_exit, 0.0, tl, out_a);
  }
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_gn_model_sums(const double* a, int64_t n_states, const double* i0, const double* mus, int32_t n_energies,
                        double* nu_out, double* g_out, double* s_out, void* stream) {
  if (!a || !i0 || !mus || !nu_out || !g_out || n_states <= 0 || n_energies <= 0) return DEXCT_EINVAL;
  const int64_t nblk = (n_states + 255) / 256;
  if (nblk > 0x7FFFFFFFll) return DEXCT_ERANGE;
  hipLaunchKernelGGL(gn_model_sums_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), a, (long long)n_states, i0, mus,
                     n_energies, nu_out, g_out, s_out);
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
