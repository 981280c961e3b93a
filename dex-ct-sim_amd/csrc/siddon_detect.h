// Shared by the traversal kernels (siddon.hip, siddon_packed.hip): launch arguments, ray indexing and the
// polychromatic detection of the per-material path lengths (internal; the public surface is include/dexct.h).
#pragma once
#include <type_traits>

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
  // 1: the planes of acc_out hold path lengths [cm] of EVERY material, material 0 included (the cone-beam group passes, which
  // accumulate material 0 like any other: a 3-D ray has no chord trick); the detection pass then reads no plan
  int acc_lengths = 0;
  // quantum noise drawn by the detection pass of the material groups (struct dexct_noise, ABI 6): the Philox counter needs the
  // GLOBAL view of a ray; 0 = the pass writes the expectation (and the variance, if asked)
  int view_begin = 0;
  int sample = 0;
  uint32_t seed_lo = 0, seed_hi = 0;
  // second output of get_sino (main.py:120-122): sino_log[s][ray] = ln(air[s] / counts[s][ray]), same ray order as
  // counts; null = not wanted.  air[s] = sum_e w[s][e] (the unattenuated signal), given by the caller.
  float* sino_log;
  float air[DEXCT_MAX_SPECTRA];
};

// log_out of the C ABI -> launch arguments (null: no log sinogram).  With a variance output the log belongs to the noisy
// counts, which only exist after dexct_add_noise: the caller then uses dexct_sino_log (`variance` non-null here = "the counts
// this launch writes are not the final ones"; a kernel that samples the noise itself passes null).
// the noise arguments of the group entry points -> launch arguments; DEXCT_EINVAL for a combination that makes no sense
inline int set_noise(ProjArgs& a, int32_t view_begin, const float* weights2, const float* variance, const dexct_noise* noise) {
  const bool sample = noise && noise->sample;
  a.view_begin = view_begin;
  a.sample = 0;
  a.seed_lo = a.seed_hi = 0u;
  if (!weights2) return (variance || sample) ? DEXCT_EINVAL : DEXCT_OK;
  if (!variance && !sample) return DEXCT_EINVAL;
  if (sample) {
    if (a.n_spectra > 2 || a.n_materials > 48) return DEXCT_ERANGE;      // (the fused variance: two spectrum slots, register lengths)
    a.sample = 1;
    a.seed_lo = (uint32_t)noise->seed;
    a.seed_hi = (uint32_t)(noise->seed >> 32);
  }
  return DEXCT_OK;
}

inline int set_log_out(ProjArgs& a, const dexct_log_out* lo, const float* variance) {
  a.sino_log = nullptr;
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) a.air[s] = 1.0f;
  if (!lo || !lo->sino_log) return DEXCT_OK;
  if (variance) return DEXCT_EINVAL;
  a.sino_log = lo->sino_log;
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) a.air[s] = lo->air[s];
  return DEXCT_OK;
}

// launch arguments of rows16_kernel (siddon_packed.hip): the 2-bit packed volume
struct PackedArgs {
  ProjArgs a;
  const uint8_t* vol_z2;   // [ny][nx][nz / 4]
  int lanes_per_pair;      // lanes a (view, channel) pair occupies (16, 32 or 64)
  int n_zchunks;           // ceil(n_rows / 1024)
  int view_tile;
  int det_masks;           // skip detection FMAs of spectrum slots with zero weights (blocks of four energies)
  int staged_store;        // store the results of a wave through LDS as whole lines (DEXCT_P16_STAGED=0: per-round 16-B stores)
  // quantum noise (the NOISY instantiations; ABI 6): the kernel detects the variance with the counts and, `sample` != 0, draws
  // the sample itself (noise_sample.h; the Philox counter needs the GLOBAL view)
  int view_begin;
  int sample;
  uint32_t seed_lo, seed_hi;
};

// The attenuation and weight tables are passed as DIRECT __restrict__ kernel arguments (not inside
// ProjArgs): only then does the compiler know they are read-only and wave-uniform and fetch them with
// s_load through the scalar cache; as struct members they were fetched with 670 vector loads per wave.
struct Tables {
  const float* __restrict__ mu;   // [M][nE] linear attenuation [1/cm]
  const float* __restrict__ w;    // [S][nE]
  const float* __restrict__ w2;   // [S][nE] w * signal per photon (variance weights), or null
};

// detection of the material-major accumulator planes a group pass leaves in a.acc_out (siddon.hip; any material count)
int launch_detect_any(const ProjArgs& a, const Tables& t, hipStream_t st);

__device__ __forceinline__ size_t ray_index(const ProjArgs& a, int v, int r, int c) {
  return a.layout == 0 ? ((size_t)v * a.g.n_rows + r) * a.g.n_channels + c
                       : ((size_t)v * a.g.n_channels + c) * a.g.n_rows + r;
}

// sum_e w[s][e] * 2^(-sum_m mu[m][e] * L2[q][m]) for R rays and SLOTS spectrum slots.  Rays are handled in pairs
// (float2 -> v_pk_fma_f32, the scalar table value broadcast to both halves); each half performs exactly the
// scalar sequence of fmaf's, so the sums do not depend on the pairing.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Which blocks of four energies each of the first two spectrum slots weights at all (bit b = block b; blocks past 63
// count as weighted).  Built once per wave by detect_block_masks, which every lane of the wave must reach (ballot).
struct BlockMasks {
  unsigned long long m[2];
  bool use;                 // false: every block runs every slot (the plain loop, no branches)
};

__device__ __forceinline__ BlockMasks detect_block_masks(const float* __restrict__ w, int n_e, int n_spectra) {
  BlockMasks bm;
  bm.use = n_spectra <= 2;
  const int b = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bool nz = true;
    if (s < n_spectra && 4 * b + 4 <= n_e) {
      const float* q = w + (size_t)s * n_e + 4 * b;
      nz = (__float_as_uint(q[0]) | __float_as_uint(q[1]) | __float_as_uint(q[2]) | __float_as_uint(q[3])) != 0u;
    }
    bm.m[s] = s < n_spectra ? __ballot(nz) : 0ull;
  }
  return bm;
}

// VAR (round 6): every slot also accumulates the VARIANCE of its signal, sum_e w2[s][e] 2^(...), from the same exponentials -
// per energy one more FMA per slot and pair instead of a second pass over exponents and exponentials (the noisy scan: 18
// instead of 28 vector instructions per energy and four rays of a dual-energy scan).  Per ray exactly the fmaf sequence of
// the separate variance loop of detect_store, so the two forms give the same bits; a block of four energies a slot does not
// weight (w = 0, hence w2 = w x gain = 0) is skipped for both sums.
template <int NM, int R, int SLOTS, bool VAR = false>
__device__ __forceinline__ void detect_energies(const f32x2 (&Lp)[(R + 1) / 2][NM], const float* __restrict__ mu,
                                                const float* __restrict__ w, int n_e,
                                                const int (&srow)[DEXCT_MAX_SPECTRA], const BlockMasks& bm,
                                                float (&acc)[DEXCT_MAX_SPECTRA][R], const float* __restrict__ w2 = nullptr,
                                                float (*accv)[DEXCT_MAX_SPECTRA][R] = nullptr) {
  constexpr int P = (R + 1) / 2;                      // pairs; an odd last ray rides alone in a pair's low half
  f32x2 ap[SLOTS][P], av[VAR ? SLOTS : 1][P];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s)
#pragma unroll
    for (int j = 0; j < P; ++j) ap[s][j] = f32x2{0.0f, 0.0f};
#pragma unroll
  for (int s = 0; s < (VAR ? SLOTS : 1); ++s)
#pragma unroll
    for (int j = 0; j < P; ++j) av[s][j] = f32x2{0.0f, 0.0f};
  // LIVE: bit s set = slot s accumulates.  A slot whose weights are all +0 over a block of four energies adds exactly
  // nothing (w * finite = 0, acc + 0 = acc), so its FMAs are skipped for the block without changing a bit: the
  // 80 kVp spectrum of a dual-energy scan weights no energy above 80 keV, a single-spectrum scan has no second slot.
  auto one_energy = [&](int e, auto live_tag) {
    constexpr uint32_t LIVE = decltype(live_tag)::value;
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
      if (!((LIVE >> s) & 1u)) continue;
      const float ws = w[srow[s] + e];
#pragma unroll
      for (int j = 0; j < P; ++j) ap[s][j] = __builtin_elementwise_fma(f32x2{ws, ws}, te[j], ap[s][j]);
      if constexpr (VAR) {
        const float ws2 = w2[srow[s] + e];
#pragma unroll
        for (int j = 0; j < P; ++j) av[s][j] = __builtin_elementwise_fma(f32x2{ws2, ws2}, te[j], av[s][j]);
      }
    }
  };
  constexpr uint32_t kAll = (1u << SLOTS) - 1u;
  auto four = [&](int e, auto live_tag) {
    one_energy(e, live_tag);
    one_energy(e + 1, live_tag);
    one_energy(e + 2, live_tag);
    one_energy(e + 3, live_tag);
  };
  int e = 0;
  bool masked = false;
  if constexpr (SLOTS == 2) masked = bm.use;
  if (masked) {
    // runs of blocks with the same class, each run a tight loop of its own (a branch per block costs more than the
    // skipped FMAs save: measured).  Wave-uniform, from registers.
    const int nblk = n_e >> 2;
    int b = 0;
    while (b < nblk) {
      uint32_t cls = 3u;
      int run = nblk - b;
      if (b < 64) {
        const unsigned long long s0 = bm.m[0] >> b, s1 = bm.m[1] >> b;
        cls = (uint32_t)(s0 & 1ull) | ((uint32_t)(s1 & 1ull) << 1);
        // first block that differs in either slot (bits past the top of the shifted masks read 0: for a weighted
        // slot the run then ends at block 64 and the blocks beyond run as class 3)
        const unsigned long long d = (s0 ^ (0ull - (s0 & 1ull))) | (s1 ^ (0ull - (s1 & 1ull)));
        const int same = d ? __builtin_ctzll(d) : 64;
        run = min(run, min(same, 64 - b));
      }
      const int e_end = 4 * (b + run);
      if (cls == 3u) for (; e < e_end; e += 4) four(e, std::integral_constant<uint32_t, 3u>{});
      else if (cls == 1u) for (; e < e_end; e += 4) four(e, std::integral_constant<uint32_t, 1u>{});
      else if (cls == 2u) for (; e < e_end; e += 4) four(e, std::integral_constant<uint32_t, 2u>{});
      else e = e_end;
      b += run;
    }
  } else {
    for (; e + 4 <= n_e; e += 4) four(e, std::integral_constant<uint32_t, kAll>{});
  }
  for (; e < n_e; ++e) one_energy(e, std::integral_constant<uint32_t, kAll>{});
#pragma unroll
  for (int s = 0; s < SLOTS; ++s)
#pragma unroll
    for (int q = 0; q < R; ++q) acc[s][q] = (q & 1) ? ap[s][q / 2].y : ap[s][q / 2].x;
  if constexpr (VAR) {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
      for (int q = 0; q < R; ++q) (*accv)[s][q] = (q & 1) ? av[s][q / 2].y : av[s][q / 2].x;
  }
}

// One ray per lane (the cone-beam kernels, whose lanes are the rows of a pair): the pairs that go through v_pk_fma_f32
// are pairs of ENERGIES - exponent FMAs with (mu[m][e], mu[m][e+1]) from the scalar cache, two v_exp_f32, one FMA per
// spectrum slot with (w[s][e], w[s][e+1]); even and odd energies are summed separately and added at the end (another
// summation order than the one-energy loop's: 1e-7 relative).  3.5 vector instructions per energy instead of 6 and four
// energies per trip (wide scalar loads); zero-weight blocks of a slot are skipped as in detect_energies.  L2 = path
// lengths x log2(e).  <= 2 spectrum slots; acc[s] for s >= n_spectra comes back 0.
// VAR: the variances sum_e w2[s][e] 2^(...) from the same exponentials (accv), as detect_energies<VAR>.
template <int NM, bool VAR = false>
__device__ __forceinline__ void detect_energy_pairs(const float (&L2)[NM], const float* __restrict__ mu,
                                                    const float* __restrict__ w, int n_e, int n_spectra,
                                                    const BlockMasks& bm, float (&acc)[2], const float* __restrict__ w2 = nullptr,
                                                    float (*accv)[2] = nullptr) {
  f32x2 ap[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
  f32x2 av[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
  const int row1 = (n_spectra > 1 ? 1 : 0) * n_e;
  auto two = [&](int e, auto live_tag) {
    constexpr uint32_t LIVE = decltype(live_tag)::value;
    f32x2 pe = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int m = 0; m < NM; ++m)
      pe = __builtin_elementwise_fma(f32x2{mu[m * n_e + e], mu[m * n_e + e + 1]}, f32x2{L2[m], L2[m]}, pe);
    const f32x2 te = f32x2{__builtin_amdgcn_exp2f(-pe.x), __builtin_amdgcn_exp2f(-pe.y)};
    if (LIVE & 1u) ap[0] = __builtin_elementwise_fma(f32x2{w[e], w[e + 1]}, te, ap[0]);
    if (LIVE & 2u) ap[1] = __builtin_elementwise_fma(f32x2{w[row1 + e], w[row1 + e + 1]}, te, ap[1]);
    if constexpr (VAR) {
      if (LIVE & 1u) av[0] = __builtin_elementwise_fma(f32x2{w2[e], w2[e + 1]}, te, av[0]);
      if (LIVE & 2u) av[1] = __builtin_elementwise_fma(f32x2{w2[row1 + e], w2[row1 + e + 1]}, te, av[1]);
    }
  };
  auto four = [&](int e, auto live_tag) {
    two(e, live_tag);
    two(e + 2, live_tag);
  };
  int e = 0;
  const int nblk = n_e >> 2;
  int b = 0;
  while (b < nblk) {                                 // runs of blocks of the same class (wave-uniform), as detect_energies
    uint32_t cls = 3u;
    int run = nblk - b;
    if (b < 64) {
      const unsigned long long s0 = bm.m[0] >> b, s1 = bm.m[1] >> b;
      cls = (uint32_t)(s0 & 1ull) | ((uint32_t)(s1 & 1ull) << 1);
      const unsigned long long d = (s0 ^ (0ull - (s0 & 1ull))) | (s1 ^ (0ull - (s1 & 1ull)));
      const int same = d ? __builtin_ctzll(d) : 64;
      run = min(run, min(same, 64 - b));
    }
    const int e_end = 4 * (b + run);
    if (cls == 3u) for (; e < e_end; e += 4) four(e, std::integral_constant<uint32_t, 3u>{});
    else if (cls == 1u) for (; e < e_end; e += 4) four(e, std::integral_constant<uint32_t, 1u>{});
    else if (cls == 2u) for (; e < e_end; e += 4) four(e, std::integral_constant<uint32_t, 2u>{});
    else e = e_end;
    b += run;
  }
  float tail0 = 0.0f, tail1 = 0.0f, tailv0 = 0.0f, tailv1 = 0.0f;
  for (; e < n_e; ++e) {                             // n_e not a multiple of 4: the last energies one at a time
    float pe = 0.0f;
#pragma unroll
    for (int m = 0; m < NM; ++m) pe = fmaf(mu[m * n_e + e], L2[m], pe);
    const float t = __builtin_amdgcn_exp2f(-pe);
    tail0 = fmaf(w[e], t, tail0);
    tail1 = fmaf(w[row1 + e], t, tail1);
    if constexpr (VAR) {
      tailv0 = fmaf(w2[e], t, tailv0);
      tailv1 = fmaf(w2[row1 + e], t, tailv1);
    }
  }
  acc[0] = (ap[0].x + ap[0].y) + tail0;
  acc[1] = n_spectra > 1 ? (ap[1].x + ap[1].y) + tail1 : 0.0f;
  if constexpr (VAR) {
    (*accv)[0] = (av[0].x + av[0].y) + tailv0;
    (*accv)[1] = n_spectra > 1 ? (av[1].x + av[1].y) + tailv1 : 0.0f;
  }
}

// The detected signal of a ray that meets nothing but material 0 (air) depends on its chord only: one value per
// (view, channel) pair of a stacked fan.  Kept per lane by kernels that detect several groups of rows per lane.
struct AirCache {
  float l0;                 // air length the values belong to
  float v[2];               // per spectrum slot
  bool have;
  float vv[2];              // the variances (kernels that detect with FVAR)
};

// counts[s] = sum_e w[s][e] * exp(-sum_m mu[m][e] * L[m]) (v_exp_f32 on the log2(e)-scaled exponent)
// for R rays at once (R = 4 in rows4_kernel: one scalar table load serves 4 rays and the FMAs pair up
// into v_pk_fma_f32).  mu and w are wave-uniform (scalar loads); NM materials in registers.
// EXTRAS = false leaves out the optional path-length and variance outputs (the caller writes them or has none).
// FVAR (with res_out and var_out, at most two spectra): the variances sum_e w2[s][e] exp(-...) come out of the same energy
// loop as the counts (detect_energies<.., VAR>) and are handed to the caller like the counts - the kernels that draw the
// noise sample themselves.  Same bits as the separate loop below.
template <int NM, int R, bool EXTRAS = true, bool FVAR = false>
__device__ __forceinline__ void detect_store(const float (&L)[R][NM], const ProjArgs& a, const float* __restrict__ mu,
                                             const float* __restrict__ w, const float* __restrict__ w2,
                                             const size_t (&ray)[R], const bool (&valid)[R],
                                             const BlockMasks& bm = BlockMasks{{~0ull, ~0ull}, false},
                                             AirCache* air_cache = nullptr, float (*res_out)[2][R] = nullptr,
                                             float (*var_out)[2][R] = nullptr) {
  const int n_e = a.n_energies;
  const size_t sstride = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  if (EXTRAS && a.pathlen) {
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
  float acc[DEXCT_MAX_SPECTRA][R], accv[FVAR ? DEXCT_MAX_SPECTRA : 1][FVAR ? R : 1];
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
  // Rays past the object: when every ray of the wave crossed air only (all other lengths exactly 0) and the rays of a
  // lane share their air length - the rows of one (view, channel) pair do - the R sums of a lane are R times the same
  // sequence of operations, so it runs once (the fan's edge channels: 15 % of the benchmark's rays).  Bit-identical.
  bool wave_air = false;
  if constexpr (R == 4) {
    if (a.n_spectra <= 2) {
      bool lane_air = true;
#pragma unroll
      for (int q = 0; q < R; ++q) {
        lane_air = lane_air && L[q][0] == L[0][0];
#pragma unroll
        for (int m = 1; m < NM; ++m) lane_air = lane_air && L[q][m] == 0.0f;
      }
      wave_air = __ballot(!lane_air) == 0ull;
    }
  }
  if (wave_air) {
    float one[DEXCT_MAX_SPECTRA][1], onev[DEXCT_MAX_SPECTRA][1];
    bool cached = false;
    if (air_cache) cached = __ballot(!(air_cache->have && air_cache->l0 == L[0][0])) == 0ull;
    if (cached) {
      one[0][0] = air_cache->v[0];
      one[1][0] = air_cache->v[1];
      if constexpr (FVAR) { onev[0][0] = air_cache->vv[0]; onev[1][0] = air_cache->vv[1]; }
    } else {
      f32x2 Lp1[1][NM];
#pragma unroll
      for (int m = 0; m < NM; ++m) Lp1[0][m] = f32x2{L2[0][m], L2[0][m]};
      if constexpr (FVAR) {
        detect_energies<NM, 1, 2, true>(Lp1, mu, w, n_e, srow, bm, one, w2, &onev);
        if (air_cache) *air_cache = AirCache{L[0][0], {one[0][0], one[1][0]}, true, {onev[0][0], onev[1][0]}};
      } else {
        detect_energies<NM, 1, 2>(Lp1, mu, w, n_e, srow, bm, one);
        if (air_cache) *air_cache = AirCache{L[0][0], {one[0][0], one[1][0]}, true, {0.0f, 0.0f}};
      }
    }
#pragma unroll
    for (int q = 0; q < R; ++q) {
      acc[0][q] = one[0][0];
      acc[1][q] = one[1][0];
      if constexpr (FVAR) { accv[0][q] = onev[0][0]; accv[1][q] = onev[1][0]; }
    }
  } else if constexpr (FVAR)
    detect_energies<NM, R, 2, true>(Lp, mu, w, n_e, srow, bm, acc, w2, &accv);
  else if (a.n_spectra <= 2)
    detect_energies<NM, R, 2>(Lp, mu, w, n_e, srow, bm, acc);
  else
    detect_energies<NM, R, DEXCT_MAX_SPECTRA>(Lp, mu, w, n_e, srow, bm, acc);
  if (EXTRAS && a.variance) {
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
  if (res_out) {           // the caller stores (rows16_kernel: whole lines after its last round)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int q = 0; q < R; ++q) (*res_out)[s][q] = acc[s][q];
    if constexpr (FVAR) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < R; ++q) (*var_out)[s][q] = accv[s][q];
    }
    return;
  }
  // 4 consecutive rays (layout 1, rows4_kernel): one 16-byte store per spectrum
  const bool vec4 = R == 4 && a.layout == 1 && (a.g.n_rows & 3) == 0 && valid[R - 1];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
    if (s < a.n_spectra) {
      if (vec4) {
        *reinterpret_cast<float4*>(a.counts + ray[0] + s * sstride) =
            make_float4(acc[s][0], acc[s][R > 1 ? 1 : 0], acc[s][R > 2 ? 2 : 0], acc[s][R > 3 ? 3 : 0]);
        if (a.sino_log)
          *reinterpret_cast<float4*>(a.sino_log + ray[0] + s * sstride) =
              make_float4(log_ratio(a.air[s], acc[s][0]), log_ratio(a.air[s], acc[s][R > 1 ? 1 : 0]),
                          log_ratio(a.air[s], acc[s][R > 2 ? 2 : 0]), log_ratio(a.air[s], acc[s][R > 3 ? 3 : 0]));
      } else {
#pragma unroll
        for (int q = 0; q < R; ++q)
          if (valid[q]) {
            a.counts[ray[q] + s * sstride] = acc[s][q];
            if (a.sino_log) a.sino_log[ray[q] + s * sstride] = log_ratio(a.air[s], acc[s][q]);
          }
      }
    }
}

// The stores of detect_store for callers that took the results into registers (res_out): counts and, if asked, the log sinogram
// of up to two spectra; R consecutive rays leave as one 16-byte piece where the layout allows.
template <int R>
__device__ __forceinline__ void store_rays(const ProjArgs& a, const size_t (&ray)[R], const bool (&valid)[R], const float (&res)[2][R]) {
  const size_t sstride = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  const bool vec4 = R == 4 && a.layout == 1 && (a.g.n_rows & 3) == 0 && valid[R - 1];
#pragma unroll
  for (int s = 0; s < 2; ++s)
    if (s < a.n_spectra) {
      if (vec4) {
        *reinterpret_cast<float4*>(a.counts + ray[0] + s * sstride) =
            make_float4(res[s][0], res[s][R > 1 ? 1 : 0], res[s][R > 2 ? 2 : 0], res[s][R > 3 ? 3 : 0]);
        if (a.sino_log)
          *reinterpret_cast<float4*>(a.sino_log + ray[0] + s * sstride) =
              make_float4(log_ratio(a.air[s], res[s][0]), log_ratio(a.air[s], res[s][R > 1 ? 1 : 0]),
                          log_ratio(a.air[s], res[s][R > 2 ? 2 : 0]), log_ratio(a.air[s], res[s][R > 3 ? 3 : 0]));
      } else {
#pragma unroll
        for (int q = 0; q < R; ++q)
          if (valid[q]) {
            a.counts[ray[q] + s * sstride] = res[s][q];
            if (a.sino_log) a.sino_log[ray[q] + s * sstride] = log_ratio(a.air[s], res[s][q]);
          }
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
