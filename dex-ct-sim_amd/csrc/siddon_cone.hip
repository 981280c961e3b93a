// Cone-beam (3-D) Siddon projection for gfx950.
//
// Same fan in the (x, y) plane as the 2-D projector - the per-(view, channel) plan is reused - but the source
// sits at height src_z and detector row r at height row_z[r], so a ray climbs linearly in z.  z gets the same
// treatment as the minor in-plane coordinate: w(i) = (W0 + i*SW) / 2^40 at the entry face of dominant-axis
// slab i (40-bit fixed point, exact voxel slices), and at most one z-plane is crossed per slab (|dw/du| <= 1 is
// required and checked by the caller).  A slab therefore splits into up to three pieces, cut at the v-crossing
// tv and the z-crossing tw: (a: t1), (middle: t2 - t1), (b: 1 - t2) with t1 = min(tv, tw), t2 = max(tv, tw).
// As in the 2-D kernels the sum over pieces is rewritten so that the common case costs integer work only:
//   sum_q l_q [id_q == m] = [idb == m] + t2 ([idm == m] - [idb == m]) + t1 ([ida == m] - [idm == m]),
// an integer count of the b voxel plus float32 corrections that vanish unless the slab straddles a material
// boundary (a piece outside the grid has no material).  L_m = ((float)count_m + corr_m) * len3d.  Every material,
// material 0 included, is accumulated (the chord trick of the 2-D kernels would need the 3-D clip of every
// ray).  oracle/dexct_oracle.c: orc_cone_pathlen mirrors the arithmetic.
//
// Mapping: one thread per ray, lanes over adjacent channels of one (view, row) - neighbouring rays visit
// neighbouring voxels of the same slices.  Accumulators in registers (<= 4 materials) or per-lane LDS columns.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "noise_sample.h"
#include "siddon_detect.h"

namespace dexct {

constexpr int kConeBlock = 128;

struct ConeArgs {
  dexct_fan_geom g;
  const dexct_ray_plan* plan;
  const double* view_cs;
  const double* chan_cs;
  const double* row_z;     // [n_rows] detector heights [cm], z = 0 at the centre of the grid
  double src_z;
  const uint8_t* vol_yx;   // [nz][ny][nx]
  const uint8_t* vol_xy;   // [nz][nx][ny]
  int view_begin, n_local_views;
  int n_materials, n_energies, n_spectra;
  float* counts;           // [S][view][row][channel]
  float* pathlen;          // optional [ray][M]
  float* sino_log;         // optional [S][view][row][channel]: ln(air[s] / counts)
  float air[DEXCT_MAX_SPECTRA];
  // quantum noise (ABI 6, struct dexct_noise; the kernels' w2 argument non-null): the variance of the signal is summed in the
  // detection's own energy loop; `variance` (optional) receives it, `sample` draws the noisy count in registers
  float* variance;         // optional [S][view][row][channel]
  int sample;
  uint32_t seed_lo, seed_hi;
  // material-group passes of the row kernels (round 6; more than 3 materials): the volume layout holds codes 0..2 of ONE group of
  // three materials (every other voxel reads as "outside", which belongs to nobody); the pass writes the path lengths [cm] of
  // its materials to acc_out[(mat_base + code) * n_rays + ray] and leaves the detection to one pass over all planes
  float* acc_out = nullptr;
  int mat_base = 0;
};

// The tail every cone kernel ends with: per-material lengths (x log2 e) of one ray -> counts of every spectrum (the weighting of
// the 2-D kernels), stored with the optional log; NOISY: with the variances from the same exponentials, the optional variance
// output and the sample (noise_sample.h).  Round 5 ran the whole kernel a second time with w2 as weights to get the variance.
template <int NM, bool NOISY>
__device__ __forceinline__ void cone_detect_store(const ConeArgs& a, const float (&L2)[NM], const float* __restrict__ mu,
                                                  const float* __restrict__ w, const float* __restrict__ w2,
                                                  const BlockMasks& bm, size_t ray, int v, int r, int c) {
  const size_t sstride = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  const int n_e = a.n_energies;
  float accs[DEXCT_MAX_SPECTRA], vars[DEXCT_MAX_SPECTRA];
#pragma unroll
  for (int sI = 0; sI < DEXCT_MAX_SPECTRA; ++sI) accs[sI] = vars[sI] = 0.0f;
  if (a.n_spectra <= 2) {                      // round 3: energies in pairs through v_pk_fma_f32, zero-weight blocks skipped
    float two[2], twov[2] = {0.0f, 0.0f};
    if constexpr (NOISY) detect_energy_pairs<NM, true>(L2, mu, w, n_e, a.n_spectra, bm, two, w2, &twov);
    else detect_energy_pairs<NM>(L2, mu, w, n_e, a.n_spectra, bm, two);
    accs[0] = two[0];
    accs[1] = two[1];
    vars[0] = twov[0];
    vars[1] = twov[1];
  } else {
    int srow[DEXCT_MAX_SPECTRA];
#pragma unroll
    for (int sI = 0; sI < DEXCT_MAX_SPECTRA; ++sI) srow[sI] = (sI < a.n_spectra ? sI : 0) * n_e;
    for (int e = 0; e < n_e; ++e) {
      float pe = 0.0f;
#pragma unroll
      for (int m = 0; m < NM; ++m) pe = fmaf(mu[m * n_e + e], L2[m], pe);
      const float t = __builtin_amdgcn_exp2f(-pe);
#pragma unroll
      for (int sI = 0; sI < DEXCT_MAX_SPECTRA; ++sI) {
        accs[sI] = fmaf(w[srow[sI] + e], t, accs[sI]);
        if constexpr (NOISY) vars[sI] = fmaf(w2[srow[sI] + e], t, vars[sI]);
      }
    }
  }
  if constexpr (NOISY) {
    if (a.variance) {
#pragma unroll
      for (int sI = 0; sI < DEXCT_MAX_SPECTRA; ++sI)
        if (sI < a.n_spectra) a.variance[ray + sI * sstride] = vars[sI];
    }
    if (a.sample) {
      float z[DEXCT_MAX_SPECTRA];
      pixel_normals<DEXCT_MAX_SPECTRA>((uint32_t)(a.view_begin + v), (uint32_t)r, (uint32_t)c, a.seed_lo, a.seed_hi, z);
#pragma unroll
      for (int sI = 0; sI < DEXCT_MAX_SPECTRA; ++sI) accs[sI] = noisy_count(accs[sI], vars[sI], z[sI]);
    }
  }
#pragma unroll
  for (int sI = 0; sI < DEXCT_MAX_SPECTRA; ++sI)
    if (sI < a.n_spectra) {
      a.counts[ray + sI * sstride] = accs[sI];
      if (a.sino_log) a.sino_log[ray + sI * sstride] = log_ratio(a.air[sI], accs[sI]);
    }
}

template <int NM, int CB = kConeBlock>   // CB: lanes per workgroup = width of the per-lane LDS columns (NM == 0)
__global__ __launch_bounds__(CB) void cone_kernel(ConeArgs a, const float* __restrict__ mu,
                                                          const float* __restrict__ w, const float* __restrict__ w2) {
  extern __shared__ float lds_acc[];     // NM == 0: counts then corrections, [n_materials][CB] each
  const int tid = threadIdx.x;
  const int c = blockIdx.x * CB + tid;
  const int r = blockIdx.y, v = blockIdx.z;
  const BlockMasks bm = detect_block_masks(w, a.n_energies, a.n_spectra);      // a ballot: before lanes leave
  if (c >= a.g.n_channels) return;
  const dexct_ray_plan p = a.plan[(size_t)v * a.g.n_channels + c];
  const int axis = p.flags & 1u;
  const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
  const int nv = axis == 0 ? a.g.ny : a.g.nx;
  // ---- z part of the plan (float64, mirrors cone_row_plan of the oracle)
  const int view = a.view_begin + v;
  const double cb = a.view_cs[2 * view], sb = a.view_cs[2 * view + 1];
  const double cg = a.chan_cs[2 * c], sg = a.chan_cs[2 * c + 1];
  const double sx = a.g.sid * cb, sy = a.g.sid * sb;
  const double ex = -(cb * cg - sb * sg), ey = -(sb * cg + cb * sg);
  const double su = axis == 0 ? sx / a.g.dx + 0.5 * a.g.nx : sy / a.g.dy + 0.5 * a.g.ny;
  const double eu = axis == 0 ? ex / a.g.dx : ey / a.g.dy;
  const double det_z = a.row_z[r];
  const double kOne = 1099511627776.0;
  const double ws = a.src_z / a.g.dz + 0.5 * a.g.nz;
  const double sw = ((det_z - a.src_z) / a.g.dz) / (a.g.sdd * eu);
  const double w0 = ws - su * sw;
  const long long SW = llrint(sw * kOne);
  const long long W0 = llrint(w0 * kOne);
  double inv = 16777216.0;
  if (SW != 0) inv = fmin(kOne / fabs((double)SW), 16777216.0);
  const float kfw = (float)(inv * (1.0 / 4294967296.0));
  const double tz = (det_z - a.src_z) / a.g.sdd;
  const float len3d = (float)((1.0 / fabs(eu)) * sqrt(1.0 + tz * tz));
  const uint32_t wpos = SW > 0 ? 0xFFFFFFFFu : 0u;

  const uint8_t* __restrict__ base = axis == 0 ? a.vol_xy : a.vol_yx;
  const uint32_t slice = (uint32_t)a.g.nx * (uint32_t)a.g.ny;
  // NM > 0: counts and corrections in registers; NM == 0: per-lane LDS columns [n_materials][CB] of each
  int32_t cnt[NM > 0 ? NM : 1];
  float corr[NM > 0 ? NM : 1];
  int32_t* lds_cnt = reinterpret_cast<int32_t*>(lds_acc);
  float* lds_corr = lds_acc + (size_t)a.n_materials * CB;
  if (NM > 0) {
#pragma unroll
    for (int m = 0; m < (NM > 0 ? NM : 1); ++m) { cnt[m] = 0; corr[m] = 0.0f; }
  } else {
    for (int m = 0; m < a.n_materials; ++m) { lds_cnt[m * CB + tid] = 0; lds_corr[m * CB + tid] = 0.0f; }
  }
  long long V = p.V0 + (long long)p.i_first * p.SV;
  long long W = W0 + (long long)p.i_first * SW;
  uint32_t off = (uint32_t)p.i_first * (uint32_t)nv;
  for (int s = 0; s < p.n_slabs; ++s) {
    const int32_t ja = (int32_t)(V >> DEXCT_FIX_FRAC), jb = (int32_t)((V + p.SV) >> DEXCT_FIX_FRAC);
    const int32_t ka = (int32_t)(W >> DEXCT_FIX_FRAC), kb = (int32_t)((W + SW) >> DEXCT_FIX_FRAC);
    const float tv = fminf((float)((uint32_t)((unsigned long long)V >> 8) ^ smask) * p.kf, 1.0f);
    const float tw = fminf((float)((uint32_t)((unsigned long long)W >> 8) ^ wpos) * kfw, 1.0f);
    const float t1 = fminf(tv, tw), t2 = fmaxf(tv, tw);
    const bool v_first = tv <= tw;
    const int32_t jm = v_first ? jb : ja, km = v_first ? ka : kb;
    auto voxel = [&](int32_t j, int32_t k) -> uint32_t {      // material id, 256 outside the grid (every uint8 value is an id)
      const bool in = (uint32_t)j < (uint32_t)nv && (uint32_t)k < (uint32_t)a.g.nz;
      return in ? (uint32_t)base[(uint32_t)k * slice + off + (uint32_t)j] : 256u;
    };
    const uint32_t ida = voxel(ja, ka), idm = voxel(jm, km), idb = voxel(jb, kb);
    if (NM > 0) {
#pragma unroll
      for (int m = 0; m < (NM > 0 ? NM : 1); ++m) cnt[m] += (idb == (uint32_t)m) ? 1 : 0;
      if (ida != idm || idm != idb) {
#pragma unroll
        for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
          corr[m] += (idm == (uint32_t)m) ? t2 : 0.0f;
          corr[m] -= (idb == (uint32_t)m) ? t2 : 0.0f;
          corr[m] += (ida == (uint32_t)m) ? t1 : 0.0f;
          corr[m] -= (idm == (uint32_t)m) ? t1 : 0.0f;
        }
      }
    } else {
      const uint32_t nm = (uint32_t)a.n_materials;
      if (idb < nm) lds_cnt[idb * CB + tid] += 1;       // a lane owns its column: plain read-modify-write
      if (ida != idm || idm != idb) {
        if (idm < nm) lds_corr[idm * CB + tid] += t2;
        if (idb < nm) lds_corr[idb * CB + tid] -= t2;
        if (ida < nm) lds_corr[ida * CB + tid] += t1;
        if (idm < nm) lds_corr[idm * CB + tid] -= t1;
      }
    }
    V += p.SV;
    W += SW;
    off += (uint32_t)nv;
  }
  // ---- detection (same weighting as the 2-D kernels)
  const size_t ray = ((size_t)v * a.g.n_rows + r) * a.g.n_channels + c;
  const size_t sstride = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  const int n_e = a.n_energies, n_mat = a.n_materials;
  if (NM > 0) {
    float L2[NM > 0 ? NM : 1];
#pragma unroll
    for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
      const float l = ((float)cnt[m] + corr[m]) * len3d;
      if (a.pathlen) a.pathlen[ray * n_mat + m] = l;
      L2[m] = l * 1.44269504088896340736f;
    }
    // (<= 2 spectra: the detection of the row kernels, energies in pairs - identical counts)
    if (w2) cone_detect_store<(NM > 0 ? NM : 1), true>(a, L2, mu, w, w2, bm, ray, v, r, c);
    else cone_detect_store<(NM > 0 ? NM : 1), false>(a, L2, mu, w, w2, bm, ray, v, r, c);
    return;
  }
  float accs[DEXCT_MAX_SPECTRA], vars[DEXCT_MAX_SPECTRA];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) accs[s] = vars[s] = 0.0f;
  int srow[DEXCT_MAX_SPECTRA];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) srow[s] = (s < a.n_spectra ? s : 0) * n_e;
  for (int m = 0; m < n_mat; ++m) {
    const float l = ((float)lds_cnt[m * CB + tid] + lds_corr[m * CB + tid]) * len3d;
    if (a.pathlen) a.pathlen[ray * n_mat + m] = l;
    lds_corr[m * CB + tid] = l * 1.44269504088896340736f;
  }
  for (int e = 0; e < n_e; ++e) {
    float pe = 0.0f;
    for (int m = 0; m < n_mat; ++m) pe = fmaf(mu[m * n_e + e], lds_corr[m * CB + tid], pe);
    const float t = __builtin_amdgcn_exp2f(-pe);
#pragma unroll
    for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) accs[s] = fmaf(w[srow[s] + e], t, accs[s]);
    if (w2) {
#pragma unroll
      for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) vars[s] = fmaf(w2[srow[s] + e], t, vars[s]);
    }
  }
  if (w2) {
    if (a.variance) {
#pragma unroll
      for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
        if (s < a.n_spectra) a.variance[ray + s * sstride] = vars[s];
    }
    if (a.sample) {
      float z[DEXCT_MAX_SPECTRA];
      pixel_normals<DEXCT_MAX_SPECTRA>((uint32_t)(a.view_begin + v), (uint32_t)r, (uint32_t)c, a.seed_lo, a.seed_hi, z);
#pragma unroll
      for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) accs[s] = noisy_count(accs[s], vars[s], z[s]);
    }
  }
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
    if (s < a.n_spectra) {
      a.counts[ray + s * sstride] = accs[s];
      if (a.sino_log) a.sino_log[ray + s * sstride] = log_ratio(a.air[s], accs[s]);
    }
}

template <int NM>
static int launch_cone(const ConeArgs& a, const float* mu, const float* w, const float* w2, hipStream_t st) {
  if (NM == 0 && a.n_materials > kManyMaterials) {       // 49..256 materials: LDS columns of 64 lanes (<= 128 KB)
    constexpr int B2 = 64;
    const size_t lds2 = (size_t)2 * a.n_materials * B2 * sizeof(float);
    DEXCT_ALLOW_LDS((cone_kernel<NM, B2>), lds2);
    dim3 grid2((a.g.n_channels + B2 - 1) / B2, a.g.n_rows, a.n_local_views);
    hipLaunchKernelGGL((cone_kernel<NM, B2>), grid2, dim3(B2), lds2, st, a, mu, w, w2);
    DEXCT_LAUNCH_CHECK();
    return DEXCT_OK;
  }
  dim3 grid((a.g.n_channels + kConeBlock - 1) / kConeBlock, a.g.n_rows, a.n_local_views);
  const size_t lds = NM > 0 ? 0 : (size_t)2 * a.n_materials * kConeBlock * sizeof(float);
  hipLaunchKernelGGL((cone_kernel<NM, kConeBlock>), grid, dim3(kConeBlock), lds, st, a, mu, w, w2);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

// ---------------------------------------------------------------------------------------------
// cone_rows_kernel: the cone-beam traversal with the rows of one (view, channel) pair as lanes.
//
// All detector rows of a (view, channel) pair share the in-plane trajectory (same V0, SV, slab range, v-crossings);
// only z differs per row.  So a workgroup = one pair (chunk of 256 rows), the in-plane slab records are computed
// once per workgroup into LDS (as in rows_kernel), and a lane carries only its z DDA: W += SW (exact 64-bit add),
// k = W >> 40.  Lanes (neighbouring rows) read neighbouring bytes of one voxel column of a z-fastest volume.
//
// The volume is the GUARDED z-fastest layout of dexct_cone_layout: column (x, y) holds cone_zs(nz) bytes - 16 guard bytes
// of 24, then 8 id(z=0), ..., 8 id(z=nz-1), then at least 16 guard bytes up to a multiple of 16 (so that a column is a
// whole number of 16-byte pieces for cone_cols_kernel's staging) - and one extra column of all 24 stands for every (x, y)
// outside the grid, so a voxel outside the grid reads as id 3 = "no material" without a bounds test (ids < 3: at most 3
// materials).
//
// Per slab and lane the oracle's sum (orc_cone_pathlen)
//     [idb] + t2 ([idm] - [idb]) + t1 ([ida] - [idm])
// needs the b voxel (jb, kb) always; the a voxel (ja, ka) is a different voxel only where the slab has a v-crossing
// (uniform per slab) or the lane a z-crossing, and the middle voxel is a third one only where it has BOTH; the
// corrections vanish unless the ids differ.  Fast path: count the b voxel (packed byte counters, one v_lshl_add),
// load the a voxel / the two possible corner voxels only for the lanes that have the crossing; only lanes that see
// differing ids enter the exact path, which evaluates the oracle's formula operation for operation - per-material
// path lengths are bit-identical to cone_kernel's.
constexpr int kConeGuard = 16;                                  // guard bytes in front of a column
__host__ __device__ inline uint32_t cone_zs(int nz) { return (((uint32_t)nz + 15u) & ~15u) + 32u; }   // bytes per column

struct ConeRec {
  uint32_t colb, cola;   // byte offsets of the b / a voxel columns in the guarded layout (outside column if out of the grid)
  float tv;
  uint32_t flags;        // bit 0: the slab has a v-crossing (ja != jb)
};

constexpr int kConeRows = 256;

// LDSC: the float32 corrections are accumulated in per-lane LDS cells, one row per id (4 ds_add_f32 per boundary slab
// instead of 12 selects + 12 adds + 9 compares in registers: the exact path was 55 of the kernel's 21.6 vector
// instructions per wave and slab).  Same additions in the same order per material - corrections of different
// materials commute - so the path lengths stay bit-identical.
template <int NM, int kB = 4, bool LDSC = true>      // kB: slabs per batch
__global__ __launch_bounds__(kConeRows) void cone_rows_kernel(ConeArgs a, const uint8_t* __restrict__ vol_zc,
                                                               const float* __restrict__ mu, const float* __restrict__ w,
                                                               const float* __restrict__ w2, int n_chunks, int view_tile) {
  __shared__ ConeRec rec[kConeRows];
  __shared__ float lds_corr[LDSC ? 4 : 1][kConeRows];     // [id][lane]; id 3 = outside the grid: a cell nobody reads
  const int tid = threadIdx.x;
  if (LDSC) {
#pragma unroll
    for (int m = 0; m < 4; ++m) lds_corr[m][tid] = 0.0f;  // each lane only ever touches its own cells: no barrier needed
  }
  // block -> (view, channel, row chunk): contiguous logical ids per XCD, views fastest inside a tile of view_tile
  const uint32_t nblk = gridDim.x, bid = blockIdx.x, per = nblk >> 3;
  const uint32_t logical = (bid < (per << 3)) ? (bid & 7u) * per + (bid >> 3) : bid;
  const uint32_t group = (uint32_t)view_tile * a.g.n_channels * n_chunks;
  const uint32_t gq = logical / group, rem = logical - gq * group;
  const uint32_t views_here = min((uint32_t)view_tile, (uint32_t)a.n_local_views - gq * view_tile);
  const int chunk = rem % n_chunks;
  const uint32_t qq = rem / n_chunks;
  const int v = gq * view_tile + qq % views_here, c = qq / views_here;
  const int r = chunk * kConeRows + tid;
  const bool live = r < a.g.n_rows;
  const dexct_ray_plan p = a.plan[(size_t)v * a.g.n_channels + c];      // uniform
  const int axis = p.flags & 1u;
  const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
  const int nv = axis == 0 ? a.g.ny : a.g.nx;
  const uint32_t zs = cone_zs(a.g.nz);                                   // bytes per column
  const uint32_t su = (axis == 0 ? 1u : (uint32_t)a.g.nx) * zs;
  const uint32_t sv = (axis == 0 ? (uint32_t)a.g.nx : 1u) * zs;
  const uint32_t col_out = (uint32_t)a.g.nx * (uint32_t)a.g.ny * zs;    // the all-3 column
  // ---- z part of the plan, float64, operation for operation as cone_kernel / the oracle's cone_row_plan
  const int view = a.view_begin + v;
  const double cb = a.view_cs[2 * view], sb = a.view_cs[2 * view + 1];
  const double cg = a.chan_cs[2 * c], sg = a.chan_cs[2 * c + 1];
  const double sx = a.g.sid * cb, sy = a.g.sid * sb;
  const double ex = -(cb * cg - sb * sg), ey = -(sb * cg + cb * sg);
  const double su_d = axis == 0 ? sx / a.g.dx + 0.5 * a.g.nx : sy / a.g.dy + 0.5 * a.g.ny;
  const double eu = axis == 0 ? ex / a.g.dx : ey / a.g.dy;
  const double det_z = a.row_z[live ? r : 0];
  const double kOne = 1099511627776.0;
  const double ws = a.src_z / a.g.dz + 0.5 * a.g.nz;
  const double sw = ((det_z - a.src_z) / a.g.dz) / (a.g.sdd * eu);
  const double w0 = ws - su_d * sw;
  const long long SW = llrint(sw * kOne);
  const long long W0 = llrint(w0 * kOne);
  double inv = 16777216.0;
  if (SW != 0) inv = fmin(kOne / fabs((double)SW), 16777216.0);
  const float kfw = (float)(inv * (1.0 / 4294967296.0));
  const double tz = (det_z - a.src_z) / a.g.sdd;
  const float len3d = (float)((1.0 / fabs(eu)) * sqrt(1.0 + tz * tz));
  const uint32_t wpos = SW > 0 ? 0xFFFFFFFFu : 0u;

  uint32_t acc = 0;                       // four byte counters: codes 0..3 of the b voxels since the last flush
  uint32_t cnt[3] = {0, 0, 0};
  float corr[3] = {0.0f, 0.0f, 0.0f};
  // W carries a bias of the guard: (W >> 40) is then k + 16, the byte index inside a guarded column
  long long W = W0 + (long long)p.i_first * SW + ((long long)kConeGuard << DEXCT_FIX_FRAC);
  const int k_hi = (int)zs - 1;
  auto slice = [&](long long Wx) {                              // clamp((int)(Wx >> 40), 0, zs - 1): v_ashr + v_med3_i32
    int kq;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(kq) : "v"((int)(Wx >> DEXCT_FIX_FRAC)), "s"(k_hi));
    return kq;
  };
  // Voxel bytes through buffer_load_ubyte with the column's byte offset as the SCALAR offset and the slice as the vector
  // offset.  Round 3: issued from inline assembly - the compiler widened the i8 of __builtin_amdgcn_raw_buffer_load_b8 (and
  // of a plain byte load) with a v_and per load (2 of the 16 vector instructions per slab), although the instruction
  // zero-extends.  The compiler does not count these loads: the batch ends with an explicit s_waitcnt vmcnt(0) that takes
  // every loaded register as an in/out operand, so no use can be scheduled above it; "s_nop 4" = the wait states between
  // the v_readfirstlane that produced a scalar offset and the load reading it.
  typedef int rsrc4 __attribute__((ext_vector_type(4)));
  const uint64_t vbase = (uint64_t)vol_zc;
  const rsrc4 rs = {(int)(uint32_t)vbase, (int)(uint32_t)(vbase >> 32), (int)(col_out + zs), 0x00020000};
  for (int s0 = 0; s0 < p.n_slabs; s0 += kConeRows) {
    const int n_here = min(kConeRows, p.n_slabs - s0);
    const int n_pad = (n_here + kB - 1) / kB * kB;      // the last batch is filled with null records (all "outside")
    if (tid < n_pad) {
      ConeRec q{col_out, col_out, 0.0f, 0u};
      if (tid < n_here) {
        const int i = p.i_first + s0 + tid;
        const SlabPieces sp = dda_slab(p.V0 + (long long)i * p.SV, p.SV, smask, p.kf);
        const bool ina = (uint32_t)sp.ja < (uint32_t)nv, inb = (uint32_t)sp.jb < (uint32_t)nv;
        q.cola = ina ? (uint32_t)i * su + (uint32_t)sp.ja * sv : col_out;
        q.colb = inb ? (uint32_t)i * su + (uint32_t)sp.jb * sv : col_out;
        q.tv = sp.t;
        q.flags = sp.ja != sp.jb ? 1u : 0u;
      }
      rec[tid] = q;
    }
    __syncthreads();
    // two levels so that the flush of the byte counters (they hold at most 128 slabs) needs no test inside the batch loop
    for (int s1 = 0; s1 < n_pad; s1 += 128) {
    const int s_end = min(n_pad, s1 + 128);
    for (int s = s1; s < s_end; s += kB) {
      // ---- all voxel bytes of kB slabs first (no data-dependent branch in between): b = (jb, kb), a = (ja, ka) and,
      // in a slab with a v-crossing, the two possible middle voxels (jb, ka) and (ja, kb).  Without a crossing these
      // are the same byte again (an L1 hit); what differs is found by comparing the ids afterwards.
      ConeRec q[kB];
      int kc[kB + 1];
      uint32_t x[kB], xa[kB], c1[kB], c2[kB];
      uint32_t vx[kB];                                           // scalar: non-zero = the slab has a v-crossing
      const long long Wb = W;
      kc[0] = slice(W);
#pragma unroll
      for (int j = 0; j < kB; ++j) q[j] = rec[s + j];            // uniform; all LDS reads first
#pragma unroll
      for (int j = 0; j < kB; ++j) {
        W += SW;
        kc[j + 1] = slice(W);
        const int sb = __builtin_amdgcn_readfirstlane((int)q[j].colb), sa = __builtin_amdgcn_readfirstlane((int)q[j].cola);
        asm volatile("s_nop 4\n\tbuffer_load_ubyte %0, %2, %4, %5 offen\n\tbuffer_load_ubyte %1, %3, %4, %6 offen"
                     : "=&v"(x[j]), "=&v"(xa[j])
                     : "v"(kc[j + 1]), "v"(kc[j]), "s"(rs), "s"(sb), "s"(sa));
        // a v-crossing slab has two different columns (two outside pieces share the all-24 column: nothing to tell apart)
        vx[j] = (uint32_t)(sb ^ sa);
        if (vx[j] != 0u) {                                           // uniform; c1 / c2 are only ever read under vx[j]
          asm volatile("buffer_load_ubyte %0, %2, %4, %5 offen\n\tbuffer_load_ubyte %1, %3, %4, %6 offen"
                       : "=&v"(c1[j]), "=&v"(c2[j])
                       : "v"(kc[j]), "v"(kc[j + 1]), "s"(rs), "s"(sb), "s"(sa));
        }
      }
#pragma unroll
      for (int j = 0; j < kB; ++j)                               // the first one waits, the others find the counter at 0
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[j]), "+v"(xa[j]));
      // ---- fast path: count the b voxel (the layout stores 8 * id: one v_lshl_add), compare the ids into LANE MASKS
      // (v_cmp writing an SGPR pair; every OR below is scalar)
      unsigned long long dm[kB], any = 0ull;
#pragma unroll
      for (int j = 0; j < kB; ++j) {
        acc = (1u << x[j]) + acc;
        // with a single crossing (c1, c2) is (x, xa) or (xa, x): nothing new; with both, the middle voxel is one of them
        dm[j] = __builtin_amdgcn_ballot_w64(xa[j] != x[j]);
        // the flag is made opaque before each further test: every test is then its own s_cmp + branch (merged, the
        // compiler keeps two 64-bit masks per slab alive - 6 scalar instructions per slab instead of 3)
        asm volatile("" : "+s"(vx[j]));
        if (vx[j] != 0u) {                                       // uniform; c1 / c2 exist only here (defined under the same test)
          asm volatile("" : "+v"(c1[j]), "+v"(c2[j]));           // ordered behind the waits above
          dm[j] |= __builtin_amdgcn_ballot_w64(c1[j] != x[j]) | __builtin_amdgcn_ballot_w64(c2[j] != x[j]);
        }
        any |= dm[j];
      }
      if (any != 0ull) {
#pragma unroll
        for (int j = 0; j < kB; ++j) {
          if (dm[j] != 0ull && __builtin_amdgcn_inverse_ballot_w64(dm[j])) {
            // the oracle's slab, operation for operation (orc_cone_pathlen); W of the slab's entry face without the bias
            const long long Wj = Wb + (long long)j * SW - ((long long)kConeGuard << DEXCT_FIX_FRAC);
            const float tv = q[j].tv;
            const float tw = fminf((float)((uint32_t)((unsigned long long)Wj >> 8) ^ wpos) * kfw, 1.0f);
            const float t1 = fminf(tv, tw), t2 = fmaxf(tv, tw);
            const bool v_first = tv <= tw;
            // middle voxel (jm, km) = v_first ? (jb, ka) : (ja, kb)
            uint32_t vxc = vx[j];
            asm volatile("" : "+s"(vxc));
            const uint32_t idm = vxc != 0u ? (v_first ? c1[j] : c2[j]) : (v_first ? xa[j] : x[j]);
            const uint32_t ida = xa[j], idb = x[j];
            if (ida != idm || idm != idb) {
              if constexpr (LDSC) {
                // the oracle's four terms, each added to the cell of the id it belongs to, in the oracle's order
                float* cell = &lds_corr[0][tid];                 // ids are stored as 8 * id: row id = cell + (8 id) * 32
                static_assert(kConeRows == 8 * 32, "cell offset of a stored id");
                __hip_atomic_fetch_add(cell + idm * 32u, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(cell + idb * 32u, -t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(cell + ida * 32u, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(cell + idm * 32u, -t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              } else {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                  corr[m] += (idm == (uint32_t)(8 * m)) ? t2 : 0.0f;
                  corr[m] -= (idb == (uint32_t)(8 * m)) ? t2 : 0.0f;
                  corr[m] += (ida == (uint32_t)(8 * m)) ? t1 : 0.0f;
                  corr[m] -= (idm == (uint32_t)(8 * m)) ? t1 : 0.0f;
                }
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) cnt[m] += (acc >> (8 * m)) & 0xFFu;
    acc = 0;
    }
    __syncthreads();
  }
  if (LDSC) {
#pragma unroll
    for (int m = 0; m < NM; ++m) corr[m] = lds_corr[m][tid];
  }
  const BlockMasks bm = detect_block_masks(w, a.n_energies, a.n_spectra);      // a ballot: before dead rows leave
  if (!live) return;
  // ---- detection (same weighting as the other kernels)
  const size_t ray = ((size_t)v * a.g.n_rows + r) * a.g.n_channels + c;
  float L2[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const float l = ((float)(int32_t)cnt[m] + corr[m]) * len3d;
    if (a.acc_out) {         // group pass: lengths out, detection later (uniform branch)
      if (a.mat_base + m < a.n_materials)
        a.acc_out[(size_t)(a.mat_base + m) * a.n_local_views * a.g.n_rows * a.g.n_channels + ray] = l;
      continue;
    }
    if (a.pathlen) a.pathlen[ray * a.n_materials + m] = l;
    L2[m] = l * 1.44269504088896340736f;
  }
  if (a.acc_out) return;
  if (w2) cone_detect_store<NM, true>(a, L2, mu, w, w2, bm, ray, v, r, c);
  else cone_detect_store<NM, false>(a, L2, mu, w, w2, bm, ray, v, r, c);
}

// ---------------------------------------------------------------------------------------------
// cone_cols_kernel (round 3): cone_rows_kernel with the voxel columns of a batch of slabs STAGED IN LDS.
//
// What bounded cone_rows_kernel was not the vector pipe (halving its instructions moved nothing) but the byte loads: a
// wave instruction that fetches one byte per lane keeps the CU's texture path busy for 5-10 cycles, and a slab needs 2
// (4 with a v-crossing) per wave.  Here the workgroup copies the two voxel columns of each slab of a batch - whole
// columns, cone_zs(nz) bytes, 16 bytes per lane and load - into LDS once (double buffered: the next batch's loads are in
// flight while this one is consumed), and the lanes read their bytes with ds_read_u8 at a static offset per (slab,
// column): 0.23 global loads per wave and slab instead of 2.8, no v_readfirstlane of the column offsets, the same ids in
// the same order - path lengths stay bit-identical.  CB = bytes reserved per staged column (>= cone_zs(nz)); kB = slabs
// per batch.  What bounds THIS loop is the latency of the staged loads (a batch waits for the next batch's columns), so
// waves in flight decide: kB = 4 with the register allocation capped at 64 VGPRs (8 waves per SIMD, 17 KB of LDS) runs
// the benchmark scan in 9.6 ms, kB = 8 (92 VGPRs, 26 KB: 5 waves) in 10.4 (profiles/r03_kernels.md).
template <int NM, int kB, int CB>
__global__ __launch_bounds__(kConeRows) __attribute__((amdgpu_waves_per_eu(kB == 4 && CB <= 544 ? 8 : 5, 8)))
void cone_cols_kernel(ConeArgs a, const uint8_t* __restrict__ vol_zc, const float* __restrict__ mu,
                      const float* __restrict__ w, const float* __restrict__ w2, int n_chunks, int view_tile) {
  constexpr int kBufB = 2 * kB * CB;                       // bytes of one staging buffer: [slab][b column, a column][CB]
  constexpr int kItems = (2 * kB * (CB / 16) + kConeRows - 1) / kConeRows;      // 16-byte pieces per lane and batch
  __shared__ ConeRec rec[kConeRows];
  __shared__ float lds_corr[4][kConeRows];                 // [id][lane]; id 3 = outside the grid: a cell nobody reads
  __shared__ __attribute__((aligned(16))) uint8_t cols[2 * kBufB];
  __shared__ unsigned long long vxm[kConeRows / 64];       // bit s: slab record s has a v-crossing (two different columns)
  const int tid = threadIdx.x;
#pragma unroll
  for (int m = 0; m < 4; ++m) lds_corr[m][tid] = 0.0f;     // each lane only ever touches its own cells
  // block -> (view, channel, row chunk): as cone_rows_kernel
  const uint32_t nblk = gridDim.x, bid = blockIdx.x, per = nblk >> 3;
  const uint32_t logical = (bid < (per << 3)) ? (bid & 7u) * per + (bid >> 3) : bid;
  const uint32_t group = (uint32_t)view_tile * a.g.n_channels * n_chunks;
  const uint32_t gq = logical / group, rem = logical - gq * group;
  const uint32_t views_here = min((uint32_t)view_tile, (uint32_t)a.n_local_views - gq * view_tile);
  const int chunk = rem % n_chunks;
  const uint32_t qq = rem / n_chunks;
  const int v = gq * view_tile + qq % views_here, c = qq / views_here;
  const int r = chunk * kConeRows + tid;
  const bool live = r < a.g.n_rows;
  const dexct_ray_plan p = a.plan[(size_t)v * a.g.n_channels + c];      // uniform
  const int axis = p.flags & 1u;
  const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
  const int nv = axis == 0 ? a.g.ny : a.g.nx;
  const uint32_t zs = cone_zs(a.g.nz);                                   // bytes per column (<= CB, checked by the host)
  const uint32_t su = (axis == 0 ? 1u : (uint32_t)a.g.nx) * zs;
  const uint32_t sv = (axis == 0 ? (uint32_t)a.g.nx : 1u) * zs;
  const uint32_t col_out = (uint32_t)a.g.nx * (uint32_t)a.g.ny * zs;    // the all-24 column
  // ---- z part of the plan, float64, operation for operation as cone_kernel / the oracle's cone_row_plan
  const int view = a.view_begin + v;
  const double cb = a.view_cs[2 * view], sb = a.view_cs[2 * view + 1];
  const double cg = a.chan_cs[2 * c], sg = a.chan_cs[2 * c + 1];
  const double sx = a.g.sid * cb, sy = a.g.sid * sb;
  const double ex = -(cb * cg - sb * sg), ey = -(sb * cg + cb * sg);
  const double su_d = axis == 0 ? sx / a.g.dx + 0.5 * a.g.nx : sy / a.g.dy + 0.5 * a.g.ny;
  const double eu = axis == 0 ? ex / a.g.dx : ey / a.g.dy;
  const double det_z = a.row_z[live ? r : 0];
  const double kOne = 1099511627776.0;
  const double ws = a.src_z / a.g.dz + 0.5 * a.g.nz;
  const double sw = ((det_z - a.src_z) / a.g.dz) / (a.g.sdd * eu);
  const double w0 = ws - su_d * sw;
  const long long SW = llrint(sw * kOne);
  const long long W0 = llrint(w0 * kOne);
  double inv = 16777216.0;
  if (SW != 0) inv = fmin(kOne / fabs((double)SW), 16777216.0);
  const float kfw = (float)(inv * (1.0 / 4294967296.0));
  const double tz = (det_z - a.src_z) / a.g.sdd;
  const float len3d = (float)((1.0 / fabs(eu)) * sqrt(1.0 + tz * tz));
  const uint32_t wpos = SW > 0 ? 0xFFFFFFFFu : 0u;

  // ---- this lane's share of the staging: pieces tid, tid + 256, ... of the 2 kB columns x (zs / 16) pieces of a batch
  const uint32_t n16 = zs >> 4, n_pieces = 2u * kB * n16;
  uint32_t pc_rec[kItems], pc_src[kItems], pc_dst[kItems];
  bool pc_on[kItems];
#pragma unroll
  for (int t = 0; t < kItems; ++t) {
    const uint32_t piece = (uint32_t)tid + (uint32_t)t * kConeRows;
    pc_on[t] = piece < n_pieces;
    const uint32_t seg = pc_on[t] ? piece / n16 : 0u, q16 = pc_on[t] ? piece - seg * n16 : 0u;    // seg = 2 slab + (0: b column, 1: a column)
    pc_rec[t] = (seg >> 1) * (uint32_t)sizeof(ConeRec) + (seg & 1u) * 4u;      // byte offset of colb / cola in rec[]
    pc_src[t] = q16 * 16u;
    pc_dst[t] = seg * (uint32_t)CB + q16 * 16u;
  }
  const uint8_t* rec_bytes = reinterpret_cast<const uint8_t*>(rec);
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));      // (HIP's uint4 keeps the array in scratch)
  u32x4 staged[kItems];
  auto stage_load = [&](int s) {                 // the global loads of the batch that starts at record s
#pragma unroll
    for (int t = 0; t < kItems; ++t) {           // a lane without a piece t loads piece 0 of the batch again and does not store it
      const uint32_t col = *reinterpret_cast<const uint32_t*>(rec_bytes + pc_rec[t] + (uint32_t)s * (uint32_t)sizeof(ConeRec));
      staged[t] = *reinterpret_cast<const u32x4*>(vol_zc + (size_t)(col + pc_src[t]));
    }
  };
  auto stage_store = [&](uint32_t bufoff) {
#pragma unroll
    for (int t = 0; t < kItems; ++t)
      if (pc_on[t]) *reinterpret_cast<u32x4*>(cols + bufoff + pc_dst[t]) = staged[t];
  };

  uint32_t acc = 0;                       // four byte counters: codes 0..3 of the b voxels since the last flush
  uint32_t cnt[3] = {0, 0, 0};
  // W carries the guard and the offset of the staging buffer in use: (W >> 40) is the LDS byte offset of the lane's
  // voxel inside column 0 of that buffer.  Only the fraction bits enter the exact path below.
  // The fast path reads its bytes with ds_read_u8 from inline assembly at RAW LDS addresses: the base of cols[] (wherever
  // the compiler placed the array - nothing in the source pins it to 0) rides in W together with the guard and the
  // buffer offset, and in the clamp bounds; the exact path, which indexes cols[] in C++, takes it off again.
  const uint32_t cbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)cols;
  long long W = W0 + (long long)p.i_first * SW + ((long long)(kConeGuard + (int)cbase) << DEXCT_FIX_FRAC);
  uint32_t bufoff = 0;
  for (int s0 = 0; s0 < p.n_slabs; s0 += kConeRows) {
    const int n_here = min(kConeRows, p.n_slabs - s0);
    const int n_pad = (n_here + kB - 1) / kB * kB;      // the last batch is filled with null records (all "outside")
    {
      ConeRec q{col_out, col_out, 0.0f, 0u};
      if (tid < n_here) {
        const int i = p.i_first + s0 + tid;
        const SlabPieces sp = dda_slab(p.V0 + (long long)i * p.SV, p.SV, smask, p.kf);
        const bool ina = (uint32_t)sp.ja < (uint32_t)nv, inb = (uint32_t)sp.jb < (uint32_t)nv;
        q.cola = ina ? (uint32_t)i * su + (uint32_t)sp.ja * sv : col_out;
        q.colb = inb ? (uint32_t)i * su + (uint32_t)sp.jb * sv : col_out;
        q.tv = sp.t;
      }
      rec[tid] = q;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(q.cola != q.colb);
      if ((tid & 63) == 0) vxm[tid >> 6] = m;
    }
    __syncthreads();
    stage_load(0);
    stage_store(bufoff);
    __syncthreads();
    const uint8_t* vx_bytes = reinterpret_cast<const uint8_t*>(vxm);
    // two levels so that the flush of the byte counters (they hold at most 128 slabs) needs no test inside the batch loop
    for (int s1 = 0; s1 < n_pad; s1 += 128) {
      const int s_end = min(n_pad, s1 + 128);
      for (int s = s1; s < s_end; s += kB) {
        const bool more = s + kB < n_pad;                        // uniform
        if (more) stage_load(s + kB);                            // in flight while this batch is consumed
        static_assert(kB == 8 || kB == 4, "a byte or a nibble of the crossing mask per batch");
        const uint32_t vxb = (uint32_t)__builtin_amdgcn_readfirstlane((int)vx_bytes[s >> 3]) >> (kB == 4 ? (s & 4) : 0);
        const int hi = (int)(cbase + bufoff + zs - 1u);
        int lo = (int)(cbase + bufoff);
        asm volatile("" : "+v"(lo));                             // a VGPR: v_med3_i32 takes one scalar operand only
        auto slice = [&](long long Wx) {                         // clamp((int)(Wx >> 40), lo, hi): v_ashr + v_med3_i32
          int kq;
          asm("v_med3_i32 %0, %1, %2, %3" : "=v"(kq) : "v"((int)(Wx >> DEXCT_FIX_FRAC)), "v"(lo), "s"(hi));
          return kq;
        };
        // ---- fast path, four slabs at a time (all LDS reads of the four first): count the b voxel (the layout stores
        // 8 * id: one v_lshl_add) and compare the ids into LANE MASKS.  Nothing but the masks is kept: the rare exact path
        // reads its bytes again.  The byte reads are issued from inline assembly (ds_read_u8 zero-extends; the compiler
        // adds a v_and per byte) and waited for by an explicit s_waitcnt that takes the registers as in/out operands.
        const long long Wb = W;
        unsigned long long dm[kB], any = 0ull;
        int k0 = slice(W);
        uint32_t vxc = vxb;
        asm volatile("" : "+s"(vxc));                            // opaque copy for the compare phase: tests stay s_bitcmp + branch
        auto half = [&](auto hc) {
          constexpr int h = decltype(hc)::value;
          uint32_t x[4], xa[4], c1[4], c2[4];
          auto slab = [&](auto jc) {
            constexpr int jj = decltype(jc)::value, j = h + jj;
            W += SW;
            int k1;                                              // the clamp of the slice rides in the same statement: no pad between
            asm volatile("v_med3_i32 %2, %3, %4, %5\n\tds_read_u8 %0, %2 offset:%7\n\tds_read_u8 %1, %6 offset:%8"
                         : "=&v"(x[jj]), "=&v"(xa[jj]), "=&v"(k1)
                         : "v"((int)(W >> DEXCT_FIX_FRAC)), "v"(lo), "s"(hi), "v"(k0), "n"((2 * j) * CB),
                           "n"((2 * j + 1) * CB));                // b = (jb, kb), a = (ja, ka)
            if ((vxb >> j) & 1u)                                 // uniform: the two possible middle voxels (jb, ka), (ja, kb)
              asm volatile("ds_read_u8 %0, %2 offset:%4\n\tds_read_u8 %1, %3 offset:%5"
                           : "=&v"(c1[jj]), "=&v"(c2[jj])
                           : "v"(k0), "v"(k1), "n"((2 * j) * CB), "n"((2 * j + 1) * CB));
            k0 = k1;
          };
          slab(std::integral_constant<int, 0>{});
          slab(std::integral_constant<int, 1>{});
          slab(std::integral_constant<int, 2>{});
          slab(std::integral_constant<int, 3>{});
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(x[0]), "+v"(xa[0]), "+v"(x[1]), "+v"(xa[1]), "+v"(x[2]), "+v"(xa[2]), "+v"(x[3]), "+v"(xa[3]));
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = h + jj;
            asm("v_lshl_add_u32 %0, 1, %1, %0" : "+v"(acc) : "v"(x[jj]));      // acc += 1 << x (left alone, the compiler shifts all four and adds them in a tree: 1.5 per slab)
            dm[j] = __builtin_amdgcn_ballot_w64(xa[jj] != x[jj]);
            if ((vxc >> j) & 1u) {
              asm volatile("" : "+v"(c1[jj]), "+v"(c2[jj]));     // ordered behind the wait above
              dm[j] |= __builtin_amdgcn_ballot_w64(c1[jj] != x[jj]) | __builtin_amdgcn_ballot_w64(c2[jj] != x[jj]);
            }
            any |= dm[j];
          }
        };
        half(std::integral_constant<int, 0>{});
        if constexpr (kB == 8) half(std::integral_constant<int, 4>{});
        if (any != 0ull) {
#pragma unroll
          for (int j = 0; j < kB; ++j) {
            if (dm[j] != 0ull && __builtin_amdgcn_inverse_ballot_w64(dm[j])) {
              // the oracle's slab, operation for operation (orc_cone_pathlen); only bits 8..39 of W enter
              const long long Wj = Wb + (long long)j * SW;
              const int ka = slice(Wj) - (int)cbase, kb = slice(Wj + SW) - (int)cbase;      // (cols[] is indexed from its own start)
              const uint32_t ida = cols[ka + (2 * j + 1) * CB], idb = cols[kb + (2 * j) * CB];
              const float tv = rec[s + j].tv;
              const float tw = fminf((float)((uint32_t)((unsigned long long)Wj >> 8) ^ wpos) * kfw, 1.0f);
              const float t1 = fminf(tv, tw), t2 = fmaxf(tv, tw);
              const bool v_first = tv <= tw;
              // middle voxel (jm, km) = v_first ? (jb, ka) : (ja, kb); without a v-crossing both columns are the same one
              const uint32_t idm = v_first ? cols[ka + (2 * j) * CB] : cols[kb + (2 * j + 1) * CB];
              if (ida != idm || idm != idb) {
                float* cell = &lds_corr[0][tid];                 // ids are stored as 8 * id: row id = cell + (8 id) * 32
                static_assert(kConeRows == 8 * 32, "cell offset of a stored id");
                __hip_atomic_fetch_add(cell + idm * 32u, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(cell + idb * 32u, -t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(cell + ida * 32u, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(cell + idm * 32u, -t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            }
          }
        }
        // ---- the next batch goes to the other buffer; the lane's W follows it
        if (more) {
          const uint32_t next = bufoff ^ (uint32_t)kBufB;        // 0 <-> kBufB
          stage_store(next);
          W += (long long)((int)next - (int)bufoff) << DEXCT_FIX_FRAC;
          bufoff = next;
        }
        __syncthreads();
      }
#pragma unroll
      for (int m = 0; m < NM; ++m) cnt[m] += (acc >> (8 * m)) & 0xFFu;
      acc = 0;
    }
  }
  const BlockMasks bm = detect_block_masks(w, a.n_energies, a.n_spectra);      // a ballot: before dead rows leave
  if (!live) return;
  // ---- detection (same weighting as the other kernels)
  const size_t ray = ((size_t)v * a.g.n_rows + r) * a.g.n_channels + c;
  float L2[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) {
    const float l = ((float)(int32_t)cnt[m] + lds_corr[m][tid]) * len3d;
    if (a.acc_out) {         // group pass: lengths out, detection later (uniform branch)
      if (a.mat_base + m < a.n_materials)
        a.acc_out[(size_t)(a.mat_base + m) * a.n_local_views * a.g.n_rows * a.g.n_channels + ray] = l;
      continue;
    }
    if (a.pathlen) a.pathlen[ray * a.n_materials + m] = l;
    L2[m] = l * 1.44269504088896340736f;
  }
  if (a.acc_out) return;
  if (w2) cone_detect_store<NM, true>(a, L2, mu, w, w2, bm, ray, v, r, c);
  else cone_detect_store<NM, false>(a, L2, mu, w, w2, bm, ray, v, r, c);
}

// vol [nz][ny][nx] -> guarded z-fastest layout [(ny*nx + 1)][cone_zs(nz)] of 8 * id (the shift of the packed byte counter the
// traversal adds with one v_lshl_add; round 3): guard slices and the extra column hold 24 = "outside", id 3.
__global__ __launch_bounds__(256) void cone_layout_kernel(const uint8_t* __restrict__ vol, int nx, int ny, int nz,
                                                          uint8_t* __restrict__ out) {
  const size_t zs = cone_zs(nz);
  const size_t total = ((size_t)nx * ny + 1) * zs;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const size_t col = i / zs;
  const int kz = (int)(i - col * zs) - kConeGuard;
  uint8_t val = 24;
  if (col < (size_t)nx * ny && kz >= 0 && kz < nz) val = (uint8_t)(vol[(size_t)kz * nx * ny + col] << 3);      // col = y*nx + x
  out[i] = val;
}

// the noise arguments of the cone entry points (struct dexct_noise): weights2 switches the variance on and needs somewhere for
// it to go; the log of a noisy sinogram is the log of the SAMPLED counts, so log_out needs the sample
static int cone_noise_args(const float* weights2, const float* variance, const dexct_noise* noise, const dexct_log_out* log_out) {
  const bool sample = noise && noise->sample;
  if (!weights2) return (variance || sample) ? DEXCT_EINVAL : DEXCT_OK;
  if (!variance && !sample) return DEXCT_EINVAL;
  if (!sample && log_out && log_out->sino_log) return DEXCT_EINVAL;
  return DEXCT_OK;
}

static void set_cone_noise(ConeArgs& a, float* variance, const dexct_noise* noise) {
  const bool sample = noise && noise->sample;
  a.variance = variance;
  a.sample = sample ? 1 : 0;
  a.seed_lo = sample ? (uint32_t)noise->seed : 0u;
  a.seed_hi = sample ? (uint32_t)(noise->seed >> 32) : 0u;
}

// the guarded layout once per group of three materials: layout g holds 8 * (id - 3 g) for ids 3g .. 3g + 2 and 24 ("outside":
// a voxel that belongs to nobody) for every other id - the row kernels then accumulate exactly the sums of these three
// materials (the per-material sums of orc_cone_pathlen are independent of each other)
__global__ __launch_bounds__(256) void cone_layout_groups_kernel(const uint8_t* __restrict__ vol, int nx, int ny, int nz, int n_groups,
                                                                 uint8_t* __restrict__ out) {
  const size_t zs = cone_zs(nz);
  const size_t total = ((size_t)nx * ny + 1) * zs;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const size_t col = i / zs;
  const int kz = (int)(i - col * zs) - kConeGuard;
  const bool in = col < (size_t)nx * ny && kz >= 0 && kz < nz;
  const uint32_t id = in ? vol[(size_t)kz * nx * ny + col] : 0u;
  for (int g = 0; g < n_groups; ++g) {
    const uint32_t rel = id - 3u * (uint32_t)g;
    out[(size_t)g * total + i] = (in && rel < 3u) ? (uint8_t)(rel << 3) : (uint8_t)24;
  }
}

}  // namespace dexct

using namespace dexct;

extern "C" int dexct_cone_project(const dexct_fan_geom* geom, const dexct_ray_plan* plan, const double* view_cs,
                                  const double* chan_cs, const double* row_z, double src_z, double max_abs_dz,
                                  int32_t view_begin, int32_t view_end, const uint8_t* vol_yx, const uint8_t* vol_xy,
                                  int32_t n_materials, int32_t n_energies, int32_t n_spectra, const float* mu,
                                  const float* weights, float* counts, float* pathlen, const dexct_log_out* log_out,
                                  const float* weights2, float* variance, const dexct_noise* noise, void* stream) {
  if (!geom || !plan || !view_cs || !chan_cs || !row_z || !vol_yx || !vol_xy || !mu || !weights || !counts)
    return DEXCT_EINVAL;
  if (cone_noise_args(weights2, variance, noise, log_out) != DEXCT_OK) return DEXCT_EINVAL;
  if (view_begin < 0 || view_end > geom->n_views || view_end <= view_begin) return DEXCT_EINVAL;
  if (n_materials < 1 || n_energies < 1 || n_spectra < 1 || geom->n_rows < 1) return DEXCT_EINVAL;
  if (n_materials > DEXCT_MAX_MATERIALS || n_spectra > DEXCT_MAX_SPECTRA) return DEXCT_ERANGE;
  if ((uint64_t)geom->nx * geom->ny * geom->nz > 0xFFFFFFFEull) return DEXCT_ERANGE;
  if (geom->n_rows > 65535 || view_end - view_begin > 65535) return DEXCT_ERANGE;
  // at most one z-plane per dominant-axis slab: |dz/du| <= 1 for every ray.  du/ds >= 1/(sqrt(2) max(dx, dy))
  // along the dominant axis, dz/ds <= max|z_det - z_src| / SDD (max_abs_dz is supplied by the caller, who
  // owns row_z on the device).
  const double dmax = geom->dx > geom->dy ? geom->dx : geom->dy;
  if (!(max_abs_dz >= 0) || max_abs_dz / geom->sdd / geom->dz * dmax * 1.4142135623730951 > 1.0) return DEXCT_ERANGE;
  ConeArgs a;
  a.g = *geom;
  a.plan = plan;
  a.view_cs = view_cs;
  a.chan_cs = chan_cs;
  a.row_z = row_z;
  a.src_z = src_z;
  a.vol_yx = vol_yx;
  a.vol_xy = vol_xy;
  a.view_begin = view_begin;
  a.n_local_views = view_end - view_begin;
  a.n_materials = n_materials;
  a.n_energies = n_energies;
  a.n_spectra = n_spectra;
  a.counts = counts;
  a.pathlen = pathlen;
  a.sino_log = log_out ? log_out->sino_log : nullptr;
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) a.air[s] = log_out ? log_out->air[s] : 1.0f;
  set_cone_noise(a, variance, noise);
  hipStream_t st = as_stream(stream);
  switch (n_materials) {
    case 1: return launch_cone<1>(a, mu, weights, weights2, st);
    case 2: return launch_cone<2>(a, mu, weights, weights2, st);
    case 3: return launch_cone<3>(a, mu, weights, weights2, st);
    case 4: return launch_cone<4>(a, mu, weights, weights2, st);
    default: return launch_cone<0>(a, mu, weights, weights2, st);
  }
}


extern "C" int64_t dexct_cone_layout_bytes(int32_t nx, int32_t ny, int32_t nz) {
  if (nx <= 0 || ny <= 0 || nz <= 0) return 0;
  return ((int64_t)nx * ny + 1) * (int64_t)cone_zs(nz);
}

extern "C" int dexct_cone_layout(const uint8_t* vol, int32_t nx, int32_t ny, int32_t nz, uint8_t* vol_zc, void* stream) {
  if (!vol || !vol_zc || nx <= 0 || ny <= 0 || nz <= 0) return DEXCT_EINVAL;
  const int64_t total = dexct_cone_layout_bytes(nx, ny, nz);
  if (total > 0xFFFFFFFEll) return DEXCT_ERANGE;
  const int64_t nblk = (total + 255) / 256;
  hipLaunchKernelGGL(cone_layout_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), vol, nx, ny, nz, vol_zc);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

// what both row-kernel entry points check
static int cone_rows_checks(const dexct_fan_geom* geom, int32_t view_begin, int32_t view_end, int32_t n_energies, int32_t n_spectra,
                            double max_abs_dz) {
  if (view_begin < 0 || view_end > geom->n_views || view_end <= view_begin) return DEXCT_EINVAL;
  if (n_energies < 1 || n_spectra < 1 || geom->n_rows < 1) return DEXCT_EINVAL;
  if (n_spectra > DEXCT_MAX_SPECTRA) return DEXCT_ERANGE;
  if (dexct_cone_layout_bytes(geom->nx, geom->ny, geom->nz) > 0xFFFFFFFEll) return DEXCT_ERANGE;
  if (geom->n_rows > 65535 || view_end - view_begin > 65535) return DEXCT_ERANGE;
  const double dmax = geom->dx > geom->dy ? geom->dx : geom->dy;
  if (!(max_abs_dz >= 0) || max_abs_dz / geom->sdd / geom->dz * dmax * 1.4142135623730951 > 1.0) return DEXCT_ERANGE;
  return DEXCT_OK;
}

// one launch of the row kernels on the layout vol_zc; nm = the materials (codes) the layout holds: 1..3
static int launch_cone_rows(const ConeArgs& a, const uint8_t* vol_zc, int nm, const float* mu, const float* weights,
                            const float* weights2, hipStream_t st) {
  const dexct_fan_geom* geom = &a.g;
  const int n_chunks = (geom->n_rows + kConeRows - 1) / kConeRows;
  const size_t nblk = (size_t)a.n_local_views * geom->n_channels * n_chunks;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  // views per tile of the block order: 1 = all channels of a view before the next view (round 3: 9.6 -> 9.2 ms on a
  // 100-view scan, whose views 3.6 degrees apart share no columns; no difference at 0.36 degrees)
  int view_tile = 1;
  if (const char* te = getenv("DEXCT_CONE_VIEW_TILE")) view_tile = atoi(te) > 0 ? atoi(te) : 1;      // tuning knob
  // round 3: columns staged in LDS (cone_cols_kernel) whenever a column fits the reserved bytes; DEXCT_CONE_COLS=0 = A/B
  const char* ce = getenv("DEXCT_CONE_COLS");
  const uint32_t zs = cone_zs(geom->nz);
  if (!(ce && atoi(ce) == 0) && zs <= 1056u) {
    // slabs per staged batch: 4 (default: 64 VGPRs and 17 KB of LDS = 8 waves per SIMD; 9.6 ms) or 8 (92 VGPRs, 26 KB:
    // 5 waves; 10.4 ms) - the loop waits for the staged loads of the next batch, so waves in flight are what counts
    const char* ke = getenv("DEXCT_CONE_KB");
    const bool kb4 = !(ke && atoi(ke) == 8);
#define DEXCT_CONE_COLS_LAUNCH(NM_, CB_) \
    do { \
      if (kb4) hipLaunchKernelGGL((cone_cols_kernel<NM_, 4, CB_>), dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile); \
      else hipLaunchKernelGGL((cone_cols_kernel<NM_, 8, CB_>), dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile); \
    } while (0)
    if (zs <= 288u) {
      switch (nm) {
        case 1: DEXCT_CONE_COLS_LAUNCH(1, 288); break;
        case 2: DEXCT_CONE_COLS_LAUNCH(2, 288); break;
        default: DEXCT_CONE_COLS_LAUNCH(3, 288); break;
      }
    } else if (zs <= 544u) {
      switch (nm) {
        case 1: DEXCT_CONE_COLS_LAUNCH(1, 544); break;
        case 2: DEXCT_CONE_COLS_LAUNCH(2, 544); break;
        default: DEXCT_CONE_COLS_LAUNCH(3, 544); break;
      }
    } else {                                   // up to 1024 slices: 4 slabs per batch only (17 KB of LDS per buffer)
      switch (nm) {
        case 1: hipLaunchKernelGGL((cone_cols_kernel<1, 4, 1056>), dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile); break;
        case 2: hipLaunchKernelGGL((cone_cols_kernel<2, 4, 1056>), dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile); break;
        default: hipLaunchKernelGGL((cone_cols_kernel<3, 4, 1056>), dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile); break;
      }
    }
#undef DEXCT_CONE_COLS_LAUNCH
    DEXCT_LAUNCH_CHECK();
    return DEXCT_OK;
  }
  switch (nm) {
    case 1: hipLaunchKernelGGL(cone_rows_kernel<1>, dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile); break;
    case 2: hipLaunchKernelGGL(cone_rows_kernel<2>, dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile); break;
    default: {
      const char* e = getenv("DEXCT_CONE_BATCH");         // tuning knob
      const int kb = e ? atoi(e) : 4;
      const char* le = getenv("DEXCT_CONE_LDSC");         // 0: corrections in registers (the round-2 form), for A/B
      if (le && atoi(le) == 0) hipLaunchKernelGGL((cone_rows_kernel<3, 4, false>), dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile);
      else if (kb == 8) hipLaunchKernelGGL((cone_rows_kernel<3, 8>), dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile);
      else if (kb == 2) hipLaunchKernelGGL((cone_rows_kernel<3, 2>), dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile);
      else hipLaunchKernelGGL((cone_rows_kernel<3, 4>), dim3((unsigned)nblk), dim3(kConeRows), 0, st, a, vol_zc, mu, weights, weights2, n_chunks, view_tile);
      break;
    }
  }
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

static void fill_cone_args(ConeArgs& a, const dexct_fan_geom* geom, const dexct_ray_plan* plan, const double* view_cs,
                           const double* chan_cs, const double* row_z, double src_z, int32_t view_begin, int32_t view_end,
                           int32_t n_materials, int32_t n_energies, int32_t n_spectra, float* counts, float* pathlen,
                           const dexct_log_out* log_out) {
  a.g = *geom;
  a.plan = plan;
  a.view_cs = view_cs;
  a.chan_cs = chan_cs;
  a.row_z = row_z;
  a.src_z = src_z;
  a.vol_yx = nullptr;
  a.vol_xy = nullptr;
  a.view_begin = view_begin;
  a.n_local_views = view_end - view_begin;
  a.n_materials = n_materials;
  a.n_energies = n_energies;
  a.n_spectra = n_spectra;
  a.counts = counts;
  a.pathlen = pathlen;
  a.sino_log = log_out ? log_out->sino_log : nullptr;
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) a.air[s] = log_out ? log_out->air[s] : 1.0f;
}

extern "C" int dexct_cone_project_rows(const dexct_fan_geom* geom, const dexct_ray_plan* plan, const double* view_cs,
                                       const double* chan_cs, const double* row_z, double src_z, double max_abs_dz,
                                       int32_t view_begin, int32_t view_end, const uint8_t* vol_zc,
                                       int32_t n_materials, int32_t n_energies, int32_t n_spectra, const float* mu,
                                       const float* weights, float* counts, float* pathlen,
                                       const dexct_log_out* log_out, const float* weights2, float* variance,
                                       const dexct_noise* noise, void* stream) {
  if (!geom || !plan || !view_cs || !chan_cs || !row_z || !vol_zc || !mu || !weights || !counts) return DEXCT_EINVAL;
  if (cone_noise_args(weights2, variance, noise, log_out) != DEXCT_OK) return DEXCT_EINVAL;
  if (n_materials < 1) return DEXCT_EINVAL;
  if (n_materials > 3) return DEXCT_ERANGE;      // code 3 is "outside the grid" (more: dexct_cone_project_grouped)
  const int rc = cone_rows_checks(geom, view_begin, view_end, n_energies, n_spectra, max_abs_dz);
  if (rc != DEXCT_OK) return rc;
  ConeArgs a;
  fill_cone_args(a, geom, plan, view_cs, chan_cs, row_z, src_z, view_begin, view_end, n_materials, n_energies, n_spectra, counts,
                 pathlen, log_out);
  set_cone_noise(a, variance, noise);
  return launch_cone_rows(a, vol_zc, n_materials, mu, weights, weights2, as_stream(stream));
}

extern "C" int dexct_cone_layout_groups(const uint8_t* vol, int32_t nx, int32_t ny, int32_t nz, int32_t n_materials,
                                        uint8_t* vol_zcg, void* stream) {
  if (!vol || !vol_zcg || nx <= 0 || ny <= 0 || nz <= 0) return DEXCT_EINVAL;
  if (n_materials < 1 || n_materials > DEXCT_MAX_MATERIALS) return DEXCT_ERANGE;
  const int64_t total = dexct_cone_layout_bytes(nx, ny, nz);
  if (total > 0xFFFFFFFEll) return DEXCT_ERANGE;
  const int n_groups = (n_materials + 2) / 3;
  const int64_t nblk = (total + 255) / 256;
  hipLaunchKernelGGL(cone_layout_groups_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), vol, nx, ny, nz, n_groups,
                     vol_zcg);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

extern "C" int dexct_cone_project_grouped(const dexct_fan_geom* geom, const dexct_ray_plan* plan, const double* view_cs,
                                          const double* chan_cs, const double* row_z, double src_z, double max_abs_dz,
                                          int32_t view_begin, int32_t view_end, const uint8_t* vol_zcg, int32_t n_materials,
                                          int32_t n_energies, int32_t n_spectra, const float* mu, const float* weights,
                                          float* counts, float* pathlen, float* acc_scratch, const dexct_log_out* log_out,
                                          const float* weights2, float* variance, const dexct_noise* noise, void* stream) {
  if (!geom || !plan || !view_cs || !chan_cs || !row_z || !vol_zcg || !mu || !weights || !counts || !acc_scratch) return DEXCT_EINVAL;
  if (n_materials < 1) return DEXCT_EINVAL;
  if (n_materials > DEXCT_MAX_MATERIALS) return DEXCT_ERANGE;
  const int rc = cone_rows_checks(geom, view_begin, view_end, n_energies, n_spectra, max_abs_dz);
  if (rc != DEXCT_OK) return rc;
  hipStream_t st = as_stream(stream);
  // ---- one traversal per group of three materials: lengths [cm] into the planes of acc_scratch
  ConeArgs a;
  fill_cone_args(a, geom, plan, view_cs, chan_cs, row_z, src_z, view_begin, view_end, n_materials, n_energies, n_spectra, counts,
                 nullptr, nullptr);
  set_cone_noise(a, nullptr, nullptr);
  a.acc_out = acc_scratch;
  const size_t layout_bytes = (size_t)dexct_cone_layout_bytes(geom->nx, geom->ny, geom->nz);
  const int n_groups = (n_materials + 2) / 3;
  for (int g = 0; g < n_groups; ++g) {
    a.mat_base = 3 * g;
    const int left = n_materials - 3 * g;
    const int lrc = launch_cone_rows(a, vol_zcg + (size_t)g * layout_bytes, left >= 3 ? 3 : left, mu, weights, nullptr, st);
    if (lrc != DEXCT_OK) return lrc;
  }
  // ---- one detection pass over all planes (the pass of the stacked fan's material groups, reading lengths)
  ProjArgs d;
  d.g = *geom;
  d.plan = plan;
  d.vol_yx = d.vol_xy = d.vol_zf = nullptr;
  d.n_local_views = view_end - view_begin;
  d.n_materials = n_materials;
  d.n_energies = n_energies;
  d.n_spectra = n_spectra;
  d.counts = counts;
  d.pathlen = pathlen;
  d.variance = variance;
  d.view_tile = 1;
  d.layout = 0;
  d.acc_out = acc_scratch;
  d.mat_base = 0;
  d.acc_lengths = 1;
  { const int nrc = set_noise(d, view_begin, weights2, variance, noise); if (nrc != DEXCT_OK) return nrc; }
  if (set_log_out(d, log_out, d.sample ? nullptr : variance) != DEXCT_OK) return DEXCT_EINVAL;
  const Tables t{mu, weights, weights2};
  return launch_detect_any(d, t, st);
}
