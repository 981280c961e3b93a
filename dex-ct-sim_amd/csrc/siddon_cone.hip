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
#include "common.h"

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
};

template <int NM>
__global__ __launch_bounds__(kConeBlock) void cone_kernel(ConeArgs a, const float* __restrict__ mu,
                                                          const float* __restrict__ w) {
  extern __shared__ float lds_acc[];     // NM == 0: counts then corrections, [n_materials][kConeBlock] each
  const int tid = threadIdx.x;
  const int c = blockIdx.x * kConeBlock + tid;
  const int r = blockIdx.y, v = blockIdx.z;
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
  // NM > 0: counts and corrections in registers; NM == 0: per-lane LDS columns [n_materials][kConeBlock] of each
  int32_t cnt[NM > 0 ? NM : 1];
  float corr[NM > 0 ? NM : 1];
  int32_t* lds_cnt = reinterpret_cast<int32_t*>(lds_acc);
  float* lds_corr = lds_acc + (size_t)a.n_materials * kConeBlock;
  if (NM > 0) {
#pragma unroll
    for (int m = 0; m < (NM > 0 ? NM : 1); ++m) { cnt[m] = 0; corr[m] = 0.0f; }
  } else {
    for (int m = 0; m < a.n_materials; ++m) { lds_cnt[m * kConeBlock + tid] = 0; lds_corr[m * kConeBlock + tid] = 0.0f; }
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
    auto voxel = [&](int32_t j, int32_t k) -> uint32_t {      // material id, 0xFF outside the grid (or an id >= n_materials)
      const bool in = (uint32_t)j < (uint32_t)nv && (uint32_t)k < (uint32_t)a.g.nz;
      return in ? (uint32_t)base[(uint32_t)k * slice + off + (uint32_t)j] : 0xFFu;
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
      if (idb < nm) lds_cnt[idb * kConeBlock + tid] += 1;       // a lane owns its column: plain read-modify-write
      if (ida != idm || idm != idb) {
        if (idm < nm) lds_corr[idm * kConeBlock + tid] += t2;
        if (idb < nm) lds_corr[idb * kConeBlock + tid] -= t2;
        if (ida < nm) lds_corr[ida * kConeBlock + tid] += t1;
        if (idm < nm) lds_corr[idm * kConeBlock + tid] -= t1;
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
  float accs[DEXCT_MAX_SPECTRA];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) accs[s] = 0.0f;
  int srow[DEXCT_MAX_SPECTRA];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) srow[s] = (s < a.n_spectra ? s : 0) * n_e;
  if (NM > 0) {
    float L2[NM > 0 ? NM : 1];
#pragma unroll
    for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
      const float l = ((float)cnt[m] + corr[m]) * len3d;
      if (a.pathlen) a.pathlen[ray * n_mat + m] = l;
      L2[m] = l * 1.44269504088896340736f;
    }
    for (int e = 0; e < n_e; ++e) {
      float pe = 0.0f;
#pragma unroll
      for (int m = 0; m < (NM > 0 ? NM : 1); ++m) pe = fmaf(mu[m * n_e + e], L2[m], pe);
      const float t = __builtin_amdgcn_exp2f(-pe);
#pragma unroll
      for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) accs[s] = fmaf(w[srow[s] + e], t, accs[s]);
    }
  } else {
    for (int m = 0; m < n_mat; ++m) {
      const float l = ((float)lds_cnt[m * kConeBlock + tid] + lds_corr[m * kConeBlock + tid]) * len3d;
      if (a.pathlen) a.pathlen[ray * n_mat + m] = l;
      lds_corr[m * kConeBlock + tid] = l * 1.44269504088896340736f;
    }
    for (int e = 0; e < n_e; ++e) {
      float pe = 0.0f;
      for (int m = 0; m < n_mat; ++m) pe = fmaf(mu[m * n_e + e], lds_corr[m * kConeBlock + tid], pe);
      const float t = __builtin_amdgcn_exp2f(-pe);
#pragma unroll
      for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) accs[s] = fmaf(w[srow[s] + e], t, accs[s]);
    }
  }
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
    if (s < a.n_spectra) a.counts[ray + s * sstride] = accs[s];
}

template <int NM>
static int launch_cone(const ConeArgs& a, const float* mu, const float* w, hipStream_t st) {
  dim3 grid((a.g.n_channels + kConeBlock - 1) / kConeBlock, a.g.n_rows, a.n_local_views);
  const size_t lds = NM > 0 ? 0 : (size_t)2 * a.n_materials * kConeBlock * sizeof(float);
  hipLaunchKernelGGL(cone_kernel<NM>, grid, dim3(kConeBlock), lds, st, a, mu, w);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

}  // namespace dexct

using namespace dexct;

extern "C" int dexct_cone_project(const dexct_fan_geom* geom, const dexct_ray_plan* plan, const double* view_cs,
                                  const double* chan_cs, const double* row_z, double src_z, double max_abs_dz,
                                  int32_t view_begin, int32_t view_end, const uint8_t* vol_yx, const uint8_t* vol_xy,
                                  int32_t n_materials, int32_t n_energies, int32_t n_spectra, const float* mu,
                                  const float* weights, float* counts, float* pathlen, void* stream) {
  if (!geom || !plan || !view_cs || !chan_cs || !row_z || !vol_yx || !vol_xy || !mu || !weights || !counts)
    return DEXCT_EINVAL;
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
  hipStream_t st = as_stream(stream);
  switch (n_materials) {
    case 1: return launch_cone<1>(a, mu, weights, st);
    case 2: return launch_cone<2>(a, mu, weights, st);
    case 3: return launch_cone<3>(a, mu, weights, st);
    case 4: return launch_cone<4>(a, mu, weights, st);
    default: return launch_cone<0>(a, mu, weights, st);
  }
}
