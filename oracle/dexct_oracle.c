/*
 * ORACLE - test infrastructure only.  Never linked into, imported by or timed as the product;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Plain-C CPU restatement of the dex-ct-sim hot path.
 *
 * PARITY STATUS
 *   Gauss-Newton half (orc_gn_*): restates /root/reference/matdecomp.py:87-127 and is pinned
 *   by golden vectors captured from the real reference (tests/golden/gn_reference.npz).
 *   Siddon half (orc_siddon_*, orc_project_*): PARITY UNPINNED.  The reference's forward
 *   projector lives in the un-vendored submodule gjadick/x-tomo-sim (path xtomosim, no pinned
 *   commit in the container; call sites main.py:19-21,120).  What is restated here is the
 *   published algorithm the reference names (README.md:27-28,41): R. L. Siddon, "Fast
 *   calculation of the exact radiological path for a three-dimensional CT array", Med. Phys.
 *   12(2) 252-255 (1985), for the fan-beam geometry of input/params.txt:18-27.  Its pins are
 *   analytic: chord lengths through boxes and discs, rotational symmetry, sum of segment
 *   lengths = chord inside the grid (tests/test_siddon_oracle.py).
 *
 * Two Siddon restatements live here on purpose:
 *   orc_siddon_classic_ray  float64, literally the 1985 paper: parametric plane crossings
 *                           alpha_x(i), alpha_y(j), merged and sorted, segment length
 *                           (alpha_k - alpha_{k-1}) * |P2 - P1|, voxel from the segment midpoint.
 *   orc_plan / orc_dda_ray  the slab-stepping fixed-point formulation the HIP kernels use
 *                           (same integer and float32 arithmetic, operation for operation), so
 *                           that voxel-index sequences and float32 segment lengths can be
 *                           compared bit for bit; it is itself checked against the classic form.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dexct.h"

#define FIX_ONE 1099511627776.0 /* 2^40 */

/* ------------------------------------------------------------------ geometry helpers */

/* Source and detector-pixel positions [cm] of (view, channel): source on a circle of radius
 * SID, equiangular fan, detector on an arc of radius SDD about the source. */
static void ray_endpoints(const dexct_fan_geom* g, const double* view_cs, const double* chan_cs, int view,
                          int chan, double* sx, double* sy, double* ex, double* ey) {
  double cb = view_cs[2 * view], sb = view_cs[2 * view + 1];
  double cg = chan_cs[2 * chan], sg = chan_cs[2 * chan + 1];
  *sx = g->sid * cb;
  *sy = g->sid * sb;
  *ex = -(cb * cg - sb * sg);
  *ey = -(sb * cg + cb * sg);
}

/* ------------------------------------------------------------------ classic Siddon 1985 */

static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

/* One ray, one z-slice.  Returns the number of segments; voxel[k] = iy*nx + ix, len[k] in cm. */
int orc_siddon_classic_ray(const dexct_fan_geom* g, const double* view_cs, const double* chan_cs, int view,
                           int chan, int max_seg, int32_t* voxel, double* len) {
  double x1, y1, ex, ey;
  ray_endpoints(g, view_cs, chan_cs, view, chan, &x1, &y1, &ex, &ey);
  double x2 = x1 + g->sdd * ex, y2 = y1 + g->sdd * ey;
  double X0 = -0.5 * g->nx * g->dx, Y0 = -0.5 * g->ny * g->dy;
  double ddx = x2 - x1, ddy = y2 - y1;
  double dconv = sqrt(ddx * ddx + ddy * ddy);
  double amin = 0.0, amax = 1.0;
  if (ddx != 0.0) {
    double a0 = (X0 - x1) / ddx, a1 = (X0 + g->nx * g->dx - x1) / ddx;
    amin = fmax(amin, fmin(a0, a1));
    amax = fmin(amax, fmax(a0, a1));
  } else if (x1 <= X0 || x1 >= X0 + g->nx * g->dx) {
    return 0;
  }
  if (ddy != 0.0) {
    double a0 = (Y0 - y1) / ddy, a1 = (Y0 + g->ny * g->dy - y1) / ddy;
    amin = fmax(amin, fmin(a0, a1));
    amax = fmin(amax, fmax(a0, a1));
  } else if (y1 <= Y0 || y1 >= Y0 + g->ny * g->dy) {
    return 0;
  }
  if (amin >= amax) return 0;
  int cap = g->nx + g->ny + 4;
  double* al = (double*)malloc(sizeof(double) * cap);
  int n = 0;
  al[n++] = amin;
  al[n++] = amax;
  if (ddx != 0.0)
    for (int i = 0; i <= g->nx; ++i) {
      double a = (X0 + i * g->dx - x1) / ddx;
      if (a > amin && a < amax) al[n++] = a;
    }
  if (ddy != 0.0)
    for (int j = 0; j <= g->ny; ++j) {
      double a = (Y0 + j * g->dy - y1) / ddy;
      if (a > amin && a < amax) al[n++] = a;
    }
  qsort(al, n, sizeof(double), cmp_double);
  int nseg = 0;
  for (int k = 1; k < n; ++k) {
    double l = (al[k] - al[k - 1]) * dconv;
    if (l <= 0.0) continue;
    double am = 0.5 * (al[k] + al[k - 1]);
    int ix = (int)floor((x1 + am * ddx - X0) / g->dx);
    int iy = (int)floor((y1 + am * ddy - Y0) / g->dy);
    if (ix < 0 || ix >= g->nx || iy < 0 || iy >= g->ny) continue;
    if (nseg < max_seg) {
      voxel[nseg] = iy * g->nx + ix;
      len[nseg] = l;
    }
    ++nseg;
  }
  free(al);
  return nseg;
}

/* ------------------------------------------------------------------ slab-stepping plan + DDA */

/* Per (view, channel) plan.  Operation order here IS the specification the HIP plan kernel
 * follows; compile with -ffp-contract=off (see Makefile). */
void orc_plan_one(const dexct_fan_geom* g, const double* view_cs, const double* chan_cs, int view, int chan,
                  dexct_ray_plan* p) {
  double sx, sy, ex, ey;
  ray_endpoints(g, view_cs, chan_cs, view, chan, &sx, &sy, &ex, &ey);
  /* continuous voxel coordinates: voxel i spans [i, i+1) */
  double sxv = sx / g->dx + 0.5 * g->nx, syv = sy / g->dy + 0.5 * g->ny;
  double exv = ex / g->dx, eyv = ey / g->dy;
  int axis = fabs(exv) >= fabs(eyv) ? 0 : 1;
  double su, sv_, eu, ev;
  int nu, nv;
  if (axis == 0) { su = sxv; sv_ = syv; eu = exv; ev = eyv; nu = g->nx; nv = g->ny; }
  else           { su = syv; sv_ = sxv; eu = eyv; ev = exv; nu = g->ny; nv = g->nx; }
  double slope = ev / eu;
  double v0 = sv_ - su * slope;
  int64_t SV = (int64_t)llrint(slope * FIX_ONE);
  int64_t V0 = (int64_t)llrint(v0 * FIX_ONE);
  double sq = (double)SV / FIX_ONE;           /* the slope actually stepped */
  double v0q = (double)V0 / FIX_ONE;
  double ulo = 0.0, uhi = (double)nu;
  int miss = 0;
  if (SV > 0) {
    ulo = fmax(ulo, (0.0 - v0q) / sq);
    uhi = fmin(uhi, ((double)nv - v0q) / sq);
  } else if (SV < 0) {
    ulo = fmax(ulo, ((double)nv - v0q) / sq);
    uhi = fmin(uhi, (0.0 - v0q) / sq);
  } else if (v0q < 0.0 || v0q >= (double)nv) {
    miss = 1;
  }
  if (!(uhi > ulo)) miss = 1;
  double inv = 16777216.0;                     /* 2^24 */
  if (SV != 0) inv = fmin(FIX_ONE / fabs((double)SV), 16777216.0);
  p->V0 = V0;
  p->SV = SV;
  p->kf = (float)(inv * (1.0 / 4294967296.0));
  p->len_per_u = (float)(1.0 / fabs(eu));
  p->flags = (uint32_t)axis | (SV > 0 ? 2u : 0u);
  if (miss) {
    p->i_first = 0;
    p->n_slabs = 0;
    p->chord_u = 0.0f;
  } else {
    int i0 = (int)floor(ulo), i1 = (int)ceil(uhi) - 1;
    if (i0 < 0) i0 = 0;
    if (i1 > nu - 1) i1 = nu - 1;
    p->i_first = i0;
    p->n_slabs = i1 - i0 + 1;
    p->chord_u = (float)(uhi - ulo);
  }
}

void orc_plan(const dexct_fan_geom* g, const double* view_cs, const double* chan_cs, int view_begin,
              int view_end, dexct_ray_plan* plan) {
  for (int v = view_begin; v < view_end; ++v)
    for (int c = 0; c < g->n_channels; ++c)
      orc_plan_one(g, view_cs, chan_cs, v, c, &plan[(size_t)(v - view_begin) * g->n_channels + c]);
}

/* One slab of the DDA: the two pieces (ja, la), (jb, lb).  float32 and integer arithmetic only. */
static inline void dda_slab(const dexct_ray_plan* p, int i, int32_t* ja, int32_t* jb, float* la, float* lb) {
  int64_t Va = p->V0 + (int64_t)i * p->SV;
  int64_t Vb = Va + p->SV;
  *ja = (int32_t)(Va >> DEXCT_FIX_FRAC);
  *jb = (int32_t)(Vb >> DEXCT_FIX_FRAC);
  uint32_t fr = (uint32_t)((uint64_t)Va >> 8);
  uint32_t smask = (p->flags & 2u) ? 0xFFFFFFFFu : 0u;
  float d = (float)(fr ^ smask);
  float t = fminf(d * p->kf, 1.0f);
  *la = t;
  *lb = 1.0f - t;
}

/* Trace of one ray in slice z: segments in order of increasing u; voxel = (z*ny + y)*nx + x,
 * len in units of u (float32).  Zero-length pieces and pieces outside the grid are dropped,
 * consecutive pieces in the same voxel are NOT merged (the kernels do not merge either). */
int orc_dda_ray(const dexct_fan_geom* g, const dexct_ray_plan* p, int z, int max_seg, int32_t* voxel,
                float* len) {
  int axis = p->flags & 1u;
  int nv = axis == 0 ? g->ny : g->nx;
  int n = 0;
  for (int s = 0; s < p->n_slabs; ++s) {
    int i = p->i_first + s;
    int32_t j[2];
    float l[2];
    dda_slab(p, i, &j[0], &j[1], &l[0], &l[1]);
    for (int q = 0; q < 2; ++q) {
      if (j[q] < 0 || j[q] >= nv || !(l[q] > 0.0f)) continue;
      int x = axis == 0 ? i : j[q], y = axis == 0 ? j[q] : i;
      if (n < max_seg) {
        voxel[n] = (z * g->ny + y) * g->nx + x;
        len[n] = l[q];
      }
      ++n;
    }
  }
  return n;
}

/* Per-material path lengths [cm], in exactly the arithmetic of the GPU kernels.  A slab contributes
 * t*[ida == m] + (1 - t)*[idb == m] = [idb == m] + t*([ida == m] - [idb == m]), so the kernels keep
 *   count[m]  integer number of slabs whose b-voxel is material m                  (exact, any order)
 *   corr[m]   float32 sum, in slab order, of +t / -t for slabs whose two voxels differ in material
 * and L_m = ((float)count[m] + corr[m]) * len_per_u.  A piece outside the grid counts as id 0;
 * material 0 (and any id >= n_mat) is never accumulated: L_0 comes from the chord. */
void orc_dda_pathlen(const dexct_fan_geom* g, const dexct_ray_plan* p, const uint8_t* vol, int z, int n_mat,
                     float* L) {
  int32_t count[256];
  float corr[256];
  for (int m = 0; m < 256; ++m) { count[m] = 0; corr[m] = 0.0f; }
  int axis = p->flags & 1u;
  int nv = axis == 0 ? g->ny : g->nx;
  for (int s = 0; s < p->n_slabs; ++s) {
    int i = p->i_first + s;
    int32_t j[2];
    float l[2];
    dda_slab(p, i, &j[0], &j[1], &l[0], &l[1]);
    int id[2];
    for (int q = 0; q < 2; ++q) {
      id[q] = 0;
      if (j[q] >= 0 && j[q] < nv) {
        int x = axis == 0 ? i : j[q], y = axis == 0 ? j[q] : i;
        id[q] = vol[((size_t)z * g->ny + y) * g->nx + x];
      }
    }
    if (j[1] >= 0 && j[1] < nv) count[id[1]] += 1;
    else if (j[0] < 0 || j[0] >= nv) continue;          /* slab entirely outside */
    if (id[0] != id[1]) {
      corr[id[0]] += l[0];
      corr[id[1]] -= l[0];
    }
  }
  float acc[256];
  float others = 0.0f;
  for (int m = 1; m < n_mat; ++m) {
    acc[m] = (float)count[m] + corr[m];
    others += acc[m];
  }
  acc[0] = p->chord_u - others;
  for (int m = 0; m < n_mat; ++m) L[m] = acc[m] * p->len_per_u;
}

/* Exact number of Siddon segments (pieces inside the grid with positive length) summed over a plan
 * table: the S_ray of the algorithmic-bytes figure in bench.py. */
long long orc_count_segments(const dexct_fan_geom* g, const dexct_ray_plan* plan, long n_plan, int n_threads) {
  long long total = 0;
#pragma omp parallel for reduction(+ : total) schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
  for (long k = 0; k < n_plan; ++k) {
    const dexct_ray_plan* p = &plan[k];
    int nv = (p->flags & 1u) == 0 ? g->ny : g->nx;
    for (int s = 0; s < p->n_slabs; ++s) {
      int32_t ja, jb;
      float la, lb;
      dda_slab(p, p->i_first + s, &ja, &jb, &la, &lb);
      if (ja >= 0 && ja < nv && la > 0.0f) ++total;
      if (jb >= 0 && jb < nv && lb > 0.0f && (jb != ja || !(la > 0.0f))) ++total;   /* same voxel counts once */
    }
  }
  return total;
}

/* ------------------------------------------------------------------ detection + projection */

/* counts[s] = sum_e weights[s][e] * exp(-sum_m mu[m][e] * L[m]); the weighting mirrors the forward
 * model inside the reference's decomposition (matdecomp.py:146-150). */
static void detect(const double* L, int n_mat, int n_e, int n_spec, const double* mu, const double* w,
                   double* counts) {
  for (int s = 0; s < n_spec; ++s) counts[s] = 0.0;
  for (int e = 0; e < n_e; ++e) {
    double P = 0.0;
    for (int m = 0; m < n_mat; ++m) P += mu[(size_t)m * n_e + e] * L[m];
    double t = exp(-P);
    for (int s = 0; s < n_spec; ++s) counts[s] += w[(size_t)s * n_e + e] * t;
  }
}

/* Full projection with the classic float64 Siddon.  counts[((s*nV + v)*n_rows + r)*n_ch + c];
 * pathlen (optional) [ray][n_mat] in cm.  n_threads > 1 uses OpenMP over (view,row) pairs. */
void orc_project_classic(const dexct_fan_geom* g, const double* view_cs, const double* chan_cs, int view_begin,
                         int view_end, const uint8_t* vol, int n_mat, int n_e, int n_spec, const double* mu,
                         const double* w, double* counts, double* pathlen, int n_threads) {
  int nV = view_end - view_begin;
  int max_seg = g->nx + g->ny + 4;
  long n_vr = (long)nV * g->n_rows;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
  {
    int32_t* vox = (int32_t*)malloc(sizeof(int32_t) * max_seg);
    double* len = (double*)malloc(sizeof(double) * max_seg);
    double L[256], cs[DEXCT_MAX_SPECTRA];
#pragma omp for schedule(dynamic, 1)
    for (long vr = 0; vr < n_vr; ++vr) {
      int v = (int)(vr / g->n_rows), r = (int)(vr % g->n_rows);
      int z = g->z_first + r;
      for (int c = 0; c < g->n_channels; ++c) {
        int ns = orc_siddon_classic_ray(g, view_cs, chan_cs, view_begin + v, c, max_seg, vox, len);
        for (int m = 0; m < n_mat; ++m) L[m] = 0.0;
        for (int k = 0; k < ns; ++k) {
          int id = vol[(size_t)z * g->nx * g->ny + vox[k]];
          if (id < n_mat) L[id] += len[k];
        }
        detect(L, n_mat, n_e, n_spec, mu, w, cs);
        size_t ray = ((size_t)v * g->n_rows + r) * g->n_channels + c;
        for (int s = 0; s < n_spec; ++s)
          counts[(((size_t)s * nV + v) * g->n_rows + r) * g->n_channels + c] = cs[s];
        if (pathlen)
          for (int m = 0; m < n_mat; ++m) pathlen[ray * n_mat + m] = L[m];
      }
    }
    free(vox);
    free(len);
  }
}

/* Full projection with the DDA mirror: float32 path lengths (bit-comparable with the GPU),
 * detection in float64 on those lengths. */
void orc_project_dda(const dexct_fan_geom* g, const double* view_cs, const double* chan_cs, int view_begin,
                     int view_end, const uint8_t* vol, int n_mat, int n_e, int n_spec, const double* mu,
                     const double* w, double* counts, float* pathlen, int n_threads) {
  int nV = view_end - view_begin;
  long n_vc = (long)nV * g->n_channels;
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads > 0 ? n_threads : 1)
  for (long vc = 0; vc < n_vc; ++vc) {
    int v = (int)(vc / g->n_channels), c = (int)(vc % g->n_channels);
    dexct_ray_plan p;
    orc_plan_one(g, view_cs, chan_cs, view_begin + v, c, &p);
    float Lf[256];
    double L[256], cs[DEXCT_MAX_SPECTRA];
    for (int r = 0; r < g->n_rows; ++r) {
      orc_dda_pathlen(g, &p, vol, g->z_first + r, n_mat, Lf);
      for (int m = 0; m < n_mat; ++m) L[m] = (double)Lf[m];
      detect(L, n_mat, n_e, n_spec, mu, w, cs);
      size_t ray = ((size_t)v * g->n_rows + r) * g->n_channels + c;
      for (int s = 0; s < n_spec; ++s)
        counts[(((size_t)s * nV + v) * g->n_rows + r) * g->n_channels + c] = cs[s];
      if (pathlen)
        for (int m = 0; m < n_mat; ++m) pathlen[ray * n_mat + m] = Lf[m];
    }
  }
}


/* ------------------------------------------------------------------ cone beam (3-D rays)
 * SURVEY 8f.4.  Same fan in the (x, y) plane; the source sits at height src_z, detector row r at height
 * row_z[r] (both in cm, z = 0 is the centre of the grid), so a ray climbs linearly in z along its path. */

/* Textbook Siddon 1985 in three dimensions, float64.  voxel = (iz*ny + iy)*nx + ix, len in cm. */
int orc_siddon_classic_ray3d(const dexct_fan_geom* g, const double* view_cs, const double* chan_cs, int view, int chan,
                             double src_z, double det_z, int max_seg, int32_t* voxel, double* len) {
  double x1, y1, ex, ey;
  ray_endpoints(g, view_cs, chan_cs, view, chan, &x1, &y1, &ex, &ey);
  double p1[3] = {x1, y1, src_z};
  double p2[3] = {x1 + g->sdd * ex, y1 + g->sdd * ey, det_z};
  double d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  int n3[3] = {g->nx, g->ny, g->nz};
  double h[3] = {g->dx, g->dy, g->dz};
  double o[3] = {-0.5 * g->nx * g->dx, -0.5 * g->ny * g->dy, -0.5 * g->nz * g->dz};
  double dconv = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  double amin = 0.0, amax = 1.0;
  for (int a = 0; a < 3; ++a) {
    if (d[a] != 0.0) {
      double a0 = (o[a] - p1[a]) / d[a], a1 = (o[a] + n3[a] * h[a] - p1[a]) / d[a];
      amin = fmax(amin, fmin(a0, a1));
      amax = fmin(amax, fmax(a0, a1));
    } else if (p1[a] <= o[a] || p1[a] >= o[a] + n3[a] * h[a]) {
      return 0;
    }
  }
  if (amin >= amax) return 0;
  int cap = g->nx + g->ny + g->nz + 6;
  double* al = (double*)malloc(sizeof(double) * cap);
  int n = 0;
  al[n++] = amin;
  al[n++] = amax;
  for (int a = 0; a < 3; ++a)
    if (d[a] != 0.0)
      for (int i = 0; i <= n3[a]; ++i) {
        double q = (o[a] + i * h[a] - p1[a]) / d[a];
        if (q > amin && q < amax) al[n++] = q;
      }
  qsort(al, n, sizeof(double), cmp_double);
  int nseg = 0;
  for (int k = 1; k < n; ++k) {
    double l = (al[k] - al[k - 1]) * dconv;
    if (l <= 0.0) continue;
    double am = 0.5 * (al[k] + al[k - 1]);
    int idx[3], ok = 1;
    for (int a = 0; a < 3; ++a) {
      idx[a] = (int)floor((p1[a] + am * d[a] - o[a]) / h[a]);
      if (idx[a] < 0 || idx[a] >= n3[a]) ok = 0;
    }
    if (!ok) continue;
    if (nseg < max_seg) {
      voxel[nseg] = (idx[2] * g->ny + idx[1]) * g->nx + idx[0];
      len[nseg] = l;
    }
    ++nseg;
  }
  free(al);
  return nseg;
}

/* z part of the plan of one (view, channel, row): fixed-point w(u) = (W0 + i*SW)/2^40 at the entry face of
 * slab i, crossing factor, and the 3-D path length per unit u.  Mirrors cone_row_plan in siddon_cone.hip. */
typedef struct { int64_t W0, SW; float kfw, len3d; uint32_t wpos; } orc_cone_row;

static void cone_row_plan(const dexct_fan_geom* g, const double* view_cs, const double* chan_cs, int view, int chan,
                          const dexct_ray_plan* p, double src_z, double det_z, orc_cone_row* c) {
  double sx, sy, ex, ey;
  ray_endpoints(g, view_cs, chan_cs, view, chan, &sx, &sy, &ex, &ey);
  int axis = p->flags & 1u;
  double su = axis == 0 ? sx / g->dx + 0.5 * g->nx : sy / g->dy + 0.5 * g->ny;
  double eu = axis == 0 ? ex / g->dx : ey / g->dy;
  double ws = src_z / g->dz + 0.5 * g->nz;
  double sw = ((det_z - src_z) / g->dz) / (g->sdd * eu);
  double w0 = ws - su * sw;
  c->SW = (int64_t)llrint(sw * FIX_ONE);
  c->W0 = (int64_t)llrint(w0 * FIX_ONE);
  double inv = 16777216.0;
  if (c->SW != 0) inv = fmin(FIX_ONE / fabs((double)c->SW), 16777216.0);
  c->kfw = (float)(inv * (1.0 / 4294967296.0));
  double tz = (det_z - src_z) / g->sdd;
  c->len3d = (float)((1.0 / fabs(eu)) * sqrt(1.0 + tz * tz));
  c->wpos = c->SW > 0 ? 0xFFFFFFFFu : 0u;
}

/* Per-material path lengths [cm] of one cone-beam ray, float32, in the kernel's arithmetic: per slab the
 * pieces are cut at the v-crossing tv and the z-crossing tw (each at most one per slab); their sum is taken as an
 * integer count of the b voxel plus float32 corrections where the slab straddles a material boundary (same
 * identity as the 2-D form).  All materials, material 0 included, are accumulated. */
void orc_cone_pathlen(const dexct_fan_geom* g, const double* view_cs, const double* chan_cs, int view, int chan,
                      const dexct_ray_plan* p, double src_z, double det_z, const uint8_t* vol, int n_mat, float* L) {
  orc_cone_row c;
  cone_row_plan(g, view_cs, chan_cs, view, chan, p, src_z, det_z, &c);
  int32_t cnt[257];                                                  /* slot 256: outside the grid */
  float corr[257];
  for (int m = 0; m < 257; ++m) { cnt[m] = 0; corr[m] = 0.0f; }
  int axis = p->flags & 1u;
  int nv = axis == 0 ? g->ny : g->nx;
  for (int s = 0; s < p->n_slabs; ++s) {
    int i = p->i_first + s;
    int32_t ja, jb;
    float tv, dummy;
    dda_slab(p, i, &ja, &jb, &tv, &dummy);
    int64_t Wa = c.W0 + (int64_t)i * c.SW, Wb = Wa + c.SW;
    int32_t ka = (int32_t)(Wa >> DEXCT_FIX_FRAC), kb = (int32_t)(Wb >> DEXCT_FIX_FRAC);
    uint32_t fr = (uint32_t)((uint64_t)Wa >> 8);
    float tw = fminf((float)(fr ^ c.wpos) * c.kfw, 1.0f);
    float t1 = fminf(tv, tw), t2 = fmaxf(tv, tw);
    int32_t jm = tv <= tw ? jb : ja, km = tv <= tw ? ka : kb;       /* the middle piece */
    int32_t jj[3] = {ja, jm, jb}, kk[3] = {ka, km, kb};
    int id[3];
    for (int q = 0; q < 3; ++q) {
      id[q] = 256;                                                  /* outside the grid: no material (every uint8 is an id) */
      if (jj[q] < 0 || jj[q] >= nv || kk[q] < 0 || kk[q] >= g->nz) continue;
      int x = axis == 0 ? i : jj[q], y = axis == 0 ? jj[q] : i;
      id[q] = vol[((size_t)kk[q] * g->ny + y) * g->nx + x];
    }
    /* t1 [ida] + (t2 - t1) [idm] + (1 - t2) [idb] = [idb] + t2 ([idm] - [idb]) + t1 ([ida] - [idm]) */
    cnt[id[2]] += 1;
    if (id[0] != id[1] || id[1] != id[2]) {
      corr[id[1]] += t2;
      corr[id[2]] -= t2;
      corr[id[0]] += t1;
      corr[id[1]] -= t1;
    }
  }
  for (int m = 0; m < n_mat; ++m) L[m] = ((float)cnt[m] + corr[m]) * c.len3d;
}

/* Cone-beam projections: classic float64 (pathlen + counts) and the DDA mirror (float32 pathlen, float64
 * detection on them).  row_z[r] detector heights [cm].  counts[((s*nV + v)*n_rows + r)*n_ch + c]. */
void orc_project_cone(const dexct_fan_geom* g, const double* view_cs, const double* chan_cs, int view_begin,
                      int view_end, const double* row_z, double src_z, const uint8_t* vol, int n_mat, int n_e,
                      int n_spec, const double* mu, const double* w, double* counts, double* pathlen_classic,
                      float* pathlen_dda, int use_dda, int n_threads) {
  int nV = view_end - view_begin;
  int max_seg = g->nx + g->ny + g->nz + 6;
  long n_vr = (long)nV * g->n_rows;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
  {
    int32_t* vox = (int32_t*)malloc(sizeof(int32_t) * max_seg);
    double* len = (double*)malloc(sizeof(double) * max_seg);
    double L[256], cs[DEXCT_MAX_SPECTRA];
    float Lf[256];
#pragma omp for schedule(dynamic, 1)
    for (long vr = 0; vr < n_vr; ++vr) {
      int v = (int)(vr / g->n_rows), r = (int)(vr % g->n_rows);
      for (int c = 0; c < g->n_channels; ++c) {
        size_t ray = ((size_t)v * g->n_rows + r) * g->n_channels + c;
        if (use_dda) {
          dexct_ray_plan p;
          orc_plan_one(g, view_cs, chan_cs, view_begin + v, c, &p);
          orc_cone_pathlen(g, view_cs, chan_cs, view_begin + v, c, &p, src_z, row_z[r], vol, n_mat, Lf);
          for (int m = 0; m < n_mat; ++m) L[m] = (double)Lf[m];
          if (pathlen_dda)
            for (int m = 0; m < n_mat; ++m) pathlen_dda[ray * n_mat + m] = Lf[m];
        } else {
          int ns = orc_siddon_classic_ray3d(g, view_cs, chan_cs, view_begin + v, c, src_z, row_z[r], max_seg, vox, len);
          for (int m = 0; m < n_mat; ++m) L[m] = 0.0;
          for (int k = 0; k < ns; ++k) {
            int id = vol[vox[k]];
            if (id < n_mat) L[id] += len[k];
          }
          if (pathlen_classic)
            for (int m = 0; m < n_mat; ++m) pathlen_classic[ray * n_mat + m] = L[m];
        }
        detect(L, n_mat, n_e, n_spec, mu, w, cs);
        for (int s = 0; s < n_spec; ++s)
          counts[(((size_t)s * nV + v) * g->n_rows + r) * g->n_channels + c] = cs[s];
      }
    }
    free(vox);
    free(len);
  }
}

/* ------------------------------------------------------------------ Gauss-Newton (float64) */

/* Restates matdecomp.py:87-127 per pixel.  i0[k][e], mus[m][e] channel-independent.
 * out_a[2p+m].  Same fixed iteration count, init 1e-6 (:98-99), exponent clip +-700 (:116),
 * full Newton step with the (g/nu - 1) * hessian term (:123), no damping. */
void orc_gn_decompose(const double* g1, const double* g2, long n_pix, const double* i0, const double* mus, int n_e,
                      int n_iters, double* out_a, int n_threads) {
  /* product tables with the reference's rounding (ssff, ssff2: matdecomp.py:102,105) */
  double* tab = (double*)malloc(sizeof(double) * 10 * n_e);
  for (int k = 0; k < 2; ++k)
    for (int e = 0; e < n_e; ++e) {
      double w = i0[k * n_e + e], m0 = mus[e], m1 = mus[n_e + e];
      double* t = tab + ((size_t)k * 5) * n_e;
      t[0 * n_e + e] = w * m0;
      t[1 * n_e + e] = w * m1;
      t[2 * n_e + e] = w * (m0 * m0);
      t[3 * n_e + e] = w * (m0 * m1);
      t[4 * n_e + e] = w * (m1 * m1);
    }
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
  for (long p = 0; p < n_pix; ++p) {
    double a0 = 1e-6, a1 = 1e-6;
    double g[2] = {g1[p], g2[p]};
    for (int it = 0; it < n_iters; ++it) {
      double nu[2] = {0, 0}, G0[2] = {0, 0}, G1[2] = {0, 0}, H00[2] = {0, 0}, H01[2] = {0, 0}, H11[2] = {0, 0};
      for (int e = 0; e < n_e; ++e) {
        double x = -(a0 * mus[e] + a1 * mus[n_e + e]);
        x = x < -700.0 ? -700.0 : (x > 700.0 ? 700.0 : x);
        double t = exp(x);
        for (int k = 0; k < 2; ++k) {
          const double* tk = tab + ((size_t)k * 5) * n_e;
          nu[k] += i0[k * n_e + e] * t;
          G0[k] += tk[0 * n_e + e] * t;
          G1[k] += tk[1 * n_e + e] * t;
          H00[k] += tk[2 * n_e + e] * t;
          H01[k] += tk[3 * n_e + e] * t;
          H11[k] += tk[4 * n_e + e] * t;
        }
      }
      double dF0 = 0, dF1 = 0, h00 = 0, h01 = 0, h11 = 0;
      for (int k = 0; k < 2; ++k) {
        double c = g[k] / nu[k] - 1.0, q = g[k] / (nu[k] * nu[k]);
        dF0 += c * G0[k];
        dF1 += c * G1[k];
        h00 += q * (G0[k] * G0[k]) - c * H00[k];
        h01 += q * (G0[k] * G1[k]) - c * H01[k];
        h11 += q * (G1[k] * G1[k]) - c * H11[k];
      }
      double det = h00 * h11 - h01 * h01;
      a0 -= (h11 * dF0 - h01 * dF1) / det;
      a1 -= (h00 * dF1 - h01 * dF0) / det;
    }
    out_a[2 * p] = a0;
    out_a[2 * p + 1] = a1;
  }
  free(tab);
}

int orc_max_threads(void) {
#ifdef _OPENMP
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
#else
  return 1;
#endif
}
