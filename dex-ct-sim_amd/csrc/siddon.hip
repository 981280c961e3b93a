// Siddon forward projection + polychromatic detection for gfx950.
//
// Replaces get_sino (reference call site main.py:120; algorithm: Siddon 1985 as named in
// README.md:27-28,41).  The exact radiological path is computed in a slab-stepping form:
// along the dominant in-plane axis u a ray crosses one slab per step and at most one plane of
// the minor axis v inside it, so every slab contributes two pieces: voxel (i, ja) with length t
// and voxel (i, jb) with length 1 - t, in units of u.  v is a 40-bit fixed-point number stepped
// by integer addition, so voxel indices are exact; t is one float32 multiply on the top 32
// fraction bits.
//
// Path lengths are accumulated per MATERIAL (energy independent).  A slab's contribution to
// material m, t*[ida == m] + (1 - t)*[idb == m], is rewritten as
//        [idb == m]  +  t * ([ida == m] - [idb == m])
// i.e. an INTEGER count of slabs (exact, order independent) plus a float32 correction that is
// non-zero only where the v-plane crossing inside the slab separates two different materials.
// L_m = ((float)count_m + corr_m) * len_per_u; material 0 comes from the chord.  The oracle
// (oracle/dexct_oracle.c: orc_dda_pathlen) performs the same operations, so per-material path
// lengths are compared bit for bit.  One pass over the energy bins then applies the attenuation
// table and the detector weighting for every spectrum of the call (weights = I0 * eta * [E] * dE,
// the forward model of matdecomp.py:146-150).
//
// Traversal kernels:
//   rays_kernel   one thread per ray, lanes over adjacent channels (coalesced on the layout whose
//                 minor axis is contiguous); any number of rows; the kernel for 2-D scans.
//   rows_kernel   one workgroup per (view, channel): all rows of a stacked fan share the in-plane
//                 traversal, so slab records are computed once per workgroup into LDS and every lane
//                 (= detector row = z-slice) reads its voxel byte from the z-fastest layout.
//   rows4_kernel  the same with 4 rows per lane: one dword load serves 4 rows and the integer counts
//                 are kept as packed bytes (SWAR), ~1.3 VALU instructions per row and slab.  Slabs
//                 without a crossing ("full") and with one are split into two LDS lists by a
//                 workgroup prefix sum (legal because counts are order independent and corrections
//                 only arise in crossing slabs, whose relative order is kept).  <= 4 materials.
// Tables are wave-uniform and are read through the scalar cache (s_load), not LDS.
#include <cstdlib>

#include "common.h"
#include "siddon_detect.h"

namespace dexct {

// Any number of materials: per-material lengths in LDS column `tid` (stride `stride`).
__device__ __forceinline__ void detect_store_lds(const float* lds_L, int tid, int stride, const ProjArgs& a,
                                                 const float* __restrict__ mu, const float* __restrict__ w,
                                                 const float* __restrict__ w2, size_t ray) {
  const int n_e = a.n_energies, n_mat = a.n_materials;
  const size_t sstride = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  if (a.pathlen)
    for (int m = 0; m < n_mat; ++m) a.pathlen[ray * n_mat + m] = lds_L[m * stride + tid];
  float acc[DEXCT_MAX_SPECTRA], var[DEXCT_MAX_SPECTRA];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) acc[s] = var[s] = 0.0f;
  for (int e = 0; e < n_e; ++e) {
    float p = 0.0f;
    for (int m = 0; m < n_mat; ++m) p = fmaf(mu[m * n_e + e], lds_L[m * stride + tid], p);
    const float t = __builtin_amdgcn_exp2f(-p * kLog2e);
#pragma unroll
    for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
      if (s < a.n_spectra) {
        acc[s] = fmaf(w[s * n_e + e], t, acc[s]);
        if (a.variance) var[s] = fmaf(w2[s * n_e + e], t, var[s]);
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

// Register accumulators of one ray for materials 1..NM-1.
template <int NM>
struct RegAcc {
  int32_t cnt[NM > 1 ? NM : 2];
  float corr[NM > 1 ? NM : 2];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int m = 0; m < (NM > 1 ? NM : 2); ++m) { cnt[m] = 0; corr[m] = 0.0f; }
  }
  __device__ __forceinline__ void slab(uint32_t ida, uint32_t idb, float t) {
#pragma unroll
    for (int m = 1; m < NM; ++m) cnt[m] += (idb == (uint32_t)m) ? 1 : 0;
    if (ida != idb) {
#pragma unroll
      for (int m = 1; m < NM; ++m) {
        corr[m] += (ida == (uint32_t)m) ? t : 0.0f;
        corr[m] -= (idb == (uint32_t)m) ? t : 0.0f;
      }
    }
  }
  // the same without a branch (adding +0.0f leaves a sum unchanged): for lockstep lanes
  __device__ __forceinline__ void slab_flat(uint32_t ida, uint32_t idb, float t) {
    const float td = ida != idb ? t : 0.0f;
#pragma unroll
    for (int m = 1; m < NM; ++m) {
      cnt[m] += (idb == (uint32_t)m) ? 1 : 0;
      corr[m] += (ida == (uint32_t)m) ? td : 0.0f;
      corr[m] -= (idb == (uint32_t)m) ? td : 0.0f;
    }
  }
  __device__ __forceinline__ void lengths(const dexct_ray_plan& p, float (&L)[NM]) const {
    float others = 0.0f;
#pragma unroll
    for (int m = 1; m < NM; ++m) {
      L[m] = (float)cnt[m] + corr[m];
      others += L[m];
    }
    L[0] = (p.chord_u - others) * p.len_per_u;
#pragma unroll
    for (int m = 1; m < NM; ++m) L[m] *= p.len_per_u;
  }
};

// LDS accumulators (any number of materials): cnt and corr columns of width `stride`, slot 0 is a sink.
struct LdsAcc {
  float* cnt;   // exact integers held as float32 (< 2^24 slabs)
  float* corr;
  int stride, tid, n_mat;
  __device__ __forceinline__ void clear() {
    for (int m = 0; m < n_mat; ++m) { cnt[m * stride + tid] = 0.0f; corr[m * stride + tid] = 0.0f; }
  }
  __device__ __forceinline__ void slab(uint32_t ida, uint32_t idb, float t) {
    const uint32_t sa = ida < (uint32_t)n_mat ? ida : 0u, sb = idb < (uint32_t)n_mat ? idb : 0u;
    __hip_atomic_fetch_add(&cnt[sb * stride + tid], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (sa != sb) {
      __hip_atomic_fetch_add(&corr[sa * stride + tid], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&corr[sb * stride + tid], -t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  // leaves the lengths [cm] in cnt[m]
  __device__ __forceinline__ void lengths(const dexct_ray_plan& p) {
    float others = 0.0f;
    for (int m = 1; m < n_mat; ++m) {
      const float l = cnt[m * stride + tid] + corr[m * stride + tid];
      cnt[m * stride + tid] = l;
      others += l;
    }
    cnt[tid] = (p.chord_u - others) * p.len_per_u;
    for (int m = 1; m < n_mat; ++m) cnt[m * stride + tid] *= p.len_per_u;
  }
};

constexpr int kLdsBlock = 128;   // block size of the LDS-accumulator instantiations (2 x M x 128 x 4 B)

// ---------------------------------------------------------------------------------------------
// rays_kernel: one thread per ray.  NM > 0: materials in registers; NM == 0: LDS accumulators.
template <int NM, int BLOCK, int BATCH = 1>
__global__ __launch_bounds__(BLOCK) void rays_kernel(ProjArgs a, const float* __restrict__ mu,
                                                      const float* __restrict__ w, const float* __restrict__ w2) {
  extern __shared__ float lds_dyn[];
  const int tid = threadIdx.x;
  const int c = blockIdx.x * BLOCK + tid;
  const int r = blockIdx.y, v = blockIdx.z;
  const bool live = c < a.g.n_channels;
  dexct_ray_plan p;
  if (live) p = a.plan[(size_t)v * a.g.n_channels + c];
  else { p.n_slabs = 0; p.V0 = 0; p.SV = 0; p.i_first = 0; p.kf = 0; p.len_per_u = 0; p.chord_u = 0; p.flags = 0; }
  const int axis = p.flags & 1u;
  const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
  const int nv = axis == 0 ? a.g.ny : a.g.nx;
  const uint8_t* __restrict__ base = (axis == 0 ? a.vol_xy : a.vol_yx) + (size_t)(a.g.z_first + r) * a.g.nx * a.g.ny;
  RegAcc<(NM > 0 ? NM : 1)> ra;
  LdsAcc la{lds_dyn, lds_dyn + a.n_materials * BLOCK, BLOCK, tid, a.n_materials};
  if (NM > 0) ra.clear(); else la.clear();
  long long V = p.V0 + (long long)p.i_first * p.SV;
  uint32_t off = (uint32_t)p.i_first * (uint32_t)nv;
  if constexpr (NM > 0 && BATCH > 1) {
    // BATCH slabs at a time: all 2 x BATCH byte loads are issued before the first id is used (a lane's slabs form a
    // serial chain otherwise: one L2 round trip per slab).  The slab geometry is integer arithmetic on V, independent of
    // the loaded ids, so it runs ahead; the accumulation keeps slab order (corrections are order dependent) and uses the
    // branch-free form (adding +0 changes no sum).  A piece outside the grid - or past the ray's last slab - reads byte 0
    // of the slice and is then given id 0, which counts nothing.
    for (int s0 = 0; s0 < p.n_slabs; s0 += BATCH) {
      uint32_t xa[BATCH], xb[BATCH];
      uint32_t in_mask = 0;
      const long long Vb = V;
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        const bool on = s0 + k < p.n_slabs;
        const int32_t ja = (int32_t)(V >> DEXCT_FIX_FRAC), jb = (int32_t)((V + p.SV) >> DEXCT_FIX_FRAC);
        const bool ina = on && (uint32_t)ja < (uint32_t)nv, inb = on && (uint32_t)jb < (uint32_t)nv;
        xa[k] = base[ina ? off + (uint32_t)ja : 0u];
        xb[k] = base[inb ? off + (uint32_t)jb : 0u];
        in_mask |= (ina ? 1u : 0u) << (2 * k) | (inb ? 2u : 0u) << (2 * k);
        V += p.SV;
        off += (uint32_t)nv;
      }
      __builtin_amdgcn_sched_barrier(0);
      // counts: integer, order independent, no branch.  Corrections (float32, slab order) only where a crossing separates
      // two different ids - rare (material boundaries), so the crossing parameter t is not even computed otherwise.
      uint32_t differ = 0;
#pragma unroll
      for (int k = 0; k < BATCH; ++k) {
        xa[k] = (in_mask >> (2 * k)) & 1u ? xa[k] : 0u;
        xb[k] = (in_mask >> (2 * k + 1)) & 1u ? xb[k] : 0u;
#pragma unroll
        for (int m = 1; m < NM; ++m) ra.cnt[m] += (xb[k] == (uint32_t)m) ? 1 : 0;
        differ |= xa[k] ^ xb[k];
      }
      if (__ballot(differ != 0u) != 0ull) {
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          if (xa[k] != xb[k]) {
            const float t = dda_slab(Vb + (long long)k * p.SV, p.SV, smask, p.kf).t;
#pragma unroll
            for (int m = 1; m < NM; ++m) {
              ra.corr[m] += (xa[k] == (uint32_t)m) ? t : 0.0f;
              ra.corr[m] -= (xb[k] == (uint32_t)m) ? t : 0.0f;
            }
          }
        }
      }
    }
  } else {
    for (int s = 0; s < p.n_slabs; ++s) {
      const SlabPieces sp = dda_slab(V, p.SV, smask, p.kf);
      const bool ina = (uint32_t)sp.ja < (uint32_t)nv, inb = (uint32_t)sp.jb < (uint32_t)nv;
      const uint32_t ida = ina ? base[off + (uint32_t)sp.ja] : 0u;
      const uint32_t idb = inb ? base[off + (uint32_t)sp.jb] : 0u;
      if (NM > 0) ra.slab(ida, idb, sp.t); else la.slab(ida, idb, sp.t);
      V += p.SV;
      off += (uint32_t)nv;
    }
  }
  if (!live) return;
  const size_t ray = ray_index(a, v, r, c);
  if (NM > 0) {
    float L[NM > 0 ? NM : 1];
    ra.lengths(p, L);
    detect_store1<(NM > 0 ? NM : 1)>(L, a, mu, w, w2, ray);
  } else {
    la.lengths(p);
    detect_store_lds(la.cnt, tid, BLOCK, a, mu, w, w2, ray);
  }
}

// ---------------------------------------------------------------------------------------------
// wave_ray_kernel: ONE WAVEFRONT PER RAY - the mapping BASELINE.json's north star names.  The 64 lanes own
// consecutive dominant-axis slabs (i = i_first + 64 k + lane) and read their voxels from the layout whose DOMINANT
// axis is contiguous (x-dominant rays: vol_yx, y-dominant: vol_xy), so a wave's load is one contiguous run per
// minor-axis row the ray crosses in those 64 slabs ("coalesced reads of phantom slabs along the dominant
// voxel-stepping axis").  Per-lane integer counts are reduced with wavefront shuffles; the float32 corrections -
// which only arise where a crossing separates two materials - are applied by the whole wave in slab order
// (ballot + readlane), so the per-material path lengths stay bit-identical to every other kernel and to the oracle.
// Detection: tables staged in LDS once per workgroup, lanes over energy bins, shuffle reduction of the sums.
// A workgroup is persistent (grid-stride over rays) so that the table staging is paid once, not per ray.
// It exists for rays that share nothing with their neighbours (single-row scans) and as the measured A/B of the
// mapping argument in DESIGN.md section 4.1; <= 4 materials, <= 2 spectra ... DEXCT_MAX_SPECTRA.
template <int NM>
__global__ __launch_bounds__(256) void wave_ray_kernel(ProjArgs a, const float* __restrict__ mu,
                                                        const float* __restrict__ w, long long n_rays_total) {
  extern __shared__ float lds_tab[];             // mu [NM][nE] (scaled by log2 e), then w [S][nE]
  const int n_e = a.n_energies, n_s = a.n_spectra;
  for (int k = threadIdx.x; k < NM * n_e; k += 256) lds_tab[k] = mu[k] * kLog2e;
  for (int k = threadIdx.x; k < n_s * n_e; k += 256) lds_tab[NM * n_e + k] = w[k];
  __syncthreads();
  const float* lmu = lds_tab;
  const float* lw = lds_tab + NM * n_e;
  const int lane = threadIdx.x & 63;
  const long long n_waves = (long long)gridDim.x * 4;
  const size_t sstride = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  for (long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); q < n_rays_total; q += n_waves) {
    const int c = (int)(q % a.g.n_channels);
    const long long vr = q / a.g.n_channels;
    const int r = (int)(vr % a.g.n_rows), v = (int)(vr / a.g.n_rows);
    const dexct_ray_plan p = a.plan[(size_t)v * a.g.n_channels + c];       // wave-uniform
    const int axis = p.flags & 1u;
    const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
    const int nv = axis == 0 ? a.g.ny : a.g.nx;
    const int nu = axis == 0 ? a.g.nx : a.g.ny;
    // dominant axis contiguous: x-dominant rays read [z][y][x], y-dominant rays [z][x][y]
    const uint8_t* __restrict__ base = (axis == 0 ? a.vol_yx : a.vol_xy) + (size_t)(a.g.z_first + r) * a.g.nx * a.g.ny;
    int32_t cnt[NM > 1 ? NM : 2];
    float corr[NM > 1 ? NM : 2];          // wave-uniform
#pragma unroll
    for (int m = 0; m < (NM > 1 ? NM : 2); ++m) { cnt[m] = 0; corr[m] = 0.0f; }
    for (int s0 = 0; s0 < p.n_slabs; s0 += 64) {
      const int s = s0 + lane;
      const bool live = s < p.n_slabs;
      const int i = p.i_first + s;
      const SlabPieces sp = dda_slab(p.V0 + (long long)i * p.SV, p.SV, smask, p.kf);
      const bool ina = live && (uint32_t)sp.ja < (uint32_t)nv, inb = live && (uint32_t)sp.jb < (uint32_t)nv;
      const uint32_t ida = ina ? base[(uint32_t)sp.ja * (uint32_t)nu + (uint32_t)i] : 0u;
      const uint32_t idb = inb ? base[(uint32_t)sp.jb * (uint32_t)nu + (uint32_t)i] : 0u;
#pragma unroll
      for (int m = 1; m < NM; ++m) cnt[m] += (idb == (uint32_t)m) ? 1 : 0;
      // corrections in slab order = lane order: the wave walks the lanes that have one
      unsigned long long ev = __ballot(live && ida != idb);
      while (ev) {
        const int l = __builtin_ctzll(ev);
        ev &= ev - 1ull;
        const float t = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(sp.t), l));
        const uint32_t ia = __builtin_amdgcn_readlane(ida, l), ib = __builtin_amdgcn_readlane(idb, l);
#pragma unroll
        for (int m = 1; m < NM; ++m) {
          corr[m] += (ia == (uint32_t)m) ? t : 0.0f;
          corr[m] -= (ib == (uint32_t)m) ? t : 0.0f;
        }
      }
    }
#pragma unroll
    for (int m = 1; m < NM; ++m)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) cnt[m] += __shfl_xor(cnt[m], o, 64);
    float L[NM];
    float others = 0.0f;
#pragma unroll
    for (int m = 1; m < NM; ++m) {
      L[m] = (float)cnt[m] + corr[m];
      others += L[m];
    }
    L[0] = (p.chord_u - others) * p.len_per_u;
#pragma unroll
    for (int m = 1; m < NM; ++m) L[m] *= p.len_per_u;
    const size_t ray = ray_index(a, v, r, c);
    if (a.pathlen) {
#pragma unroll
      for (int m = 0; m < NM; ++m)
        if (lane == m) a.pathlen[ray * NM + m] = L[m];
    }
    // detection: lanes over the energy bins
    float acc[DEXCT_MAX_SPECTRA];
#pragma unroll
    for (int sI = 0; sI < DEXCT_MAX_SPECTRA; ++sI) acc[sI] = 0.0f;
    for (int e = lane; e < n_e; e += 64) {
      float pe = 0.0f;
#pragma unroll
      for (int m = 0; m < NM; ++m) pe = fmaf(lmu[m * n_e + e], L[m], pe);
      const float t = __builtin_amdgcn_exp2f(-pe);
#pragma unroll
      for (int sI = 0; sI < DEXCT_MAX_SPECTRA; ++sI)
        if (sI < n_s) acc[sI] = fmaf(lw[sI * n_e + e], t, acc[sI]);
    }
#pragma unroll
    for (int sI = 0; sI < DEXCT_MAX_SPECTRA; ++sI)
      if (sI < n_s) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[sI] += __shfl_xor(acc[sI], o, 64);
        if (lane == 0) {
          a.counts[ray + sI * sstride] = acc[sI];
          if (a.sino_log) a.sino_log[ray + sI * sstride] = log_ratio(a.air[sI], acc[sI]);
        }
      }
  }
}

template <int NM>
static int launch_wave_ray(const ProjArgs& a, const Tables& t, hipStream_t st) {
  const long long n_rays = (long long)a.n_local_views * a.g.n_rows * a.g.n_channels;
  const size_t lds = (size_t)(NM + a.n_spectra) * a.n_energies * sizeof(float);
  if (lds > 64 * 1024) return DEXCT_ERANGE;
  long long nblk = (n_rays + 3) / 4;
  const long long cap = 256ll * 8 * 4;                  // persistent: 8 workgroups per CU, each strides over the rays
  if (nblk > cap) nblk = cap;
  hipLaunchKernelGGL((wave_ray_kernel<NM>), dim3((unsigned)nblk), dim3(256), lds, st, a, t.mu, t.w, n_rays);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

// ---------------------------------------------------------------------------------------------
// rows_kernel: one workgroup per (view, channel, chunk of BLOCK rows), one row per lane.
struct SlabRec {
  uint32_t offa, offb;  // byte offsets of the (x, y) columns in the z-fastest layout (0 where outside the grid)
  float t;
  uint32_t masks;       // bits 0-7: 0xFF if piece a lies inside the grid, bits 8-15: the same for piece b
};

// Block -> (view, channel, row chunk).  Two levels of locality:
//  * XCD: hardware deals consecutive block ids round-robin over the 8 XCDs, so a contiguous range of
//    LOGICAL ids is given to each XCD (its L2 then sees neighbouring rays);
//  * tile: logical ids walk groups of kViewTile consecutive views with the view index fastest, then the
//    channel.  The ~400 workgroups an XCD runs at once are then (kViewTile views x ~50 channels): rays of
//    neighbouring views through the same voxel columns are in flight together and hit in L2 instead of
//    going to the Infinity Cache / HBM once per view.
constexpr int kViewTileDefault = 8;

struct BlockRay { int v, c, chunk; };

__device__ __forceinline__ BlockRay block_to_ray(int n_views, int n_channels, int n_chunks, int kViewTile) {
  const uint32_t nblk = gridDim.x, b = blockIdx.x, per = nblk >> 3;
  const uint32_t logical = (b < (per << 3)) ? (b & 7u) * per + (b >> 3) : b;
  const uint32_t group_size = (uint32_t)kViewTile * n_channels * n_chunks;
  const uint32_t g = logical / group_size, rem = logical - g * group_size;
  const uint32_t views_here = min((uint32_t)kViewTile, (uint32_t)n_views - g * kViewTile);
  BlockRay r;
  r.chunk = rem % n_chunks;
  const uint32_t q = rem / n_chunks;
  r.v = g * kViewTile + q % views_here;
  r.c = q / views_here;
  return r;
}

template <int NM, int BLOCK>
__global__ __launch_bounds__(BLOCK) void rows_kernel(ProjArgs a, const float* __restrict__ mu,
                                                      const float* __restrict__ w, const float* __restrict__ w2, int n_chunks) {
  __shared__ SlabRec rec[BLOCK];
  extern __shared__ float lds_dyn[];
  const int tid = threadIdx.x;
  const BlockRay br = block_to_ray(a.n_local_views, a.g.n_channels, n_chunks, a.view_tile);
  const int chunk = br.chunk, c = br.c, v = br.v;
  const int r = chunk * BLOCK + tid;
  const bool live = r < a.g.n_rows;
  const int z = a.g.z_first + (live ? r : 0);
  const dexct_ray_plan p = a.plan[(size_t)v * a.g.n_channels + c];   // uniform
  const int axis = p.flags & 1u;
  const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
  const int nv = axis == 0 ? a.g.ny : a.g.nx;
  // offset of column (i, j): axis 0: (j*nx + i)*nz, axis 1: (i*nx + j)*nz
  const uint32_t su = (axis == 0 ? 1u : (uint32_t)a.g.nx) * (uint32_t)a.g.nz;
  const uint32_t sv = (axis == 0 ? (uint32_t)a.g.nx : 1u) * (uint32_t)a.g.nz;
  const uint8_t* __restrict__ col = a.vol_zf + z;
  RegAcc<(NM > 0 ? NM : 1)> ra;
  LdsAcc la{lds_dyn, lds_dyn + a.n_materials * BLOCK, BLOCK, tid, a.n_materials};
  if (NM > 0) ra.clear(); else la.clear();
  for (int s0 = 0; s0 < p.n_slabs; s0 += BLOCK) {
    const int n_here = min(BLOCK, p.n_slabs - s0);
    if (tid < n_here) {
      const int i = p.i_first + s0 + tid;
      const SlabPieces sp = dda_slab(p.V0 + (long long)i * p.SV, p.SV, smask, p.kf);
      const bool ina = (uint32_t)sp.ja < (uint32_t)nv, inb = (uint32_t)sp.jb < (uint32_t)nv;
      SlabRec q;
      q.offa = ina ? (uint32_t)i * su + (uint32_t)sp.ja * sv : 0u;
      q.offb = inb ? (uint32_t)i * su + (uint32_t)sp.jb * sv : 0u;
      q.t = sp.t;
      q.masks = (ina ? 0xFFu : 0u) | (inb ? 0xFF00u : 0u);
      rec[tid] = q;
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < n_here; ++s) {
      const SlabRec q = rec[s];
      const uint32_t ida = col[q.offa] & q.masks;            // outside the grid -> id 0, no branch
      const uint32_t idb = col[q.offb] & (q.masks >> 8);
      if (NM > 0) ra.slab_flat(ida, idb, q.t); else la.slab(ida, idb, q.t);   // no branch: keeps the 8 loads of the unrolled body in flight
    }
    __syncthreads();
  }
  if (!live) return;
  const size_t ray = ray_index(a, v, r, c);
  if (NM > 0) {
    float L[NM > 0 ? NM : 1];
    ra.lengths(p, L);
    detect_store1<(NM > 0 ? NM : 1)>(L, a, mu, w, w2, ray);
  } else {
    la.lengths(p);
    detect_store_lds(la.cnt, tid, BLOCK, a, mu, w, w2, ray);
  }
}

// ---------------------------------------------------------------------------------------------
// rows4_kernel: 4 rows per lane, packed-byte counts.  Requires nz % 4 == 0, z_first % 4 == 0,
// 2 <= NM <= 4 and material ids < NM in the volume.
constexpr int kSuper = 512;    // slabs staged per pass: 2 KB full list + 8 KB crossing list


// GROUPED: a material-group pass (a.acc_out set): raw accumulators out, no detection.  A template parameter rather
// than a run-time branch so that the detection loop's registers do not set the occupancy of the group passes.
template <int NM, int BLOCK, bool GROUPED>
__global__ __launch_bounds__(BLOCK) void rows4_kernel(ProjArgs a, const float* __restrict__ mu,
                                                       const float* __restrict__ w, const float* __restrict__ w2, int n_chunks) {
  __shared__ uint32_t list_full[kSuper];
  __shared__ CrossRec list_cross[kSuper];
  __shared__ uint32_t wave_tot[kSuper / BLOCK][BLOCK / 64][2];
  // A slab with one piece outside the grid exists at most twice per ray: where it enters through a v-face (piece
  // a outside; before every crossing slab) and where it leaves through one (piece b outside; after every crossing
  // slab).  They are kept out of the crossing list, whose records then need no validity masks.
  __shared__ CrossRec edge_rec[2];       // [0] entering, [1] leaving; offset ~0u: absent in this pass
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  BlockMasks bm{{~0ull, ~0ull}, false};
  if (!GROUPED) bm = detect_block_masks(w, a.n_energies, a.n_spectra);       // every lane of every wave is here
  const BlockRay br = block_to_ray(a.n_local_views, a.g.n_channels, n_chunks, a.view_tile);
  const int chunk = br.chunk, c = br.c, v = br.v;
  const int r0 = (chunk * BLOCK + tid) * 4;          // first of this lane's 4 rows
  const dexct_ray_plan p = a.plan[(size_t)v * a.g.n_channels + c];   // uniform
  const int axis = p.flags & 1u;
  const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
  const int nv = axis == 0 ? a.g.ny : a.g.nx;
  const uint32_t su = (axis == 0 ? 1u : (uint32_t)a.g.nx) * (uint32_t)a.g.nz;
  const uint32_t sv = (axis == 0 ? (uint32_t)a.g.nx : 1u) * (uint32_t)a.g.nz;
  // lanes past the last row read the last aligned dword of the column (harmless) and store nothing
  const uint32_t zl = (uint32_t)min(a.g.z_first + r0, a.g.nz - 4);
  // SRSRC buffer loads: the column offset is wave-uniform (it comes from the LDS lists) and goes into the
  // scalar offset, the lane's row offset zl is the vector offset: no per-load address arithmetic, and an
  // out-of-range offset reads 0 instead of faulting.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(a.vol_zf), 0, (int)((size_t)a.g.nx * a.g.ny * a.g.nz), 0x00020000);
  auto ld4 = [&](uint32_t off) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)zl, (int)__builtin_amdgcn_readfirstlane((int)off), 0);
  };
  // Packed byte counters (one byte per row of the lane), wide per-row counters behind them.
  //   NM <= 3 (ids 0, 1, 2):  c0 = sum of (x & 1) = n1,  c1 = sum of x = n1 + 2 n2   -> 3 vector ops per slab, or 2
  //                           when two slabs share a v_add3_u32; a byte grows by at most 2 per slab
  //   NM == 4 (ids 0..3):     c0 = bit-0 plane, c1 = bit-1 plane, c01 = both bits    -> 7 ops, 1 per slab and byte
  // flush threshold: the full-slab loop stops below it, a group of 4 crossing slabs may then still be counted
  // before the next check: 247 + 4 <= 255 increments of 1, or (122 + 4) * 2 <= 255 increments of 2
  constexpr int kFlushAt = NM > 3 ? 248 : 123;
  uint32_t c0 = 0, c1 = 0, c01 = 0;
  uint32_t w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0}, w01[4] = {0, 0, 0, 0};
  float corr[4][4];   // [material][row]
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) corr[m][q] = 0.0f;
  int pending = 0;    // slabs counted into the packed bytes since the last flush (<= kFlushAt)

  auto flush = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (NM != 2) w0[q] += (c0 >> (8 * q)) & 0xFFu;
      w1[q] += (c1 >> (8 * q)) & 0xFFu;
      if (NM > 3) w01[q] += (c01 >> (8 * q)) & 0xFFu;
    }
    c0 = c1 = c01 = 0;
    pending = 0;
  };
  auto count4 = [&](uint32_t x) {
    if (NM > 3) {
      const uint32_t b0 = x & 0x01010101u, b1 = (x >> 1) & 0x01010101u;
      c0 += b0;
      c1 += b1;
      c01 += b0 & b1;
    } else {
      if (NM == 3) c0 += x & 0x01010101u;
      c1 += x;
    }
  };

  constexpr int kPasses = kSuper / BLOCK, kWaves = BLOCK / 64;
  for (int s0 = 0; s0 < p.n_slabs; s0 += kSuper) {
    const int n_super = min(kSuper, p.n_slabs - s0);
    // ---- geometry: every thread classifies kPasses slabs (slab = q*BLOCK + tid), one ballot each; the
    // per-wave totals meet in LDS once, then everybody derives its list positions (slab order is kept).
    uint32_t offa[kPasses], offb[kPasses], below_f[kPasses], below_c[kPasses];
    float tt[kPasses];
    bool is_full[kPasses], is_cross[kPasses];
    if (tid < 2) edge_rec[tid].offa = ~0u;               // ordered before the writes below by the first barrier
#pragma unroll
    for (int q = 0; q < kPasses; ++q) {
      const int s = q * BLOCK + tid;
      is_full[q] = is_cross[q] = false;
      offa[q] = offb[q] = ~0u;
      tt[q] = 0.0f;
      if (s < n_super) {
        const int i = p.i_first + s0 + s;
        const SlabPieces sp = dda_slab(p.V0 + (long long)i * p.SV, p.SV, smask, p.kf);
        const bool ina = (uint32_t)sp.ja < (uint32_t)nv, inb = (uint32_t)sp.jb < (uint32_t)nv;
        if (ina) offa[q] = (uint32_t)i * su + (uint32_t)sp.ja * sv;
        if (inb) offb[q] = (uint32_t)i * su + (uint32_t)sp.jb * sv;
        tt[q] = sp.t;
        is_full[q] = ina && inb && sp.ja == sp.jb;      // both pieces in one voxel column: count only
        is_cross[q] = ina && inb && sp.ja != sp.jb;
      }
      const unsigned long long mf = __ballot(is_full[q]), mc = __ballot(is_cross[q]);
      const unsigned long long lower = (1ull << lane) - 1ull;
      below_f[q] = (uint32_t)__popcll(mf & lower);
      below_c[q] = (uint32_t)__popcll(mc & lower);
      if (lane == 0) { wave_tot[q][wid][0] = (uint32_t)__popcll(mf); wave_tot[q][wid][1] = (uint32_t)__popcll(mc); }
    }
    __syncthreads();
    uint32_t run_f = 0, run_c = 0;
#pragma unroll
    for (int q = 0; q < kPasses; ++q) {
#pragma unroll
      for (int k = 0; k < kWaves; ++k) {
        if (k == wid) {
          if (is_full[q]) list_full[run_f + below_f[q]] = offb[q];
          if (is_cross[q]) list_cross[run_c + below_c[q]] = CrossRec{offa[q], offb[q], tt[q], 0u};
          if (!is_full[q] && !is_cross[q] && (offa[q] != ~0u || offb[q] != ~0u)) {
            // one piece inside: the record keeps the inside column in offa; [0] = piece a is the outside one
            const bool entering = offa[q] == ~0u;
            edge_rec[entering ? 0 : 1] = CrossRec{entering ? offb[q] : offa[q], 0u, tt[q], 0u};
          }
        }
        run_f += wave_tot[q][k][0];
        run_c += wave_tot[q][k][1];
      }
    }
    const int n_full = (int)run_f, n_cross = (int)run_c;
    __syncthreads();
    // ---- full slabs: one dword (4 rows) per slab, integer counts only; 8 loads in flight
    int s = 0;
    while (s < n_full) {
      const int batch = min(n_full - s, kFlushAt - pending);
      int k = 0;
      for (; k + 8 <= batch; k += 8) {
        uint32_t x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = ld4(list_full[s + k + q]);
        __builtin_amdgcn_sched_barrier(0);              // all 8 loads in flight before the first count (see below)
#pragma unroll
        for (int q = 0; q < 8; ++q) count4(x[q]);
      }
      for (; k < batch; ++k) count4(ld4(list_full[s + k]));
      s += batch;
      pending += batch;
      if (pending >= kFlushAt) flush();
    }
    // ---- crossing slabs: two columns; count the b voxel, correct where the two voxels differ.
    // Groups of 4 slabs: 8 loads in flight, no branch unless some row of some slab differs.
    auto correct = [&](uint32_t xa, uint32_t xb, float t) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const uint32_t ia = (xa >> (8 * rr)) & 0xFFu, ib = (xb >> (8 * rr)) & 0xFFu;
        const float td = ia != ib ? t : 0.0f;
#pragma unroll
        for (int m = 1; m < NM; ++m) {
          corr[m][rr] += (ia == (uint32_t)m) ? td : 0.0f;
          corr[m][rr] -= (ib == (uint32_t)m) ? td : 0.0f;
        }
      }
    };
    // ---- the entering edge slab: piece a outside the grid (ids 0), piece b inside
    if (edge_rec[0].offa != ~0u) {
      const uint32_t xb = ld4(edge_rec[0].offa);
      count4(xb);
      if (++pending >= kFlushAt - 4) flush();
      if (xb != 0u) correct(0u, xb, edge_rec[0].t);
    }
    int k = 0;
    for (; k + 4 <= n_cross; k += 4) {
      uint32_t xa[4], xb[4];
      float t4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const CrossRec q = list_cross[k + j];
        xa[j] = ld4(q.offa);
        xb[j] = ld4(q.offb);
        t4[j] = q.t;
      }
      // all 8 loads leave before the first result is used (for 4 materials the scheduler otherwise interleaved
      // each load with the arithmetic on the previous one: one memory latency per slab instead of one per batch)
      __builtin_amdgcn_sched_barrier(0);
      uint32_t any = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        count4(xb[j]);
        any |= xa[j] ^ xb[j];
      }
      pending += 4;
      if (pending >= kFlushAt - 4) flush();
      if (any) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (xa[j] != xb[j]) correct(xa[j], xb[j], t4[j]);
      }
    }
    for (; k < n_cross; ++k) {
      const CrossRec q = list_cross[k];
      const uint32_t xa = ld4(q.offa);
      const uint32_t xb = ld4(q.offb);
      count4(xb);
      if (++pending >= kFlushAt - 4) flush();
      if (xa != xb) correct(xa, xb, q.t);
    }
    // ---- the leaving edge slab: piece a inside, piece b outside the grid (ids 0): nothing to count
    if (edge_rec[1].offa != ~0u) {
      const uint32_t xa = ld4(edge_rec[1].offa);
      if (xa != 0u) correct(xa, 0u, edge_rec[1].t);
    }
    __syncthreads();
  }
  flush();
  float L[4][NM];
  size_t rays[4];
  bool valid[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = r0 + rr;
    valid[rr] = r < a.g.n_rows;
    rays[rr] = ray_index(a, v, valid[rr] ? r : 0, c);
    // ids are 0..NM-1: bit-plane counts -> per-material counts
    uint32_t n[4];
    n[3] = NM > 3 ? w01[rr] : 0u;
    n[1] = NM > 3 ? w0[rr] - n[3] : (NM == 3 ? w0[rr] : w1[rr]);
    n[2] = NM > 3 ? w1[rr] - n[3] : (w1[rr] - w0[rr]) >> 1;
#pragma unroll
    for (int m = 1; m < NM; ++m) L[rr][m] = (float)(int32_t)n[m] + corr[m][rr];
  }
  if (GROUPED) {         // material-group pass: hand the raw accumulators to detect_kernel, one plane per material
    const size_t n_rays = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
    const bool vec4 = a.layout == 1 && (a.g.n_rows & 3) == 0 && valid[3];
#pragma unroll
    for (int m = 1; m < NM; ++m) {
      float* plane = a.acc_out + (size_t)(a.mat_base + m) * n_rays;
      if (vec4) {
        *reinterpret_cast<float4*>(plane + rays[0]) = make_float4(L[0][m], L[1][m], L[2][m], L[3][m]);
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (valid[rr]) plane[rays[rr]] = L[rr][m];
      }
    }
    return;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    float others = 0.0f;
#pragma unroll
    for (int m = 1; m < NM; ++m) others += L[rr][m];
    L[rr][0] = (p.chord_u - others) * p.len_per_u;
#pragma unroll
    for (int m = 1; m < NM; ++m) L[rr][m] *= p.len_per_u;
  }
  detect_store<NM, 4>(L, a, mu, w, w2, rays, valid, bm);
}

// ---------------------------------------------------------------------------------------------
// rows4t_kernel: the packed 4-rows-per-lane traversal with several (view, channel) pairs per workgroup walking the
// volume IN STEP.  Why: a ray plane's voxel columns are read by ~250 other ray planes of the scan, but in
// rows4_kernel every workgroup runs at its own pace (a workgroup lives ~1400 slab steps and starts whenever a slot
// frees), so two workgroups that touch the same column do so hundreds of microseconds apart and the 4 MiB L2 of an
// XCD has long replaced it: at 1024^3 the L2 -> fabric traffic equals the algorithmic bytes (no reuse at all, 87x
// the compulsory bytes).  Here a workgroup is a TILE of NPAIR = TV x TC neighbouring
// (view, channel) pairs, one wave per pair, each wave covering 256 detector rows (64 lanes x 4 rows) of one
// z-chunk.  All waves loop over the same ABSOLUTE slab windows (slab i = plane u = i of the volume) and meet at a
// workgroup barrier every SUB slabs, so at any time the tile reads the few adjacent columns its rays share in
// plane i: the first wave's miss is the other waves' L1 / L2 hit.
// Per wave the structure is rows4_kernel's: every lane classifies one slab of a 64-slab window, ballots compact
// the window into a full list and a crossing list (wave-local LDS, slab order kept), the lanes then sweep the
// lists with packed-byte counts; corrections are applied in slab order, so per-material path lengths stay
// bit-identical to rows4_kernel's and the oracle's.
constexpr int kTW = 64;   // slabs per geometry window: one per lane

template <int NM, int NPAIR, int SUB, bool GROUPED, int FB = 8>      // FB: dword loads in flight per lane and batch
__global__ __launch_bounds__(NPAIR * 64) void rows4t_kernel(ProjArgs a, const float* __restrict__ mu,
                                                              const float* __restrict__ w, const float* __restrict__ w2,
                                                              int tile_v, int n_zchunks) {
  // SUB = 128: no barrier at all (the waves of a tile start together and do the same amount of work per window)
  static_assert(SUB == 128 || (kTW % SUB == 0 && SUB >= 8), "sub-windows of at least 8 slabs that divide the window");
  constexpr int kSub = SUB > kTW ? kTW : SUB;
  __shared__ uint32_t list_full[NPAIR][kTW];
  __shared__ CrossRec list_cross[NPAIR][kTW];
  __shared__ CrossRec edge_rec[NPAIR][2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  BlockMasks bm{{~0ull, ~0ull}, false};
  if (!GROUPED) bm = detect_block_masks(w, a.n_energies, a.n_spectra);       // every lane of every wave is here
  const int tile_c = NPAIR / tile_v;
  // block -> (z-chunk fastest, then view tile, then channel tile), a contiguous range of logical ids per XCD
  const uint32_t nblk = gridDim.x, b = blockIdx.x, per = nblk >> 3;
  uint32_t logical = (b < (per << 3)) ? (b & 7u) * per + (b >> 3) : b;
  const int zc = logical % (uint32_t)n_zchunks;
  logical /= (uint32_t)n_zchunks;
  const int n_tv = (a.n_local_views + tile_v - 1) / tile_v;
  const int tv = logical % (uint32_t)n_tv, tc = logical / (uint32_t)n_tv;
  const int v = tv * tile_v + wid % tile_v, c = tc * tile_c + wid / tile_v;
  const bool pair_live = v < a.n_local_views && c < a.g.n_channels;      // wave-uniform
  dexct_ray_plan p;
  if (pair_live) {
    const dexct_ray_plan q = a.plan[(size_t)v * a.g.n_channels + c];
    // the plan is the same for the whole wave: keep it in scalar registers
    p.V0 = ((long long)__builtin_amdgcn_readfirstlane((int)(q.V0 >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)q.V0);
    p.SV = ((long long)__builtin_amdgcn_readfirstlane((int)(q.SV >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)q.SV);
    p.i_first = __builtin_amdgcn_readfirstlane(q.i_first);
    p.n_slabs = __builtin_amdgcn_readfirstlane(q.n_slabs);
    p.kf = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(q.kf)));
    p.len_per_u = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(q.len_per_u)));
    p.chord_u = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(q.chord_u)));
    p.flags = __builtin_amdgcn_readfirstlane(q.flags);
  } else {
    p.V0 = 0; p.SV = 0; p.i_first = 0; p.n_slabs = 0; p.kf = 0; p.len_per_u = 0; p.chord_u = 0; p.flags = 0;
  }
  const int axis = p.flags & 1u;
  const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
  const int nv = axis == 0 ? a.g.ny : a.g.nx;
  const uint32_t su = (axis == 0 ? 1u : (uint32_t)a.g.nx) * (uint32_t)a.g.nz;
  const uint32_t sv = (axis == 0 ? (uint32_t)a.g.nx : 1u) * (uint32_t)a.g.nz;
  const int r0 = zc * 256 + lane * 4;                    // first of this lane's 4 rows
  const uint32_t zl = (uint32_t)min(a.g.z_first + r0, a.g.nz - 4);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(a.vol_zf), 0, (int)((size_t)a.g.nx * a.g.ny * a.g.nz), 0x00020000);
  auto ld4 = [&](uint32_t off) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)zl, (int)__builtin_amdgcn_readfirstlane((int)off), 0);
  };
  constexpr int kFlushAt = NM > 3 ? 248 : 123;
  uint32_t c0 = 0, c1 = 0, c01 = 0;
  uint32_t w0[4] = {0, 0, 0, 0}, w1[4] = {0, 0, 0, 0}, w01[4] = {0, 0, 0, 0};
  float corr[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) corr[m][q] = 0.0f;
  int pending = 0;
  auto flush = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (NM != 2) w0[q] += (c0 >> (8 * q)) & 0xFFu;
      w1[q] += (c1 >> (8 * q)) & 0xFFu;
      if (NM > 3) w01[q] += (c01 >> (8 * q)) & 0xFFu;
    }
    c0 = c1 = c01 = 0;
    pending = 0;
  };
  auto count4 = [&](uint32_t x) {
    if (NM > 3) {
      const uint32_t b0 = x & 0x01010101u, b1 = (x >> 1) & 0x01010101u;
      c0 += b0;
      c1 += b1;
      c01 += b0 & b1;
    } else {
      if (NM == 3) c0 += x & 0x01010101u;
      c1 += x;
    }
  };
  auto correct = [&](uint32_t xa, uint32_t xb, float t) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const uint32_t ia = (xa >> (8 * rr)) & 0xFFu, ib = (xb >> (8 * rr)) & 0xFFu;
      const float td = ia != ib ? t : 0.0f;
#pragma unroll
      for (int m = 1; m < NM; ++m) {
        corr[m][rr] += (ia == (uint32_t)m) ? td : 0.0f;
        corr[m][rr] -= (ib == (uint32_t)m) ? td : 0.0f;
      }
    }
  };
  const int n_u = max(a.g.nx, a.g.ny);                  // the same trip count for every wave: the barriers need it
  const int i_end = p.i_first + p.n_slabs;
  for (int W0 = 0; W0 < n_u; W0 += kTW) {
    // ---- geometry of slabs W0 .. W0 + 63 (absolute slab index = plane of the volume), one per lane
    unsigned long long mf = 0ull, mc = 0ull;
    const bool window_live = W0 < i_end && W0 + kTW > p.i_first;       // wave-uniform
    if (window_live) {
      const int i = W0 + lane;
      bool is_full = false, is_cross = false;
      uint32_t offa = ~0u, offb = ~0u;
      float tt = 0.0f;
      if (i >= p.i_first && i < i_end) {
        const SlabPieces sp = dda_slab(p.V0 + (long long)i * p.SV, p.SV, smask, p.kf);
        const bool ina = (uint32_t)sp.ja < (uint32_t)nv, inb = (uint32_t)sp.jb < (uint32_t)nv;
        if (ina) offa = (uint32_t)i * su + (uint32_t)sp.ja * sv;
        if (inb) offb = (uint32_t)i * su + (uint32_t)sp.jb * sv;
        tt = sp.t;
        is_full = ina && inb && sp.ja == sp.jb;
        is_cross = ina && inb && sp.ja != sp.jb;
      }
      mf = __ballot(is_full);
      mc = __ballot(is_cross);
      const unsigned long long lower = (1ull << lane) - 1ull;
      if (lane < 2) edge_rec[wid][lane].offa = ~0u;
      if (is_full) list_full[wid][__popcll(mf & lower)] = offb;
      if (is_cross) list_cross[wid][__popcll(mc & lower)] = CrossRec{offa, offb, tt, 0u};
      if (!is_full && !is_cross && (offa != ~0u || offb != ~0u)) {
        const bool entering = offa == ~0u;
        edge_rec[wid][entering ? 0 : 1] = CrossRec{entering ? offb : offa, 0u, tt, (uint32_t)lane};
      }
    }
    // (no barrier needed here: the lists are wave-local and a wave's LDS operations execute in order)
    // the entering edge slab precedes every crossing slab of the ray, the leaving one follows them all
    if (window_live && edge_rec[wid][0].offa != ~0u) {
      const uint32_t xb = ld4(edge_rec[wid][0].offa);
      count4(xb);
      if (++pending >= kFlushAt - 2 * FB) flush();
      if (xb != 0u) correct(0u, xb, edge_rec[wid][0].t);
    }
#pragma unroll 1
    for (int sub = 0; sub < kTW; sub += kSub) {
      if (window_live) {
        const unsigned long long below = (sub == 0) ? 0ull : ((1ull << sub) - 1ull);
        const unsigned long long upto = (sub + kSub >= 64) ? ~0ull : ((1ull << (sub + kSub)) - 1ull);
        const int f0 = __popcll(mf & below), f1 = __popcll(mf & upto);
        const int x0 = __popcll(mc & below), x1 = __popcll(mc & upto);
        // ---- full slabs of this sub-window: 8 dword loads in flight, integer counts only.  Whole batches first;
        // the tail batch re-reads the last entry for its missing slots (an L1 hit) and masks those values to 0, so
        // that its loads, too, leave together instead of one basic block - and one LDS latency - at a time.
        int k = f0;
        for (; k + FB <= f1; k += FB) {
          uint32_t x[FB];
#pragma unroll
          for (int q = 0; q < FB; ++q) x[q] = ld4(list_full[wid][k + q]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < FB; ++q) count4(x[q]);
          pending += FB;
          if (pending >= kFlushAt - 2 * FB) flush();
        }
        if (k < f1) {
          uint32_t x[FB];
#pragma unroll
          for (int q = 0; q < FB; ++q) x[q] = ld4(list_full[wid][min(k + q, f1 - 1)]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < FB; ++q) count4(x[q] & ((k + q < f1) ? ~0u : 0u));      // uniform mask
          pending += f1 - k;
          if (pending >= kFlushAt - 2 * FB) flush();
        }
        // ---- crossing slabs of this sub-window: two columns each, FB loads in flight
        constexpr int CB = FB / 2;
        for (k = x0; k < x1; k += CB) {
          uint32_t xa[CB], xb[CB];
          float t4[CB];
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            const CrossRec q = list_cross[wid][min(k + j, x1 - 1)];
            xa[j] = ld4(q.offa);
            xb[j] = ld4(q.offb);
            t4[j] = q.t;
          }
          __builtin_amdgcn_sched_barrier(0);
          if (k + CB > x1) {                    // tail batch: the repeated last entry counts and corrects nothing
#pragma unroll
            for (int j = 1; j < CB; ++j)
              if (k + j >= x1) { xa[j] = 0u; xb[j] = 0u; }
          }
          uint32_t any = 0;
#pragma unroll
          for (int j = 0; j < CB; ++j) {
            count4(xb[j]);
            any |= xa[j] ^ xb[j];
          }
          pending += CB;
          if (pending >= kFlushAt - 2 * FB) flush();
          if (any) {
#pragma unroll
            for (int j = 0; j < CB; ++j)
              if (xa[j] != xb[j]) correct(xa[j], xb[j], t4[j]);
          }
        }
      }
      if (sub + kSub >= kTW && window_live && edge_rec[wid][1].offa != ~0u) {
        const uint32_t xa = ld4(edge_rec[wid][1].offa);
        if (xa != 0u) correct(xa, 0u, edge_rec[wid][1].t);
      }
      if (SUB <= kTW) __syncthreads();          // pace the tile: no wave runs more than SUB slabs ahead of the others
    }
  }
  if (!pair_live) return;
  flush();
  float L[4][NM];
  size_t rays[4];
  bool valid[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int r = r0 + rr;
    valid[rr] = r < a.g.n_rows;
    rays[rr] = ray_index(a, v, valid[rr] ? r : 0, c);
    uint32_t n[4];
    n[3] = NM > 3 ? w01[rr] : 0u;
    n[1] = NM > 3 ? w0[rr] - n[3] : (NM == 3 ? w0[rr] : w1[rr]);
    n[2] = NM > 3 ? w1[rr] - n[3] : (w1[rr] - w0[rr]) >> 1;
#pragma unroll
    for (int m = 1; m < NM; ++m) L[rr][m] = (float)(int32_t)n[m] + corr[m][rr];
  }
  if (!valid[0]) return;                                  // lanes past the last row of a ragged z-chunk
  if (GROUPED) {
    const size_t n_rays = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
    const bool vec4 = a.layout == 1 && (a.g.n_rows & 3) == 0 && valid[3];
#pragma unroll
    for (int m = 1; m < NM; ++m) {
      float* plane = a.acc_out + (size_t)(a.mat_base + m) * n_rays;
      if (vec4) {
        *reinterpret_cast<float4*>(plane + rays[0]) = make_float4(L[0][m], L[1][m], L[2][m], L[3][m]);
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (valid[rr]) plane[rays[rr]] = L[rr][m];
      }
    }
    return;
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    float others = 0.0f;
#pragma unroll
    for (int m = 1; m < NM; ++m) others += L[rr][m];
    L[rr][0] = (p.chord_u - others) * p.len_per_u;
#pragma unroll
    for (int m = 1; m < NM; ++m) L[rr][m] *= p.len_per_u;
  }
  detect_store<NM, 4>(L, a, mu, w, w2, rays, valid, bm);
}

// ---------------------------------------------------------------------------------------------
// Material groups (5..16 materials).  group_codes_kernel re-encodes the z-fastest volume once per group g:
// ids 3g+1..3g+3 -> codes 1..3, everything else -> 0; rows4_kernel<4> runs once per group on its code volume
// (counts and corrections of a material depend only on whether a voxel IS that material, so the per-material
// accumulators are exactly those of a single pass); detect_kernel then forms L_0 from the chord and applies
// the tables for all materials.
__global__ __launch_bounds__(256) void group_codes_kernel(const uint8_t* __restrict__ vol, size_t n_vox, int n_groups,
                                                          uint8_t* __restrict__ out) {
  const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n_vox) return;
  uint32_t x = 0;
  if (i4 + 4 <= n_vox) x = *reinterpret_cast<const uint32_t*>(vol + i4);
  else for (size_t k = i4; k < n_vox; ++k) x |= (uint32_t)vol[k] << (8 * (k - i4));
  for (int g = 0; g < n_groups; ++g) {
    uint32_t y = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t id = (x >> (8 * b)) & 0xFFu, rel = id - 3u * g;      // 1..3 inside the group
      y |= ((rel >= 1u && rel <= 3u) ? rel : 0u) << (8 * b);
    }
    uint8_t* o = out + (size_t)g * n_vox + i4;
    if (i4 + 4 <= n_vox) *reinterpret_cast<uint32_t*>(o) = y;
    else for (size_t k = i4; k < n_vox; ++k) o[k - i4] = (uint8_t)(y >> (8 * (k - i4)));
  }
}

constexpr int kMaxGrouped = DEXCT_MAX_MATERIALS;   // register detect kernels up to 48 materials, LDS columns beyond

// ---------------------------------------------------------------------------------------------
// Trace: the voxel-index sequence and float32 piece lengths of selected rays (parity tests).
__global__ __launch_bounds__(64) void trace_kernel(dexct_fan_geom g, const dexct_ray_plan* __restrict__ plan,
                                                   const int32_t* __restrict__ ray_vrc, int n_rays, int max_seg,
                                                   int32_t* __restrict__ seg_voxel, float* __restrict__ seg_len,
                                                   int32_t* __restrict__ n_seg) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_rays) return;
  const int v = ray_vrc[3 * k], r = ray_vrc[3 * k + 1], c = ray_vrc[3 * k + 2];
  const dexct_ray_plan p = plan[(size_t)v * g.n_channels + c];
  const int axis = p.flags & 1u;
  const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
  const int nv = axis == 0 ? g.ny : g.nx;
  const int z = g.z_first + r;
  int n = 0;
  long long V = p.V0 + (long long)p.i_first * p.SV;
  for (int s = 0; s < p.n_slabs; ++s) {
    const int i = p.i_first + s;
    const SlabPieces sp = dda_slab(V, p.SV, smask, p.kf);
    const int j[2] = {sp.ja, sp.jb};
    const float l[2] = {sp.t, 1.0f - sp.t};
    for (int q = 0; q < 2; ++q) {
      if ((uint32_t)j[q] >= (uint32_t)nv || !(l[q] > 0.0f)) continue;
      const int x = axis == 0 ? i : j[q], y = axis == 0 ? j[q] : i;
      if (n < max_seg) {
        seg_voxel[(size_t)k * max_seg + n] = (z * g.ny + y) * g.nx + x;
        seg_len[(size_t)k * max_seg + n] = l[q];
      }
      ++n;
    }
    V += p.SV;
  }
  n_seg[k] = n;
}

template <int NM>
static int launch_rays(const ProjArgs& a, const Tables& t, hipStream_t st) {
  constexpr int B = NM > 0 ? kBlock : kLdsBlock;
  size_t lds = NM > 0 ? 0 : (size_t)2 * a.n_materials * B * sizeof(float);
  if constexpr (NM > 0) {
    // register accumulators: one wave per workgroup (no lanes idle beyond the last partial wave of a view: 800 channels
    // are 12.5 waves, not 4 x 256 threads) and batches of slabs with all their loads in flight (DEXCT_RAYS_BATCH=1/4/8/16;
    // measured on the reference's 1000 x 800 single-row scan: 0.620 / 0.440 / 0.450 / 0.457 ms, profiles/r03_kernels.md)
    int batch = 4;
    if (const char* e = getenv("DEXCT_RAYS_BATCH")) batch = atoi(e);
    dim3 grid1((a.g.n_channels + 63) / 64, a.g.n_rows, a.n_local_views);
    if (batch == 8) hipLaunchKernelGGL((rays_kernel<NM, 64, 8>), grid1, dim3(64), 0, st, a, t.mu, t.w, t.w2);
    else if (batch == 4) hipLaunchKernelGGL((rays_kernel<NM, 64, 4>), grid1, dim3(64), 0, st, a, t.mu, t.w, t.w2);
    else if (batch == 16) hipLaunchKernelGGL((rays_kernel<NM, 64, 16>), grid1, dim3(64), 0, st, a, t.mu, t.w, t.w2);
    else {
      dim3 grid((a.g.n_channels + B - 1) / B, a.g.n_rows, a.n_local_views);
      hipLaunchKernelGGL((rays_kernel<NM, B>), grid, dim3(B), lds, st, a, t.mu, t.w, t.w2);
    }
  } else if (a.n_materials > kManyMaterials) {          // 49..256 materials: LDS columns of 64 lanes
    constexpr int B2 = 64;
    lds = (size_t)2 * a.n_materials * B2 * sizeof(float);
    DEXCT_ALLOW_LDS((rays_kernel<NM, B2>), lds);
    dim3 grid((a.g.n_channels + B2 - 1) / B2, a.g.n_rows, a.n_local_views);
    hipLaunchKernelGGL((rays_kernel<NM, B2>), grid, dim3(B2), lds, st, a, t.mu, t.w, t.w2);
  } else {
    dim3 grid((a.g.n_channels + B - 1) / B, a.g.n_rows, a.n_local_views);
    hipLaunchKernelGGL((rays_kernel<NM, B>), grid, dim3(B), lds, st, a, t.mu, t.w, t.w2);
  }
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

template <int NM>
static int launch_rows(const ProjArgs& a, const Tables& t, hipStream_t st) {
  constexpr int B = NM > 0 ? kBlock : kLdsBlock;
  const int n_chunks = (a.g.n_rows + B - 1) / B;
  const size_t nblk = (size_t)a.n_local_views * a.g.n_channels * n_chunks;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  size_t lds = NM > 0 ? 0 : (size_t)2 * a.n_materials * B * sizeof(float);
  if (NM == 0 && a.n_materials > kManyMaterials) {      // 49..256 materials: LDS columns of 64 lanes
    constexpr int B2 = 64;
    const int n_chunks2 = (a.g.n_rows + B2 - 1) / B2;
    const size_t nblk2 = (size_t)a.n_local_views * a.g.n_channels * n_chunks2;
    if (nblk2 > 0x7FFFFFFFull) return DEXCT_ERANGE;
    lds = (size_t)2 * a.n_materials * B2 * sizeof(float);
    DEXCT_ALLOW_LDS((rows_kernel<NM, B2>), lds);
    hipLaunchKernelGGL((rows_kernel<NM, B2>), dim3((unsigned)nblk2), dim3(B2), lds, st, a, t.mu, t.w, t.w2, n_chunks2);
    DEXCT_LAUNCH_CHECK();
    return DEXCT_OK;
  }
  hipLaunchKernelGGL((rows_kernel<NM, B>), dim3((unsigned)nblk), dim3(B), lds, st, a, t.mu, t.w, t.w2, n_chunks);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

template <int NM, int B>
static int launch_rows4_b(const ProjArgs& a, const Tables& t, hipStream_t st) {
  const int rows_per_block = 4 * B;
  const int n_chunks = (a.g.n_rows + rows_per_block - 1) / rows_per_block;
  const size_t nblk = (size_t)a.n_local_views * a.g.n_channels * n_chunks;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  if (a.acc_out)
    hipLaunchKernelGGL((rows4_kernel<NM, B, true>), dim3((unsigned)nblk), dim3(B), 0, st, a, t.mu, t.w, t.w2, n_chunks);
  else
    hipLaunchKernelGGL((rows4_kernel<NM, B, false>), dim3((unsigned)nblk), dim3(B), 0, st, a, t.mu, t.w, t.w2, n_chunks);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

template <int NM>
static int launch_rows4(const ProjArgs& a, const Tables& t, hipStream_t st) {
  // one lane per 4 rows: pick the smallest block that covers the rows in one chunk (up to 256 lanes)
  const int lanes = (a.g.n_rows + 3) / 4;
  if (lanes <= 64) return launch_rows4_b<NM, 64>(a, t, st);
  if (lanes <= 128) return launch_rows4_b<NM, 128>(a, t, st);
  return launch_rows4_b<NM, 256>(a, t, st);
}

// Tile shape of rows4t_kernel: 8 pairs = 2 views x 4 channels, barrier every 64 slabs - the fastest tile of the sweep of
// round 2 (profiles/r02_tiled_1024.md: 21.8 ms on the 1024^3 share; 16 : 4 : 16 has the lowest fabric traffic, 61-66 GB, but
// takes 40.6 ms).  The kernel is opt-in (kernel = 5, A/B against the row kernels; rows16_kernel on the packed volume superseded
// both); the other tile shapes and the 16-row fetch form of the sweep left the library in round 5 (they were 100 s of its
// 2-minute build).
template <int NM>
static int launch_rows4t(const ProjArgs& a, const Tables& t, hipStream_t st) {
  constexpr int kPairs = 8, kSub = 64;
  const int tile_v = 2, tile_c = kPairs / tile_v;
  const int n_z = (a.g.n_rows + 255) / 256;
  const size_t nblk = (size_t)((a.n_local_views + tile_v - 1) / tile_v) * ((a.g.n_channels + tile_c - 1) / tile_c) * n_z;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  if (a.acc_out)
    hipLaunchKernelGGL((rows4t_kernel<NM, kPairs, kSub, true>), dim3((unsigned)nblk), dim3(kPairs * 64), 0, st, a, t.mu, t.w, t.w2,
                       tile_v, n_z);
  else
    hipLaunchKernelGGL((rows4t_kernel<NM, kPairs, kSub, false>), dim3((unsigned)nblk), dim3(kPairs * 64), 0, st, a, t.mu, t.w, t.w2,
                       tile_v, n_z);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

}  // namespace dexct

using namespace dexct;

extern "C" {

int dexct_siddon_project(const dexct_fan_geom* geom, const dexct_ray_plan* plan, int32_t view_begin, int32_t view_end,
                         const uint8_t* vol_yx, const uint8_t* vol_xy, const uint8_t* vol_zf, int32_t n_materials,
                         int32_t n_energies, int32_t n_spectra, const float* mu, const float* weights, float* counts,
                         float* pathlen, int32_t kernel, int32_t layout, const float* weights2, float* variance,
                         const dexct_log_out* log_out, void* stream) {
  if (!geom || !plan || !mu || !weights || !counts) return DEXCT_EINVAL;
  if (view_begin < 0 || view_end > geom->n_views || view_end <= view_begin) return DEXCT_EINVAL;
  if (n_materials < 1 || n_energies < 1 || n_spectra < 1 || geom->n_rows < 1) return DEXCT_EINVAL;
  if (n_materials > DEXCT_MAX_MATERIALS || n_spectra > DEXCT_MAX_SPECTRA) return DEXCT_ERANGE;
  if (geom->z_first < 0 || geom->z_first + geom->n_rows > geom->nz) return DEXCT_EINVAL;
  if ((uint64_t)geom->nx * geom->ny * geom->nz > 0xFFFFFFFEull) return DEXCT_ERANGE;  // 32-bit voxel offsets
  const bool can4 = vol_zf && n_materials >= 2 && n_materials <= 4 && geom->nz % 4 == 0 && geom->z_first % 4 == 0;
  if (kernel == 0) kernel = (vol_zf && geom->n_rows >= 64) ? (can4 ? 3 : 2) : 1;
  if (kernel == 1 && (!vol_yx || !vol_xy)) return DEXCT_EINVAL;
  if (kernel == 2 && !vol_zf) return DEXCT_EINVAL;
  if ((kernel == 3 || kernel == 5) && !can4) return DEXCT_EINVAL;
  if (kernel < 1 || kernel > 6 || kernel == 4) return DEXCT_EINVAL;
  if (kernel == 6 && (!vol_yx || !vol_xy || n_materials > 4 || variance)) return DEXCT_EINVAL;
  if (layout != 0 && layout != 1) return DEXCT_EINVAL;
  if ((variance != nullptr) != (weights2 != nullptr)) return DEXCT_EINVAL;
  if (geom->n_rows > 65535 || view_end - view_begin > 65535) return DEXCT_ERANGE;
  ProjArgs a;
  a.g = *geom;
  a.plan = plan;
  a.vol_yx = vol_yx;
  a.vol_xy = vol_xy;
  a.vol_zf = vol_zf;
  a.n_local_views = view_end - view_begin;
  a.n_materials = n_materials;
  a.n_energies = n_energies;
  a.n_spectra = n_spectra;
  a.counts = counts;
  a.pathlen = pathlen;
  a.variance = variance;
  a.acc_out = nullptr;
  a.mat_base = 0;
  a.layout = layout;
  if (set_log_out(a, log_out, variance) != DEXCT_OK) return DEXCT_EINVAL;
  const Tables t{mu, weights, weights2};
  a.view_tile = kViewTileDefault;
  if (const char* e = getenv("DEXCT_VIEW_TILE")) { const int t = atoi(e); if (t >= 1 && t <= 4096) a.view_tile = t; }   // tuning knob
  hipStream_t st = as_stream(stream);
  if (kernel == 1) {
    switch (n_materials) {
      case 1: return launch_rays<1>(a, t, st);
      case 2: return launch_rays<2>(a, t, st);
      case 3: return launch_rays<3>(a, t, st);
      case 4: return launch_rays<4>(a, t, st);
      default: return launch_rays<0>(a, t, st);
    }
  }
  if (kernel == 2) {
    switch (n_materials) {
      case 1: return launch_rows<1>(a, t, st);
      case 2: return launch_rows<2>(a, t, st);
      case 3: return launch_rows<3>(a, t, st);
      case 4: return launch_rows<4>(a, t, st);
      default: return launch_rows<0>(a, t, st);
    }
  }
  if (kernel == 6) {        // one wavefront per ray (the north-star mapping; A/B and single-row scans)
    switch (n_materials) {
      case 1: return launch_wave_ray<1>(a, t, st);
      case 2: return launch_wave_ray<2>(a, t, st);
      case 3: return launch_wave_ray<3>(a, t, st);
      default: return launch_wave_ray<4>(a, t, st);
    }
  }
  if (kernel == 5) {        // tiled lockstep form: several (view, channel) pairs per workgroup, 256 rows per wave
    switch (n_materials) {
      case 2: return launch_rows4t<2>(a, t, st);
      case 3: return launch_rows4t<3>(a, t, st);
      default: return launch_rows4t<4>(a, t, st);
    }
  }
  switch (n_materials) {
    case 2: return launch_rows4<2>(a, t, st);
    case 3: return launch_rows4<3>(a, t, st);
    default: return launch_rows4<4>(a, t, st);
  }
}

int dexct_volume_groups(const uint8_t* vol_zf, int64_t n_voxels, int32_t n_materials, uint8_t* codes, void* stream) {
  if (!vol_zf || !codes || n_voxels <= 0 || n_materials < 2) return DEXCT_EINVAL;
  if (n_materials > kMaxGrouped) return DEXCT_ERANGE;
  const int n_groups = (n_materials - 1 + 2) / 3;
  const size_t nblk = ((size_t)n_voxels / 4 + 256) / 256;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  hipLaunchKernelGGL(group_codes_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), vol_zf, (size_t)n_voxels,
                     n_groups, codes);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_siddon_project_grouped(const dexct_fan_geom* geom, const dexct_ray_plan* plan, int32_t view_begin,
                                 int32_t view_end, const uint8_t* codes, int32_t n_materials, int32_t n_energies,
                                 int32_t n_spectra, const float* mu, const float* weights, float* counts, float* pathlen,
                                 float* acc_scratch, int32_t layout, const float* weights2, float* variance,
                                 const dexct_log_out* log_out, const dexct_noise* noise, void* stream) {
  if (!geom || !plan || !codes || !mu || !weights || !counts || !acc_scratch) return DEXCT_EINVAL;
  if (view_begin < 0 || view_end > geom->n_views || view_end <= view_begin) return DEXCT_EINVAL;
  if (n_materials < 2 || n_energies < 1 || n_spectra < 1 || geom->n_rows < 1) return DEXCT_EINVAL;
  if (n_materials > kMaxGrouped || n_spectra > DEXCT_MAX_SPECTRA) return DEXCT_ERANGE;
  if (geom->z_first < 0 || geom->z_first + geom->n_rows > geom->nz) return DEXCT_EINVAL;
  if (geom->nz % 4 != 0 || geom->z_first % 4 != 0) return DEXCT_EINVAL;
  if ((uint64_t)geom->nx * geom->ny * geom->nz > 0xFFFFFFFEull) return DEXCT_ERANGE;
  if (layout != 0 && layout != 1) return DEXCT_EINVAL;
  if (geom->n_rows > 65535 || view_end - view_begin > 65535) return DEXCT_ERANGE;
  ProjArgs a;
  a.g = *geom;
  a.plan = plan;
  a.vol_yx = nullptr;
  a.vol_xy = nullptr;
  a.n_local_views = view_end - view_begin;
  a.n_materials = n_materials;
  a.n_energies = n_energies;
  a.n_spectra = n_spectra;
  a.counts = counts;
  a.pathlen = pathlen;
  a.variance = variance;
  a.layout = layout;
  a.acc_out = acc_scratch;
  { const int nrc = set_noise(a, view_begin, weights2, variance, noise); if (nrc != DEXCT_OK) return nrc; }
  // (the log of a noisy sinogram is the log of the SAMPLED counts: written here only when the detection pass draws the sample)
  if (set_log_out(a, log_out, a.sample ? nullptr : variance) != DEXCT_OK) return DEXCT_EINVAL;
  a.view_tile = kViewTileDefault;
  if (const char* e = getenv("DEXCT_VIEW_TILE")) { const int t = atoi(e); if (t >= 1 && t <= 4096) a.view_tile = t; }
  const Tables t{mu, weights, weights2};
  hipStream_t st = as_stream(stream);
  const size_t n_vox = (size_t)geom->nx * geom->ny * geom->nz;
  const int n_groups = (n_materials - 1 + 2) / 3;
  for (int g = 0; g < n_groups; ++g) {
    a.vol_zf = codes + (size_t)g * n_vox;
    a.mat_base = 3 * g;
    const int left = n_materials - 1 - 3 * g;           // materials in this group: 1..3
    int rc;
    if (left >= 3) rc = launch_rows4<4>(a, t, st);
    else if (left == 2) rc = launch_rows4<3>(a, t, st);
    else rc = launch_rows4<2>(a, t, st);
    if (rc != DEXCT_OK) return rc;
  }
  return launch_detect_any(a, t, st);
}

int dexct_siddon_trace(const dexct_fan_geom* geom, const dexct_ray_plan* plan, const int32_t* ray_vrc, int32_t n_rays,
                       int32_t max_seg, int32_t* seg_voxel, float* seg_len, int32_t* n_seg, void* stream) {
  if (!geom || !plan || !ray_vrc || !seg_voxel || !seg_len || !n_seg || n_rays <= 0 || max_seg <= 0)
    return DEXCT_EINVAL;
  hipLaunchKernelGGL(trace_kernel, dim3((n_rays + 63) / 64), dim3(64), 0, as_stream(stream), *geom, plan, ray_vrc,
                     n_rays, max_seg, seg_voxel, seg_len, n_seg);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

}  // extern "C"
