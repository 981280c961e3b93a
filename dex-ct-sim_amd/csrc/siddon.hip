// Siddon forward projection + polychromatic detection for gfx950.
//
// Replaces get_sino (reference call site main.py:120; algorithm: Siddon 1985 as named in
// README.md:27-28,41).  The exact radiological path is computed in a slab-stepping form:
// along the dominant in-plane axis u a ray crosses one slab per step and at most one plane of
// the minor axis v inside it, so every slab contributes two pieces (voxel (i, ja) with length t
// and voxel (i, jb) with length 1 - t, in units of u).  v is a 40-bit fixed-point number stepped
// by integer addition, so voxel indices are exact; t is one float32 multiply on the top 32
// fraction bits.  Path lengths are accumulated per MATERIAL (energy independent), then one
// pass over the energy bins applies the attenuation table and the detector weighting for every
// spectrum (weights = I0 * eta * [E] * dE, the forward model of matdecomp.py:146-150).
//
// Two traversal kernels:
//   rays_kernel  one thread per ray, lanes over adjacent channels (coalesced on the layout whose
//                minor axis is contiguous); any number of rows, the kernel for 2-D scans.
//   rows_kernel  one workgroup per (view, channel): all rows of a stacked fan share the in-plane
//                traversal, so the slab records are computed once per workgroup into LDS and every
//                lane (= detector row = z-slice) only loads its voxel byte from the z-fastest
//                layout (64 consecutive bytes per wave) and accumulates.
// Tables are wave-uniform and are read through the scalar cache (s_load), not LDS.
#include "common.h"

namespace dexct {

constexpr int kBlock = 256;
constexpr float kLog2e = 1.44269504088896340736f;

struct SlabPieces {
  int32_t ja, jb;
  float la, lb;
};

// One slab of the fixed-point DDA (mirror: oracle/dexct_oracle.c dda_slab).
__device__ __forceinline__ SlabPieces dda_slab(long long Va, long long SV, uint32_t smask, float kf) {
  SlabPieces s;
  const long long Vb = Va + SV;
  s.ja = (int32_t)(Va >> DEXCT_FIX_FRAC);
  s.jb = (int32_t)(Vb >> DEXCT_FIX_FRAC);
  const uint32_t fr = (uint32_t)((unsigned long long)Va >> 8);
  const float d = (float)(fr ^ smask);
  const float t = fminf(d * kf, 1.0f);
  s.la = t;
  s.lb = 1.0f - t;
  return s;
}

// counts[s] = sum_e w[s][e] * exp(-sum_m mu2[m][e] * L[m]) (v_exp_f32 on the log2(e)-scaled exponent).
// mu2 and w are wave-uniform; NM is the number of materials held in registers.
template <int NM>
__device__ __forceinline__ void detect_store(const float (&L)[NM], int n_energies, int n_spectra,
                                             const float* __restrict__ mu2, const float* __restrict__ w,
                                             float* __restrict__ counts, size_t out_index, size_t spectrum_stride) {
  float acc[DEXCT_MAX_SPECTRA][2];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) acc[s][0] = acc[s][1] = 0.0f;
  int e = 0;
  for (; e + 1 < n_energies; e += 2) {
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      p0 = fmaf(mu2[m * n_energies + e], L[m], p0);
      p1 = fmaf(mu2[m * n_energies + e + 1], L[m], p1);
    }
    const float t0 = __builtin_amdgcn_exp2f(-p0 * kLog2e), t1 = __builtin_amdgcn_exp2f(-p1 * kLog2e);
#pragma unroll
    for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
      if (s < n_spectra) {
        acc[s][0] = fmaf(w[s * n_energies + e], t0, acc[s][0]);
        acc[s][1] = fmaf(w[s * n_energies + e + 1], t1, acc[s][1]);
      }
  }
  if (e < n_energies) {
    float p0 = 0.0f;
#pragma unroll
    for (int m = 0; m < NM; ++m) p0 = fmaf(mu2[m * n_energies + e], L[m], p0);
    const float t0 = __builtin_amdgcn_exp2f(-p0 * kLog2e);
#pragma unroll
    for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
      if (s < n_spectra) acc[s][0] = fmaf(w[s * n_energies + e], t0, acc[s][0]);
  }
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
    if (s < n_spectra) counts[out_index + s * spectrum_stride] = acc[s][0] + acc[s][1];
}

// Same with the per-material lengths in LDS (column `tid` of lds_L[m*kBlock + tid]).
__device__ __forceinline__ void detect_store_lds(const float* lds_L, int tid, int n_mat, int n_energies,
                                                 int n_spectra, const float* __restrict__ mu2,
                                                 const float* __restrict__ w, float* __restrict__ counts,
                                                 size_t out_index, size_t spectrum_stride) {
  float acc[DEXCT_MAX_SPECTRA];
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s) acc[s] = 0.0f;
  for (int e = 0; e < n_energies; ++e) {
    float p = 0.0f;
    for (int m = 0; m < n_mat; ++m) p = fmaf(mu2[m * n_energies + e], lds_L[m * kBlock + tid], p);
    const float t = __builtin_amdgcn_exp2f(-p * kLog2e);
#pragma unroll
    for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
      if (s < n_spectra) acc[s] = fmaf(w[s * n_energies + e], t, acc[s]);
  }
#pragma unroll
  for (int s = 0; s < DEXCT_MAX_SPECTRA; ++s)
    if (s < n_spectra) counts[out_index + s * spectrum_stride] = acc[s];
}

struct ProjArgs {
  dexct_fan_geom g;
  const dexct_ray_plan* plan;
  const uint8_t* vol_yx;
  const uint8_t* vol_xy;
  const uint8_t* vol_zf;
  int n_local_views;
  int n_materials, n_energies, n_spectra;
  const float* mu2;   // [M][nE] linear attenuation [1/cm]
  const float* w;     // [S][nE]
  float* counts;      // [S][nV][rows][channels]
  float* pathlen;     // optional [ray][M]
};

// ---------------------------------------------------------------------------------------------
// rays_kernel: one thread per ray.  NM > 0: materials 1..NM-1 accumulate in registers;
// NM == 0: any number of materials, accumulators in LDS (one column per thread, conflict free).
template <int NM>
__global__ __launch_bounds__(kBlock) void rays_kernel(ProjArgs a) {
  extern __shared__ float lds_acc[];  // NM == 0: [n_materials][kBlock]
  const int tid = threadIdx.x;
  const int c = blockIdx.x * kBlock + tid;
  const int r = blockIdx.y, v = blockIdx.z;
  const bool live = c < a.g.n_channels;
  dexct_ray_plan p;
  if (live) p = a.plan[(size_t)v * a.g.n_channels + c];
  else { p.n_slabs = 0; p.V0 = 0; p.SV = 0; p.i_first = 0; p.kf = 0; p.len_per_u = 0; p.chord_u = 0; p.flags = 0; }
  const int axis = p.flags & 1u;
  const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
  const int nv = axis == 0 ? a.g.ny : a.g.nx;
  const uint8_t* __restrict__ base = (axis == 0 ? a.vol_xy : a.vol_yx) + (size_t)(a.g.z_first + r) * a.g.nx * a.g.ny;
  float acc[NM > 0 ? NM : 1];
  if (NM > 0) {
#pragma unroll
    for (int m = 0; m < (NM > 0 ? NM : 1); ++m) acc[m] = 0.0f;
  } else {
    for (int m = 0; m < a.n_materials; ++m) lds_acc[m * kBlock + tid] = 0.0f;
  }
  long long V = p.V0 + (long long)p.i_first * p.SV;
  uint32_t off = (uint32_t)p.i_first * (uint32_t)nv;
  for (int s = 0; s < p.n_slabs; ++s) {
    const SlabPieces sp = dda_slab(V, p.SV, smask, p.kf);
    const bool ina = (uint32_t)sp.ja < (uint32_t)nv, inb = (uint32_t)sp.jb < (uint32_t)nv;
    const uint32_t ida = ina ? base[off + (uint32_t)sp.ja] : 0u;
    const uint32_t idb = inb ? base[off + (uint32_t)sp.jb] : 0u;
    if (NM > 0) {
#pragma unroll
      for (int m = 1; m < (NM > 0 ? NM : 1); ++m) {
        acc[m] += (ida == (uint32_t)m) ? sp.la : 0.0f;
        acc[m] += (idb == (uint32_t)m) ? sp.lb : 0.0f;
      }
    } else {
      // slot 0 collects air / outside pieces and is never read
      __hip_atomic_fetch_add(&lds_acc[ida * kBlock + tid], sp.la, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(&lds_acc[idb * kBlock + tid], sp.lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    V += p.SV;
    off += (uint32_t)nv;
  }
  if (!live) return;
  const size_t ray = ((size_t)v * a.g.n_rows + r) * a.g.n_channels + c;
  const size_t sstride = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  if (NM > 0) {
    float L[NM > 0 ? NM : 1];
    float others = 0.0f;
#pragma unroll
    for (int m = 1; m < (NM > 0 ? NM : 1); ++m) others += acc[m];
    L[0] = (p.chord_u - others) * p.len_per_u;
#pragma unroll
    for (int m = 1; m < (NM > 0 ? NM : 1); ++m) L[m] = acc[m] * p.len_per_u;
    if (a.pathlen)
      for (int m = 0; m < a.n_materials; ++m) a.pathlen[ray * a.n_materials + m] = m < NM ? L[m < NM ? m : 0] : 0.0f;
    detect_store<(NM > 0 ? NM : 1)>(L, a.n_energies, a.n_spectra, a.mu2, a.w, a.counts, ray, sstride);
  } else {
    float others = 0.0f;
    for (int m = 1; m < a.n_materials; ++m) others += lds_acc[m * kBlock + tid];
    lds_acc[tid] = (p.chord_u - others) * p.len_per_u;
    for (int m = 1; m < a.n_materials; ++m) lds_acc[m * kBlock + tid] *= p.len_per_u;
    if (a.pathlen)
      for (int m = 0; m < a.n_materials; ++m) a.pathlen[ray * a.n_materials + m] = lds_acc[m * kBlock + tid];
    detect_store_lds(lds_acc, tid, a.n_materials, a.n_energies, a.n_spectra, a.mu2, a.w, a.counts, ray, sstride);
  }
}

// ---------------------------------------------------------------------------------------------
// rows_kernel: one workgroup per (view, channel, chunk of kBlock rows).
struct SlabRec {
  uint32_t offa, offb;  // byte offsets of the (x, y) columns in the z-fastest layout
  float la, lb;         // 0 where the piece lies outside the grid
};

template <int NM>
__global__ __launch_bounds__(kBlock) void rows_kernel(ProjArgs a, int n_chunks) {
  __shared__ SlabRec rec[kBlock];
  extern __shared__ float lds_acc[];
  const int tid = threadIdx.x;
  // XCD-aware remap: consecutive logical ids (adjacent channels of one view: rays that share
  // voxel columns) land on the same XCD and therefore in the same L2.
  const uint32_t nblk = gridDim.x;
  const uint32_t b = blockIdx.x;
  const uint32_t per = nblk >> 3;
  const uint32_t logical = (b < (per << 3)) ? (b & 7u) * per + (b >> 3) : b;
  const int chunk = logical % n_chunks;
  const uint32_t vc = logical / n_chunks;
  const int c = vc % a.g.n_channels, v = vc / a.g.n_channels;
  const int r = chunk * kBlock + tid;
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
  float acc[NM > 0 ? NM : 1];
  if (NM > 0) {
#pragma unroll
    for (int m = 0; m < (NM > 0 ? NM : 1); ++m) acc[m] = 0.0f;
  } else {
    for (int m = 0; m < a.n_materials; ++m) lds_acc[m * kBlock + tid] = 0.0f;
  }
  for (int s0 = 0; s0 < p.n_slabs; s0 += kBlock) {
    const int n_here = min(kBlock, p.n_slabs - s0);
    if (tid < n_here) {
      const int i = p.i_first + s0 + tid;
      const SlabPieces sp = dda_slab(p.V0 + (long long)i * p.SV, p.SV, smask, p.kf);
      const bool ina = (uint32_t)sp.ja < (uint32_t)nv, inb = (uint32_t)sp.jb < (uint32_t)nv;
      SlabRec q;
      q.offa = ina ? (uint32_t)i * su + (uint32_t)sp.ja * sv : 0u;
      q.offb = inb ? (uint32_t)i * su + (uint32_t)sp.jb * sv : 0u;
      q.la = ina ? sp.la : 0.0f;
      q.lb = inb ? sp.lb : 0.0f;
      rec[tid] = q;
    }
    __syncthreads();
#pragma unroll 4
    for (int s = 0; s < n_here; ++s) {
      const SlabRec q = rec[s];
      const uint32_t ida = col[q.offa];
      const uint32_t idb = col[q.offb];
      if (NM > 0) {
#pragma unroll
        for (int m = 1; m < (NM > 0 ? NM : 1); ++m) {
          acc[m] += (ida == (uint32_t)m) ? q.la : 0.0f;
          acc[m] += (idb == (uint32_t)m) ? q.lb : 0.0f;
        }
      } else {
        __hip_atomic_fetch_add(&lds_acc[ida * kBlock + tid], q.la, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&lds_acc[idb * kBlock + tid], q.lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    __syncthreads();
  }
  if (!live) return;
  const size_t ray = ((size_t)v * a.g.n_rows + r) * a.g.n_channels + c;
  const size_t sstride = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  if (NM > 0) {
    float L[NM > 0 ? NM : 1];
    float others = 0.0f;
#pragma unroll
    for (int m = 1; m < (NM > 0 ? NM : 1); ++m) others += acc[m];
    L[0] = (p.chord_u - others) * p.len_per_u;
#pragma unroll
    for (int m = 1; m < (NM > 0 ? NM : 1); ++m) L[m] = acc[m] * p.len_per_u;
    if (a.pathlen)
      for (int m = 0; m < a.n_materials; ++m) a.pathlen[ray * a.n_materials + m] = m < NM ? L[m < NM ? m : 0] : 0.0f;
    detect_store<(NM > 0 ? NM : 1)>(L, a.n_energies, a.n_spectra, a.mu2, a.w, a.counts, ray, sstride);
  } else {
    float others = 0.0f;
    for (int m = 1; m < a.n_materials; ++m) others += lds_acc[m * kBlock + tid];
    lds_acc[tid] = (p.chord_u - others) * p.len_per_u;
    for (int m = 1; m < a.n_materials; ++m) lds_acc[m * kBlock + tid] *= p.len_per_u;
    if (a.pathlen)
      for (int m = 0; m < a.n_materials; ++m) a.pathlen[ray * a.n_materials + m] = lds_acc[m * kBlock + tid];
    detect_store_lds(lds_acc, tid, a.n_materials, a.n_energies, a.n_spectra, a.mu2, a.w, a.counts, ray, sstride);
  }
}

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
    const float l[2] = {sp.la, sp.lb};
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
static int launch_rays(const ProjArgs& a, hipStream_t st) {
  dim3 grid((a.g.n_channels + kBlock - 1) / kBlock, a.g.n_rows, a.n_local_views);
  size_t lds = NM > 0 ? 0 : (size_t)a.n_materials * kBlock * sizeof(float);
  hipLaunchKernelGGL(rays_kernel<NM>, grid, dim3(kBlock), lds, st, a);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

template <int NM>
static int launch_rows(const ProjArgs& a, hipStream_t st) {
  const int n_chunks = (a.g.n_rows + kBlock - 1) / kBlock;
  const size_t nblk = (size_t)a.n_local_views * a.g.n_channels * n_chunks;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  size_t lds = NM > 0 ? 0 : (size_t)a.n_materials * kBlock * sizeof(float);
  hipLaunchKernelGGL(rows_kernel<NM>, dim3((unsigned)nblk), dim3(kBlock), lds, st, a, n_chunks);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

}  // namespace dexct

using namespace dexct;

extern "C" {

int dexct_siddon_project(const dexct_fan_geom* geom, const dexct_ray_plan* plan, int32_t view_begin, int32_t view_end,
                         const uint8_t* vol_yx, const uint8_t* vol_xy, const uint8_t* vol_zf, int32_t n_materials,
                         int32_t n_energies, int32_t n_spectra, const float* mu, const float* weights, float* counts,
                         float* pathlen, int32_t kernel, void* stream) {
  if (!geom || !plan || !mu || !weights || !counts) return DEXCT_EINVAL;
  if (view_begin < 0 || view_end > geom->n_views || view_end <= view_begin) return DEXCT_EINVAL;
  if (n_materials < 1 || n_energies < 1 || n_spectra < 1 || geom->n_rows < 1) return DEXCT_EINVAL;
  if (n_materials > DEXCT_MAX_MATERIALS || n_spectra > DEXCT_MAX_SPECTRA) return DEXCT_ERANGE;
  if (geom->z_first < 0 || geom->z_first + geom->n_rows > geom->nz) return DEXCT_EINVAL;
  if ((uint64_t)geom->nx * geom->ny * geom->nz > 0xFFFFFFFFull) return DEXCT_ERANGE;  // 32-bit voxel offsets
  if (kernel == 0) kernel = (vol_zf && geom->n_rows >= 64) ? 2 : 1;
  if (kernel == 1 && (!vol_yx || !vol_xy)) return DEXCT_EINVAL;
  if (kernel == 2 && !vol_zf) return DEXCT_EINVAL;
  if (kernel != 1 && kernel != 2) return DEXCT_EINVAL;
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
  a.mu2 = mu;
  a.w = weights;
  a.counts = counts;
  a.pathlen = pathlen;
  hipStream_t st = as_stream(stream);
  if (kernel == 1) {
    switch (n_materials) {
      case 1: return launch_rays<1>(a, st);
      case 2: return launch_rays<2>(a, st);
      case 3: return launch_rays<3>(a, st);
      case 4: return launch_rays<4>(a, st);
      default: return launch_rays<0>(a, st);
    }
  }
  switch (n_materials) {
    case 1: return launch_rows<1>(a, st);
    case 2: return launch_rows<2>(a, st);
    case 3: return launch_rows<3>(a, st);
    case 4: return launch_rows<4>(a, st);
    default: return launch_rows<0>(a, st);
  }
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
