// rows16_kernel: the stacked-fan traversal on a 2-BIT PACKED volume with bit-sliced counters (gfx950).
//
// Material ids of the fast path are < 4, so a voxel needs 2 bits, not 8.  dexct_volume_pack2 stores the z-fastest
// volume with four voxels per byte: column (x, y) is nz / 4 bytes, row z in bits 2(z % 16), 2(z % 16) + 1 of dword
// z / 16.  One dword load then serves SIXTEEN detector rows (rows4_kernel: four), the volume is a quarter of the bytes
// (a 1024^3 phantom is 256 MiB: it fits the Infinity Cache) and a (view, channel) pair of 1024 rows is one wave.
//
// Counting.  The loaded word x IS 32 one-bit flags: bit 2r = "row r sees id 1", bit 2r + 1 = "row r sees id 2" (ids
// 0, 1, 2).  n_1 and n_2 per row are therefore per-bit-position population counts over the slabs of the ray.  They are
// kept BIT-SLICED (Harley-Seal): words ones / twos / fours hold bits 0..2 of the 32 counters, hi[0..7] bits 3..10; eight
// (Id 3 sets both flags: with 4 ids - a full material group, or a 4-material phantom - a second set of counters runs
// on x & (x >> 1), the "both" flags at the even positions, and n_1 = n(bit 0) - n_3, n_2 = n(bit 1) - n_3.)  Eight
// loaded words are folded by seven carry-save adders (3 logic instructions each: a ^ b, (a ^ b) ^ c, majority by bit
// select); two of the resulting weight-8 words meet in an eighth adder and the carry ripples into hi[] once per 16
// words: 30 instructions per 8 dwords = 0.23 per row and slab against 0.75 (+ a v_readfirstlane per 4 rows) in
// rows4_kernel.  The counters are un-sliced once per ray plane.
// Corrections (float32, in slab order, only where a crossing separates two materials) are applied per row exactly as
// in rows4_kernel, so per-material path lengths are bit-identical to every other kernel and to the oracle.
//
// Lanes: a pair of n_rows rows needs n_rows / 16 lanes.  For n_rows / 16 < 64 a wave carries 64 / (n_rows / 16)
// neighbouring channels of one view (per-lane pair index; each pair has its own slab lists in LDS and list entries
// beyond a pair's count point outside the buffer, which reads as air and counts nothing).
#include <cstdlib>

#include "common.h"
#include "noise_sample.h"
#include "siddon_detect.h"

namespace dexct {

constexpr int kP16Super = 128;      // slabs staged per pass
constexpr uint32_t kOob = 0xF0000000u;   // a list offset outside every buffer (< 3.75 GiB): the load returns 0 (air), no traffic

// carry-save adder on 32 one-bit lanes: (h, l) = a + b + c
__device__ __forceinline__ void csa(uint32_t& h, uint32_t& l, uint32_t a, uint32_t b, uint32_t c) {
  const uint32_t u = a ^ b;
  h = (a & b) | (u & c);
  l = u ^ c;
}

struct Sliced {
  uint32_t ones = 0, twos = 0, fours = 0, eights = 0;
  uint32_t hi[7] = {0, 0, 0, 0, 0, 0, 0};         // bits 4..10 of the 32 counters
  uint32_t pend = 0;                               // a word of weight 8 waiting for a partner
  bool have_pend = false;                          // wave-uniform
  __device__ __forceinline__ void add_sixteens(uint32_t carry) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const uint32_t t = hi[k] & carry;
      hi[k] ^= carry;
      carry = t;
    }
  }
  // a word of weight 8: two of them and the running eights go through one more carry-save adder, so the ripple into
  // the high bits runs once per 16 loaded words
  __device__ __forceinline__ void add_eights(uint32_t e) {
    if (!have_pend) {
      pend = e;
      have_pend = true;
    } else {
      uint32_t c;
      csa(c, eights, eights, pend, e);
      add_sixteens(c);
      have_pend = false;
    }
  }
  __device__ __forceinline__ void finish() {
    if (have_pend) {
      const uint32_t c = eights & pend;
      eights ^= pend;
      add_sixteens(c);
      have_pend = false;
    }
  }
  __device__ __forceinline__ void add8(const uint32_t (&x)[8]) {
    uint32_t ta, tb, fa, fb, e;
    csa(ta, ones, ones, x[0], x[1]);
    csa(tb, ones, ones, x[2], x[3]);
    csa(fa, twos, twos, ta, tb);
    csa(ta, ones, ones, x[4], x[5]);
    csa(tb, ones, ones, x[6], x[7]);
    csa(fb, twos, twos, ta, tb);
    csa(e, fours, fours, fa, fb);
    add_eights(e);
  }
  __device__ __forceinline__ void add4(const uint32_t (&x)[4]) {
    uint32_t ta, tb, fa;
    csa(ta, ones, ones, x[0], x[1]);
    csa(tb, ones, ones, x[2], x[3]);
    csa(fa, twos, twos, ta, tb);
    const uint32_t e = fours & fa;
    fours ^= fa;
    add_eights(e);
  }
  // all 32 counters at once (after finish()): out[p] = counter of bit position p.  A 32 x 32 bit-matrix transpose
  // (five rounds of masked swaps) of the 11 planes; the 21 zero rows fold away at compile time: about a third of
  // extracting every bit of every counter on its own.
  __device__ __forceinline__ void unslice(uint32_t (&out)[32]) const {
    out[0] = ones; out[1] = twos; out[2] = fours; out[3] = eights;
#pragma unroll
    for (int k = 0; k < 7; ++k) out[4 + k] = hi[k];
#pragma unroll
    for (int k = 11; k < 32; ++k) out[k] = 0u;
    uint32_t m = 0x0000FFFFu;
#pragma unroll
    for (int j = 16; j != 0; j >>= 1) {
#pragma unroll
      for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
        const uint32_t t = ((out[k] >> j) ^ out[k + j]) & m;
        out[k] ^= t << j;
        out[k + j] ^= t;
      }
      m ^= m << (j >> 1);
    }
  }
};

// GROUPED: a material-group pass (ids = codes 0..3 of one group of three materials): raw accumulators (units of u) go to
// acc_out[(mat_base + code) * n_rays + ray], no detection (dexct_siddon_project_grouped_packed).
// STAGED: results leave through LDS as whole lines (see below); the host picks it whenever it applies.
// NOISY (round 6; at most two spectra): the scan with quantum noise - the reference's live mode, spectra scaled to a dose
// (main.py:68,101).  The detection rounds accumulate the variance of the signal next to the signal (detect_energies<VAR>:
// the exponentials are shared) and, pa.sample, draw the sample in the registers that hold both (noise_sample.h: one Philox
// block per ray serves both spectra): what leaves the kernel is the noisy sinogram - no variance array, no pass over it.
// a.variance (optional) receives the variances as well (tests; callers that sample with dexct_add_noise).
template <int NM, int MINW = 4, bool GROUPED = false, bool STAGED = false, bool NOISY = false>      // MINW: waves per SIMD the register allocation must allow
__global__ __launch_bounds__(64, MINW) void rows16_kernel(PackedArgs pa, const float* __restrict__ mu, const float* __restrict__ w,
                                                    const float* __restrict__ w2) {
  static_assert(NM >= 2 && NM <= 4, "ids 0..3");
  constexpr bool BOTH = NM > 3;              // id 3 exists: count the "both flags" plane too
  extern __shared__ uint32_t lds_lists[];          // per pair: kP16Super crossing records (16 B), then kP16Super offsets
  const ProjArgs& a = pa.a;
  const int lane = threadIdx.x;
  BlockMasks bm{{~0ull, ~0ull}, false};
  if (!GROUPED) {
    bm = detect_block_masks(w, a.n_energies, a.n_spectra);      // all 64 lanes are here
    bm.use = bm.use && pa.det_masks;
  }
  const int lpp = pa.lanes_per_pair, n_pairs = 64 / lpp;
  CrossRec (*list_cross)[kP16Super] = reinterpret_cast<CrossRec (*)[kP16Super]>(lds_lists);
  uint32_t (*list_full)[kP16Super] = reinterpret_cast<uint32_t (*)[kP16Super]>(lds_lists + (size_t)n_pairs * kP16Super * 4);
  const int g = lane / lpp, li = lane - g * lpp;                 // this lane's pair and its index inside the pair
  // block -> (view, channel group, z-chunk): contiguous logical ids per XCD, views fastest inside a tile
  const int n_cg = (a.g.n_channels + n_pairs - 1) / n_pairs;
  const uint32_t nblk = gridDim.x, bid = blockIdx.x, per = nblk >> 3;
  const uint32_t logical = (bid < (per << 3)) ? (bid & 7u) * per + (bid >> 3) : bid;
  const uint32_t group = (uint32_t)pa.view_tile * n_cg * pa.n_zchunks;
  const uint32_t gq = logical / group, rem = logical - gq * group;
  const uint32_t views_here = min((uint32_t)pa.view_tile, (uint32_t)a.n_local_views - gq * pa.view_tile);
  const int zc = rem % pa.n_zchunks;
  const uint32_t qq = rem / pa.n_zchunks;
  const int v = gq * pa.view_tile + qq % views_here;
  const int c = (qq / views_here) * n_pairs + g;
  const bool pair_live = c < a.g.n_channels;
  dexct_ray_plan p;
  if (pair_live) p = a.plan[(size_t)v * a.g.n_channels + c];
  else { p.V0 = 0; p.SV = 0; p.i_first = 0; p.n_slabs = 0; p.kf = 0; p.len_per_u = 0; p.chord_u = 0; p.flags = 0; }
  const int axis = p.flags & 1u;
  const uint32_t smask = (p.flags & 2u) ? 0xFFFFFFFFu : 0u;
  const int nv = axis == 0 ? a.g.ny : a.g.nx;
  const uint32_t zb = (uint32_t)a.g.nz >> 2;                               // bytes per column
  const uint32_t su = (axis == 0 ? 1u : (uint32_t)a.g.nx) * zb;
  const uint32_t sv = (axis == 0 ? (uint32_t)a.g.nx : 1u) * zb;
  const int r0 = (zc * 64 + li) * 16;                                      // first of this lane's 16 rows
  // byte offset of the lane's dword inside a column; lanes past the last row read the last dword and store nothing
  const uint32_t zoff = (uint32_t)min((a.g.z_first + r0) >> 2, (int)zb - 4);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(pa.vol_z2), 0, (int)((size_t)a.g.nx * a.g.ny * zb), 0x00020000);
  auto ld16 = [&](uint32_t off) {           // off + zoff beyond the buffer (list padding, edge pieces): 0, no access
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(off + zoff), 0, 0);
  };
  Sliced cnt, cnt3;                          // cnt3: flags of id 3 at the even bit positions (BOTH only)
  float corr[NM - 1][16];                    // [material - 1][row]
#pragma unroll
  for (int m = 0; m < NM - 1; ++m)
#pragma unroll
    for (int q = 0; q < 16; ++q) corr[m][q] = 0.0f;
  auto correct = [&](uint32_t xa, uint32_t xb, float t) {
    const uint32_t d = xa ^ xb;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
      if ((d >> (2 * rr)) & 3u) {
        const uint32_t ia = (xa >> (2 * rr)) & 3u, ib = (xb >> (2 * rr)) & 3u;
#pragma unroll
        for (int m = 1; m < NM; ++m) {
          corr[m - 1][rr] += (ia == (uint32_t)m) ? t : 0.0f;
          corr[m - 1][rr] -= (ib == (uint32_t)m) ? t : 0.0f;
        }
      }
    }
  };
  // the longest ray of the wave sets the trip count (wave-uniform)
  int n_slabs_max = 0;
  for (int k = 0; k < n_pairs; ++k) n_slabs_max = max(n_slabs_max, __builtin_amdgcn_readlane(p.n_slabs, k * lpp));
  const unsigned long long group_mask = (lpp == 64 ? ~0ull : ((1ull << lpp) - 1ull)) << (g * lpp);
  const unsigned long long below_mask = group_mask & ((1ull << lane) - 1ull);
  for (int s0 = 0; s0 < n_slabs_max; s0 += kP16Super) {
    // ---- geometry: the lanes of a pair classify its slabs s0 + li, s0 + li + lpp, ... (slab order kept per pair)
    int run_f = 0, run_c = 0;               // per pair (identical in all lanes of a pair)
    for (int q0 = 0; q0 < kP16Super; q0 += lpp) {
      const int s = s0 + q0 + li;
      bool is_full = false, is_cross = false;
      uint32_t offa = kOob, offb = kOob;
      float tt = 0.0f;
      if (s < p.n_slabs) {
        const int i = p.i_first + s;
        const SlabPieces sp = dda_slab(p.V0 + (long long)i * p.SV, p.SV, smask, p.kf);
        const bool ina = (uint32_t)sp.ja < (uint32_t)nv, inb = (uint32_t)sp.jb < (uint32_t)nv;
        if (ina) offa = (uint32_t)i * su + (uint32_t)sp.ja * sv;
        if (inb) offb = (uint32_t)i * su + (uint32_t)sp.jb * sv;
        tt = sp.t;
        is_full = ina && inb && sp.ja == sp.jb;
        // everything else that touches the grid goes to the crossing list: a piece outside the grid keeps kOob and
        // reads as air, which is what the edge slabs of rows4_kernel do (count b, correct against id 0)
        is_cross = !is_full && (ina || inb);
      }
      const unsigned long long mf = __ballot(is_full), mc = __ballot(is_cross);
      if (is_full) list_full[g][run_f + __popcll(mf & below_mask)] = offb;
      if (is_cross) list_cross[g][run_c + __popcll(mc & below_mask)] = CrossRec{offa, offb, tt, 0u};
      run_f += __popcll(mf & group_mask);
      run_c += __popcll(mc & group_mask);
    }
    // pad each pair's lists to the wave's longest (whole batches), so that the sweeps below need no per-pair bounds
    int n_full = 0, n_cross = 0;
    for (int k = 0; k < n_pairs; ++k) {
      n_full = max(n_full, __builtin_amdgcn_readlane(run_f, k * lpp));
      n_cross = max(n_cross, __builtin_amdgcn_readlane(run_c, k * lpp));
    }
    n_full = (n_full + 7) & ~7;
    n_cross = (n_cross + 3) & ~3;
    for (int k = run_f + li; k < n_full; k += lpp) list_full[g][k] = kOob;
    for (int k = run_c + li; k < n_cross; k += lpp) list_cross[g][k] = CrossRec{kOob, kOob, 0.0f, 0u};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- full slabs: 8 dword loads (16 rows each) in flight, seven carry-save adders
    for (int k = 0; k < n_full; k += 8) {
      uint32_t x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = ld16(list_full[g][k + q]);
      __builtin_amdgcn_sched_barrier(0);
      cnt.add8(x);
      if (BOTH) {
        uint32_t b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) b[q] = x[q] & (x[q] >> 1);
        cnt3.add8(b);
      }
    }
    // ---- crossing slabs (and the edge slabs): count the b voxel, correct where the two voxels differ
    for (int k = 0; k < n_cross; k += 4) {
      uint32_t xa[4], xb[4];
      float t4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const CrossRec q = list_cross[g][k + j];
        xa[j] = ld16(q.offa);
        xb[j] = ld16(q.offb);
        t4[j] = q.t;
      }
      __builtin_amdgcn_sched_barrier(0);
      cnt.add4(xb);
      if (BOTH) {
        uint32_t b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = xb[j] & (xb[j] >> 1);
        cnt3.add4(b);
      }
      uint32_t any = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) any |= xa[j] ^ xb[j];
      if (any) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (xa[j] != xb[j]) correct(xa[j], xb[j], t4[j]);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  cnt.finish();
  if (BOTH) cnt3.finish();
  // STAGED (the usual case: row-fastest output, whole groups of 4 rows, one or two spectra): each detection round leaves
  // its results in LDS and the wave stores them at the end, so that one store instruction writes 1 KiB of consecutive
  // addresses (whole lines) instead of 64 separate 16-byte pieces per round - every lane's 64 output bytes used to
  // leave as four partial writes (WRITE_SIZE 2.1x the output bytes).  All lanes then stay to the end, because a lane
  // also stores pieces of other lanes.
  constexpr int kResSlots = 2;               // spectra whose results the stage holds
  static_assert(!(GROUPED && STAGED), "group passes hand over raw accumulators");
  static_assert(!(GROUPED && NOISY), "group passes hand over raw accumulators");
  constexpr bool staged = STAGED;            // host: layout 1, n_rows % 4 == 0, n_spectra <= kResSlots
  const bool lane_live = pair_live && r0 < a.g.n_rows;
  if (!staged && !lane_live) return;
  // (STAGED: lanes without rows of their own run the code below too, on counters that saw nothing but air - branching
  // around it per lane made the detection loops divergent in the compiler's eyes, and the tables then arrived by vector
  // loads instead of through the scalar cache: +23 %)
  // ---- un-slice the counters (in place of the corrections), then detect 4 rows at a time.  The rounds are a real
  // loop with ONE copy of the detection code (rows move down the register array between rounds): four inlined copies
  // made the kernel as large as the instruction cache two CUs share.
  {
    // the sums are formed exactly as in rows4_kernel: (float) count + corrections
    uint32_t n[32], n3[32];
    cnt.unslice(n);
    if (BOTH) cnt3.unslice(n3);
#pragma unroll
    for (int row = 0; row < 16; ++row) {
      if (BOTH) {
        corr[0][row] = (float)(int32_t)(n[2 * row] - n3[2 * row]) + corr[0][row];
        corr[1][row] = (float)(int32_t)(n[2 * row + 1] - n3[2 * row]) + corr[1][row];
        corr[2][row] = (float)(int32_t)n3[2 * row] + corr[2][row];
      } else {
        corr[0][row] = (float)(int32_t)n[2 * row] + corr[0][row];
        if (NM > 2) corr[1][row] = (float)(int32_t)n[2 * row + 1] + corr[1][row];
      }
    }
  }
  if (GROUPED) {         // hand the raw accumulators to the detection kernel, one plane per material of the group
    if (!lane_live) return;
    const size_t n_rays = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int r = r0 + 4 * q4;
      if (r >= a.g.n_rows) break;
      const bool vec4 = a.layout == 1 && (a.g.n_rows & 3) == 0;
#pragma unroll
      for (int m = 1; m < NM; ++m) {
        float* plane = a.acc_out + (size_t)(a.mat_base + m) * n_rays;
        if (vec4) {
          *reinterpret_cast<float4*>(plane + ray_index(a, v, r, c)) =
              make_float4(corr[m - 1][4 * q4], corr[m - 1][4 * q4 + 1], corr[m - 1][4 * q4 + 2], corr[m - 1][4 * q4 + 3]);
        } else {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            if (r + rr < a.g.n_rows) plane[ray_index(a, v, r + rr, c)] = corr[m - 1][4 * q4 + rr];
        }
      }
    }
    return;
  }
  AirCache air_cache{0.0f, {0.0f, 0.0f}, false, {0.0f, 0.0f}};
  const size_t sstride_n = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
#pragma unroll 1
  for (int q4 = 0; q4 < 4; ++q4) {
    const bool round_live = lane_live && r0 + 4 * q4 < a.g.n_rows;
    if (!staged && !round_live) break;
    if (staged && __ballot(round_live) == 0ull) break;          // wave-uniform: nobody has rows left
    float res[2][4];
    {
      float L[4][NM];
      size_t rays[4];
      bool valid[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = r0 + 4 * q4 + rr;
        valid[rr] = r < a.g.n_rows;
        rays[rr] = ray_index(a, v, valid[rr] ? r : 0, c);
        float others = 0.0f;
#pragma unroll
        for (int m = 1; m < NM; ++m) others += corr[m - 1][rr];
        L[rr][0] = (p.chord_u - others) * p.len_per_u;
#pragma unroll
        for (int m = 1; m < NM; ++m) L[rr][m] = corr[m - 1][rr] * p.len_per_u;
      }
      if constexpr (staged || NOISY) {
        if (a.pathlen) {             // (test output) written here so that the detection holds no store addresses
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            if (valid[rr] && round_live) {
#pragma unroll
              for (int m = 0; m < NM; ++m) a.pathlen[rays[rr] * NM + m] = L[rr][m];
            }
        }
      }
      if constexpr (NOISY) {
        float var[2][4];
        detect_store<NM, 4, false, true>(L, a, mu, w, w2, rays, valid, bm, &air_cache, &res, &var);
        if (a.variance && round_live) {            // (optional output: plain stores)
#pragma unroll
          for (int s = 0; s < 2; ++s)
            if (s < a.n_spectra) {
#pragma unroll
              for (int rr = 0; rr < 4; ++rr)
                if (valid[rr]) a.variance[rays[rr] + s * sstride_n] = var[s][rr];
            }
        }
        if (pa.sample) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float z[2];
            pixel_normals<2>((uint32_t)(pa.view_begin + v), (uint32_t)(r0 + 4 * q4 + rr), (uint32_t)c, pa.seed_lo, pa.seed_hi, z);
            res[0][rr] = noisy_count(res[0][rr], var[0][rr], z[0]);
            res[1][rr] = noisy_count(res[1][rr], var[1][rr], z[1]);
          }
        }
        if constexpr (!staged) {                   // the round's stores (what detect_store does for the noise-free kernel)
          if (round_live) {
            const bool vec4 = a.layout == 1 && (a.g.n_rows & 3) == 0 && valid[3];
#pragma unroll
            for (int s = 0; s < 2; ++s)
              if (s < a.n_spectra) {
                if (vec4) {
                  *reinterpret_cast<float4*>(a.counts + rays[0] + s * sstride_n) = make_float4(res[s][0], res[s][1], res[s][2], res[s][3]);
                  if (a.sino_log)
                    *reinterpret_cast<float4*>(a.sino_log + rays[0] + s * sstride_n) =
                        make_float4(log_ratio(a.air[s], res[s][0]), log_ratio(a.air[s], res[s][1]),
                                    log_ratio(a.air[s], res[s][2]), log_ratio(a.air[s], res[s][3]));
                } else {
#pragma unroll
                  for (int rr = 0; rr < 4; ++rr)
                    if (valid[rr]) {
                      a.counts[rays[rr] + s * sstride_n] = res[s][rr];
                      if (a.sino_log) a.sino_log[rays[rr] + s * sstride_n] = log_ratio(a.air[s], res[s][rr]);
                    }
                }
              }
          }
        }
      } else if constexpr (staged) {
        detect_store<NM, 4, false>(L, a, mu, w, w2, rays, valid, bm, &air_cache, &res);
      } else {
        detect_store<NM, 4>(L, a, mu, w, w2, rays, valid, bm, &air_cache);
      }
    }
    // STAGED: the round's results go to LDS right away (plane s, float index 16 lane + row): no registers are held for
    // them, and the (corr[0][row], corr[1][row]) register pairs of the traversal's v_pk_add_f32 stay undisturbed
    if constexpr (staged) {
      if (round_live) {
        float* stage_w = reinterpret_cast<float*>(lds_lists) + lane * 16 + 4 * q4;
#pragma unroll
        for (int s = 0; s < kResSlots; ++s)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) stage_w[s * 1024 + rr] = res[s][rr];
      }
    }
#pragma unroll
    for (int m = 0; m < NM - 1; ++m)
#pragma unroll
      for (int row = 0; row < 12; ++row) corr[m][row] = corr[m][row + 4];
  }
  if constexpr (!staged) return;
  // ---- whole-line stores: the rounds left the 16 results of lane l and spectrum s in LDS at plane s, float index
  // 16 l + row; store instruction k takes the 16-byte piece 64 k + lane = (lane l' = piece / 4, rows 4 (piece % 4) ...):
  // consecutive lanes write consecutive 16 bytes
  float* stage = reinterpret_cast<float*>(lds_lists);
  const int lpp_shift = __builtin_ctz((unsigned)lpp);
  const int c_base = c - g;                                    // first channel of the wave
  const size_t sstride = (size_t)a.n_local_views * a.g.n_rows * a.g.n_channels;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int s = 0; s < kResSlots; ++s) {
    if (s >= a.n_spectra) break;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int piece = k * 64 + lane, l2 = piece >> 2, c2 = piece & 3;
      const int g2 = l2 >> lpp_shift, li2 = l2 - (g2 << lpp_shift);
      const int row = (zc * 64 + li2) * 16 + 4 * c2, chan = c_base + g2;
      if (chan < a.g.n_channels && row < a.g.n_rows) {
        const float4 x = *reinterpret_cast<const float4*>(stage + s * 1024 + piece * 4);
        const size_t at = s * sstride + ((size_t)v * a.g.n_channels + chan) * a.g.n_rows + row;
        *reinterpret_cast<float4*>(a.counts + at) = x;
        if (a.sino_log)
          *reinterpret_cast<float4*>(a.sino_log + at) = make_float4(log_ratio(a.air[s], x.x), log_ratio(a.air[s], x.y),
                                                                    log_ratio(a.air[s], x.z), log_ratio(a.air[s], x.w));
      }
    }
  }
}

// vol_zf [ny][nx][nz] (one byte per voxel, ids < 4) -> vol_z2 [ny][nx][nz / 4] (2 bits per voxel, z fastest)
__global__ __launch_bounds__(256) void pack2_kernel(const uint8_t* __restrict__ vol_zf, size_t n_bytes_out,
                                                    uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_bytes_out) return;
  const uint32_t x = *reinterpret_cast<const uint32_t*>(vol_zf + 4 * i);
  out[i] = (uint8_t)((x & 3u) | (((x >> 8) & 3u) << 2) | (((x >> 16) & 3u) << 4) | (((x >> 24) & 3u) << 6));
}

// Material groups on the packed volume: ids 3g+1..3g+3 -> codes 1..3, everything else 0 (group_codes_kernel of
// siddon.hip), packed four voxels per byte: out [n_groups][n_voxels / 4]
__global__ __launch_bounds__(256) void group_codes_pack2_kernel(const uint8_t* __restrict__ vol_zf, size_t n_bytes_out,
                                                                int n_groups, uint8_t* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_bytes_out) return;
  const uint32_t x = *reinterpret_cast<const uint32_t*>(vol_zf + 4 * i);
  for (int g = 0; g < n_groups; ++g) {
    uint32_t y = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t rel = ((x >> (8 * b)) & 0xFFu) - 3u * g;      // 1..3 inside the group
      y |= ((rel >= 1u && rel <= 3u) ? rel : 0u) << (2 * b);
    }
    out[(size_t)g * n_bytes_out + i] = (uint8_t)y;
  }
}

// what both entry points check; fills the launch shape
static int packed_shape(const dexct_fan_geom* geom, int32_t view_begin, int32_t view_end, int32_t layout, PackedArgs& pa,
                        size_t& nblk, size_t& lds) {
  if (view_begin < 0 || view_end > geom->n_views || view_end <= view_begin) return DEXCT_EINVAL;
  if (geom->n_rows < 1) return DEXCT_EINVAL;
  if (geom->z_first < 0 || geom->z_first + geom->n_rows > geom->nz) return DEXCT_EINVAL;
  if (geom->nz % 16 != 0 || geom->z_first % 16 != 0) return DEXCT_EINVAL;
  if ((uint64_t)geom->nx * geom->ny * (geom->nz / 4) > 0xEFFF0000ull) return DEXCT_ERANGE;     // below kOob
  if (geom->nx > 2047 || geom->ny > 2047) return DEXCT_ERANGE;             // 11-bit sliced counters: one count per slab
  if (layout != 0 && layout != 1) return DEXCT_EINVAL;
  if (geom->n_rows > 65535 || view_end - view_begin > 65535) return DEXCT_ERANGE;
  // a (view, channel) pair takes ceil(n_rows / 16) lanes: 16 or 32 lanes per pair (4 or 2 pairs per wave) or whole
  // waves; lanes past the last row idle (the caller decides whether that is still worth it)
  const int lanes = (geom->n_rows + 15) / 16;
  pa.lanes_per_pair = lanes > 32 ? 64 : (lanes > 16 ? 32 : 16);
  pa.n_zchunks = lanes > 64 ? (lanes + 63) / 64 : 1;
  pa.view_tile = 16;                 // (8 until round 5: 16 measured 2 - 4 % faster on every stacked-fan configuration, tools/probes/p16_knobs.py)
  pa.det_masks = 1;
  if (const char* e = getenv("DEXCT_DET_MASKS")) pa.det_masks = atoi(e) != 0;
  pa.staged_store = 1;
  if (const char* e = getenv("DEXCT_P16_STAGED")) pa.staged_store = atoi(e) != 0;
  if (const char* e = getenv("DEXCT_VIEW_TILE")) { const int t = atoi(e); if (t >= 1 && t <= 4096) pa.view_tile = t; }
  const int n_pairs = 64 / pa.lanes_per_pair;
  nblk = (size_t)(view_end - view_begin) * ((geom->n_channels + n_pairs - 1) / n_pairs) * pa.n_zchunks;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  lds = (size_t)n_pairs * kP16Super * (sizeof(CrossRec) + sizeof(uint32_t));
  if (lds < 2 * 64 * 64) lds = 2 * 64 * 64;   // the staged store of the results: 64 bytes per lane and spectrum
  return DEXCT_OK;
}

}  // namespace dexct

using namespace dexct;

extern "C" {

int dexct_volume_pack2(const uint8_t* vol_zf, int64_t n_voxels, uint8_t* vol_z2, void* stream) {
  if (!vol_zf || !vol_z2 || n_voxels <= 0 || (n_voxels & 3)) return DEXCT_EINVAL;
  const size_t nb = (size_t)n_voxels / 4;
  const size_t nblk = (nb + 255) / 256;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  hipLaunchKernelGGL(pack2_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), vol_zf, nb, vol_z2);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_volume_groups_pack2(const uint8_t* vol_zf, int64_t n_voxels, int32_t n_materials, uint8_t* codes2, void* stream) {
  if (!vol_zf || !codes2 || n_voxels <= 0 || (n_voxels & 3)) return DEXCT_EINVAL;
  if (n_materials < 2 || n_materials > DEXCT_MAX_MATERIALS) return DEXCT_ERANGE;
  const int n_groups = (n_materials - 1 + 2) / 3;
  const size_t nb = (size_t)n_voxels / 4;
  const size_t nblk = (nb + 255) / 256;
  if (nblk > 0x7FFFFFFFull) return DEXCT_ERANGE;
  hipLaunchKernelGGL(group_codes_pack2_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), vol_zf, nb, n_groups,
                     codes2);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_siddon_project_packed(const dexct_fan_geom* geom, const dexct_ray_plan* plan, int32_t view_begin,
                                int32_t view_end, const uint8_t* vol_z2, int32_t n_materials, int32_t n_energies,
                                int32_t n_spectra, const float* mu, const float* weights, float* counts, float* pathlen,
                                int32_t layout, const dexct_log_out* log_out, const float* weights2, float* variance,
                                const dexct_noise* noise, void* stream) {
  if (!geom || !plan || !vol_z2 || !mu || !weights || !counts) return DEXCT_EINVAL;
  if (n_energies < 1 || n_spectra < 1) return DEXCT_EINVAL;
  if (n_materials < 2 || n_materials > 4 || n_spectra > DEXCT_MAX_SPECTRA) return DEXCT_ERANGE;     // ids 0..3
  // quantum noise: weights2 switches the variance on; it then needs somewhere to go - the optional output, the sample, or both
  const bool sample = noise && noise->sample;
  const bool noisy = weights2 != nullptr;
  if (!noisy && (variance || sample)) return DEXCT_EINVAL;
  if (noisy && !variance && !sample) return DEXCT_EINVAL;
  if (noisy && n_spectra > 2) return DEXCT_ERANGE;              // (the fused variance runs in the two-slot detection)
  PackedArgs pa;
  size_t nblk, lds;
  const int rc = packed_shape(geom, view_begin, view_end, layout, pa, nblk, lds);
  if (rc != DEXCT_OK) return rc;
  ProjArgs& a = pa.a;
  a.g = *geom;
  a.plan = plan;
  a.vol_yx = nullptr;
  a.vol_xy = nullptr;
  a.vol_zf = nullptr;
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
  // the log of a noisy sinogram is the log of the SAMPLED counts: the kernel writes it only when it draws the sample itself
  if (set_log_out(a, log_out, (noisy && !sample) ? weights2 : nullptr) != DEXCT_OK) return DEXCT_EINVAL;
  a.view_tile = pa.view_tile;
  pa.vol_z2 = vol_z2;
  pa.view_begin = view_begin;
  pa.sample = sample ? 1 : 0;
  pa.seed_lo = sample ? (uint32_t)noise->seed : 0u;
  pa.seed_hi = sample ? (uint32_t)(noise->seed >> 32) : 0u;
  hipStream_t st = as_stream(stream);
  int minw = 4;
  if (const char* e = getenv("DEXCT_P16_MINW")) minw = atoi(e);      // tuning knob
  const float* none = nullptr;
  // whole-line stores through LDS (STAGED) wherever the output allows them: row-fastest layout, whole groups of 4 rows,
  // one or two spectra (what the stage holds)
  const bool staged = pa.staged_store && layout == 1 && geom->n_rows % 4 == 0 && n_spectra <= 2;
  if (noisy) {
    if (n_materials == 2 && staged)
      hipLaunchKernelGGL((rows16_kernel<2, 4, false, true, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, weights2);
    else if (n_materials == 2)
      hipLaunchKernelGGL((rows16_kernel<2, 4, false, false, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, weights2);
    else if (n_materials == 4 && staged)
      hipLaunchKernelGGL((rows16_kernel<4, 3, false, true, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, weights2);
    else if (n_materials == 4)
      hipLaunchKernelGGL((rows16_kernel<4, 3, false, false, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, weights2);
    else if (staged)
      hipLaunchKernelGGL((rows16_kernel<3, 4, false, true, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, weights2);
    else
      hipLaunchKernelGGL((rows16_kernel<3, 4, false, false, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, weights2);
    DEXCT_LAUNCH_CHECK();
    return DEXCT_OK;
  }
  if (n_materials == 2 && staged)
    hipLaunchKernelGGL((rows16_kernel<2, 4, false, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, none);
  else if (n_materials == 2)
    hipLaunchKernelGGL((rows16_kernel<2>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, none);
  else if (n_materials == 4 && staged)
    hipLaunchKernelGGL((rows16_kernel<4, 3, false, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, none);
  else if (n_materials == 4)
    hipLaunchKernelGGL((rows16_kernel<4, 3>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, none);
  else if (staged && minw == 4)
    hipLaunchKernelGGL((rows16_kernel<3, 4, false, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, none);
  else if (staged && minw == 5)
    hipLaunchKernelGGL((rows16_kernel<3, 5, false, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, none);
  else if (minw == 5)
    hipLaunchKernelGGL((rows16_kernel<3, 5>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, none);
  else if (minw == 6)
    hipLaunchKernelGGL((rows16_kernel<3, 6>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, none);
  else if (minw == 8)
    hipLaunchKernelGGL((rows16_kernel<3, 8>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, none);
  else
    hipLaunchKernelGGL((rows16_kernel<3>), dim3((unsigned)nblk), dim3(64), lds, st, pa, mu, weights, none);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_siddon_project_grouped_packed(const dexct_fan_geom* geom, const dexct_ray_plan* plan, int32_t view_begin,
                                        int32_t view_end, const uint8_t* codes2, int32_t n_materials, int32_t n_energies,
                                        int32_t n_spectra, const float* mu, const float* weights, float* counts,
                                        float* pathlen, float* acc_scratch, int32_t layout, const float* weights2,
                                        float* variance, const dexct_log_out* log_out, const dexct_noise* noise, void* stream) {
  if (!geom || !plan || !codes2 || !mu || !weights || !counts || !acc_scratch) return DEXCT_EINVAL;
  if (n_materials < 2 || n_energies < 1 || n_spectra < 1) return DEXCT_EINVAL;
  if (n_materials > DEXCT_MAX_MATERIALS || n_spectra > DEXCT_MAX_SPECTRA) return DEXCT_ERANGE;
  PackedArgs pa;
  size_t nblk, lds;
  const int rc = packed_shape(geom, view_begin, view_end, layout, pa, nblk, lds);
  if (rc != DEXCT_OK) return rc;
  ProjArgs& a = pa.a;
  a.g = *geom;
  a.plan = plan;
  a.vol_yx = nullptr;
  a.vol_xy = nullptr;
  a.vol_zf = nullptr;
  a.n_local_views = view_end - view_begin;
  a.n_materials = n_materials;
  a.n_energies = n_energies;
  a.n_spectra = n_spectra;
  a.counts = counts;
  a.pathlen = pathlen;
  a.variance = variance;
  a.acc_out = acc_scratch;
  a.mat_base = 0;
  a.layout = layout;
  { const int nrc = set_noise(a, view_begin, weights2, variance, noise); if (nrc != DEXCT_OK) return nrc; }
  if (set_log_out(a, log_out, a.sample ? nullptr : variance) != DEXCT_OK) return DEXCT_EINVAL;
  a.view_tile = pa.view_tile;
  pa.view_begin = view_begin;
  pa.sample = 0;
  pa.seed_lo = pa.seed_hi = 0u;
  hipStream_t st = as_stream(stream);
  const size_t group_bytes = (size_t)geom->nx * geom->ny * (geom->nz / 4);
  const int n_groups = (n_materials - 1 + 2) / 3;
  const float* none = nullptr;
  for (int g = 0; g < n_groups; ++g) {
    pa.vol_z2 = codes2 + (size_t)g * group_bytes;
    a.mat_base = 3 * g;
    const int left = n_materials - 1 - 3 * g;           // materials in this group: 1..3
    if (left >= 3)
      hipLaunchKernelGGL((rows16_kernel<4, 3, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, none, none, none);
    else if (left == 2)
      hipLaunchKernelGGL((rows16_kernel<3, 4, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, none, none, none);
    else
      hipLaunchKernelGGL((rows16_kernel<2, 4, true>), dim3((unsigned)nblk), dim3(64), lds, st, pa, none, none, none);
    DEXCT_LAUNCH_CHECK();
  }
  const Tables t{mu, weights, weights2};
  return launch_detect_any(a, t, st);
}

}  // extern "C"
