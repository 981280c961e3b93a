// Volume layouts and per-(view, channel) traversal plans.
//
// The plan is the float64 part of the projector: source / detector placement of the fan-beam
// geometry (input/params.txt:18-27 of the reference), dominant in-plane axis, slope and
// intercept of the minor coordinate, quantised once to 40-bit fixed point.  Everything the
// traversal kernels do afterwards is integer or float32 arithmetic on these numbers, which is
// what makes voxel-index sequences reproducible bit for bit (oracle/dexct_oracle.c orc_plan_one
// restates the same operations in the same order; build with -ffp-contract=off).
#include "common.h"
#include <dlfcn.h>
#include <sys/mman.h>
#include <thread>
#include <vector>

namespace dexct {

thread_local int g_last_hip_error = 0;

__global__ __launch_bounds__(256) void fan_plan_kernel(dexct_fan_geom g, const double* __restrict__ view_cs,
                                                       const double* __restrict__ chan_cs, int view_begin,
                                                       int n_local_views, dexct_ray_plan* __restrict__ plan) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int v = blockIdx.y;
  if (c >= g.n_channels || v >= n_local_views) return;
  const int view = view_begin + v;
  const double cb = view_cs[2 * view], sb = view_cs[2 * view + 1];
  const double cg = chan_cs[2 * c], sg = chan_cs[2 * c + 1];
  const double sx = g.sid * cb, sy = g.sid * sb;
  const double ex = -(cb * cg - sb * sg), ey = -(sb * cg + cb * sg);
  const double sxv = sx / g.dx + 0.5 * g.nx, syv = sy / g.dy + 0.5 * g.ny;
  const double exv = ex / g.dx, eyv = ey / g.dy;
  const int axis = fabs(exv) >= fabs(eyv) ? 0 : 1;
  double su, sv_, eu, ev;
  int nu, nv;
  if (axis == 0) { su = sxv; sv_ = syv; eu = exv; ev = eyv; nu = g.nx; nv = g.ny; }
  else           { su = syv; sv_ = sxv; eu = eyv; ev = exv; nu = g.ny; nv = g.nx; }
  const double kOne = 1099511627776.0;  // 2^40
  const double slope = ev / eu;
  const double v0 = sv_ - su * slope;
  const long long SV = llrint(slope * kOne);
  const long long V0 = llrint(v0 * kOne);
  const double sq = (double)SV / kOne;
  const double v0q = (double)V0 / kOne;
  double ulo = 0.0, uhi = (double)nu;
  bool miss = false;
  if (SV > 0) {
    ulo = fmax(ulo, (0.0 - v0q) / sq);
    uhi = fmin(uhi, ((double)nv - v0q) / sq);
  } else if (SV < 0) {
    ulo = fmax(ulo, ((double)nv - v0q) / sq);
    uhi = fmin(uhi, (0.0 - v0q) / sq);
  } else if (v0q < 0.0 || v0q >= (double)nv) {
    miss = true;
  }
  if (!(uhi > ulo)) miss = true;
  double inv = 16777216.0;
  if (SV != 0) inv = fmin(kOne / fabs((double)SV), 16777216.0);
  dexct_ray_plan p;
  p.V0 = V0;
  p.SV = SV;
  p.kf = (float)(inv * (1.0 / 4294967296.0));
  p.len_per_u = (float)(1.0 / fabs(eu));
  p.flags = (uint32_t)axis | (SV > 0 ? 2u : 0u);
  if (miss) {
    p.i_first = 0;
    p.n_slabs = 0;
    p.chord_u = 0.0f;
  } else {
    int i0 = (int)floor(ulo), i1 = (int)ceil(uhi) - 1;
    if (i0 < 0) i0 = 0;
    if (i1 > nu - 1) i1 = nu - 1;
    p.i_first = i0;
    p.n_slabs = i1 - i0 + 1;
    p.chord_u = (float)(uhi - ulo);
  }
  plan[(size_t)v * g.n_channels + c] = p;
}

// [nz][ny][nx] -> [nz][nx][ny] through a 64x64 LDS tile (coalesced on both sides), and
// optionally -> [ny][nx][nz] (z fastest).
__global__ __launch_bounds__(256) void transpose_xy_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                           int nx, int ny) {
  __shared__ uint8_t tile[64][65];
  const size_t slice = (size_t)blockIdx.z * nx * ny;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
  for (int r = ty; r < 64; r += 4) {
    int x = x0 + tx, y = y0 + r;
    if (x < nx && y < ny) tile[r][tx] = src[slice + (size_t)y * nx + x];
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    int y = y0 + tx, x = x0 + r;
    if (x < nx && y < ny) dst[slice + (size_t)x * ny + y] = tile[tx][r];
  }
}

// [nz][ny*nx] -> [ny*nx][nz]: a 2-D transpose with rows = z, columns = in-plane index.
__global__ __launch_bounds__(256) void transpose_z_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                          size_t nxy, int nz) {
  __shared__ uint8_t tile[64][65];
  const size_t p0 = (size_t)blockIdx.x * 64;
  const int z0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int r = ty; r < 64; r += 4) {
    size_t p = p0 + tx;
    int z = z0 + r;
    if (p < nxy && z < nz) tile[r][tx] = src[(size_t)z * nxy + p];
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    size_t p = p0 + r;
    int z = z0 + tx;
    if (p < nxy && z < nz) dst[p * nz + z] = tile[tx][r];
  }
}

// [batch][rows][cols] -> [batch][cols][rows], 32x32 tiles through LDS (coalesced both ways).
template <typename T>
__global__ __launch_bounds__(256) void transpose_batched_kernel(const T* __restrict__ src, T* __restrict__ dst,
                                                                int rows, int cols) {
  __shared__ T tile[32][33];
  const size_t base = (size_t)blockIdx.z * rows * cols;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    if (r < rows && c < cols) tile[k][tx] = src[base + (size_t)r * cols + c];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (r < rows && c < cols) dst[base + (size_t)c * rows + r] = tile[tx][k];
  }
}

template <typename T>
static int launch_transpose(const void* src, void* dst, int64_t batch, int rows, int cols, hipStream_t st) {
  // gridDim.z is limited to 65535: walk the batch in slices
  for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
    const int nb = (int)((batch - b0) < 65535 ? (batch - b0) : 65535);
    dim3 grid((cols + 31) / 32, (rows + 31) / 32, nb);
    hipLaunchKernelGGL(transpose_batched_kernel<T>, grid, dim3(256), 0, st,
                       reinterpret_cast<const T*>(src) + (size_t)b0 * rows * cols,
                       reinterpret_cast<T*>(dst) + (size_t)b0 * rows * cols, rows, cols);
    DEXCT_LAUNCH_CHECK();
  }
  return DEXCT_OK;
}

}  // namespace dexct

namespace dexct {

// Which of the 256 possible ids a uint8 volume holds (counts saturate nowhere: uint32 per workgroup, uint64 in all).
__global__ __launch_bounds__(256) void volume_ids_kernel(const uint8_t* __restrict__ vol, size_t n, unsigned long long* __restrict__ counts) {
  __shared__ unsigned int h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const size_t n16 = n / 16;
  const uint4* __restrict__ v16 = reinterpret_cast<const uint4*>(vol);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 q = v16[i];
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      atomicAdd(&h[w[k] & 255u], 1u);
      atomicAdd(&h[(w[k] >> 8) & 255u], 1u);
      atomicAdd(&h[(w[k] >> 16) & 255u], 1u);
      atomicAdd(&h[w[k] >> 24], 1u);
    }
  }
  if (blockIdx.x == 0)
    for (size_t i = n16 * 16 + threadIdx.x; i < n; i += 256) atomicAdd(&h[vol[i]], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(counts + threadIdx.x, (unsigned long long)h[threadIdx.x]);
}

struct Lut256 { uint8_t v[256]; };

__global__ __launch_bounds__(256) void volume_remap_kernel(uint8_t* __restrict__ vol, size_t n, Lut256 lut) {
  __shared__ uint8_t l[256];
  l[threadIdx.x] = lut.v[threadIdx.x];
  __syncthreads();
  const size_t n4 = n / 4;
  uint32_t* __restrict__ v4 = reinterpret_cast<uint32_t*>(vol);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const uint32_t w = v4[i];
    v4[i] = (uint32_t)l[w & 255u] | ((uint32_t)l[(w >> 8) & 255u] << 8) | ((uint32_t)l[(w >> 16) & 255u] << 16) |
            ((uint32_t)l[w >> 24] << 24);
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) vol[i] = l[vol[i]];
}

}  // namespace dexct

using namespace dexct;

extern "C" {

const char* dexct_strerror(int code) {
  switch (code) {
    case DEXCT_OK: return "ok";
    case DEXCT_EINVAL: return "invalid argument";
    case DEXCT_ERANGE: return "size out of supported range";
    case DEXCT_EHIP: return "HIP runtime error (see dexct_last_hip_error)";
    case DEXCT_ERCCL: return "RCCL library not found, or the collective failed (see dexct_last_hip_error)";
    default: return "unknown dexct error";
  }
}

int dexct_abi_version(void) { return DEXCT_ABI_VERSION; }

// The one data-path collective of the sharded scan: every rank contributes its view shard of the sinogram and
// receives all of them (ncclAllGather over xGMI).  The library does not link RCCL: the symbol is taken from the
// copy the process has already loaded (the host framework's, e.g. torch's) or, failing that, from librccl.so.1 -
// two RCCL instances in one process must not happen.
typedef int (*dexct_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*dexct_send_fn)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*dexct_recv_fn)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*dexct_group_fn)(void);

struct RcclSymbols {
  dexct_allgather_fn all_gather = nullptr;
  dexct_send_fn send = nullptr;
  dexct_recv_fn recv = nullptr;
  dexct_group_fn group_start = nullptr, group_end = nullptr;
};

static const RcclSymbols& rccl() {
  static const RcclSymbols sym = [] {
    RcclSymbols r;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    for (const char* n : names)
      if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) {
      r.all_gather = reinterpret_cast<dexct_allgather_fn>(dlsym(h, "ncclAllGather"));
      r.send = reinterpret_cast<dexct_send_fn>(dlsym(h, "ncclSend"));
      r.recv = reinterpret_cast<dexct_recv_fn>(dlsym(h, "ncclRecv"));
      r.group_start = reinterpret_cast<dexct_group_fn>(dlsym(h, "ncclGroupStart"));
      r.group_end = reinterpret_cast<dexct_group_fn>(dlsym(h, "ncclGroupEnd"));
    }
    return r;
  }();
  return sym;
}

int dexct_sino_allgather(const float* local, float* gathered, int64_t count_per_rank, void* rccl_comm, void* stream) {
  if (!local || !gathered || !rccl_comm || count_per_rank <= 0) return DEXCT_EINVAL;
  dexct_allgather_fn fn = rccl().all_gather;
  if (!fn) return DEXCT_ERCCL;
  const int rc = fn(local, gathered, (size_t)count_per_rank, /* ncclFloat32 */ 7, rccl_comm, as_stream(stream));
  if (rc != 0) {
    g_last_hip_error = rc;          // ncclResult_t of the failed call
    return DEXCT_ERCCL;
  }
  return DEXCT_OK;
}

// The same assembly as point-to-point transfers, one per peer, all inside ONE RCCL group (they run concurrently, each over
// the direct xGMI link of its pair of GPUs): what the step costs does not depend on which algorithm RCCL picks for an
// all-gather (a ring is bound by one link).  Ragged shards need no padding.
int dexct_sino_gather(const float* local, float* gathered, const int64_t* counts, const int64_t* offsets, int32_t rank,
                      int32_t world, int32_t root, void* rccl_comm, void* stream) {
  if (!local || !counts || !offsets || !rccl_comm || world < 1 || rank < 0 || rank >= world || root < -1 || root >= world)
    return DEXCT_EINVAL;
  const bool receives = root < 0 || root == rank;
  if (receives && !gathered) return DEXCT_EINVAL;
  for (int k = 0; k < world; ++k)
    if (counts[k] < 0 || offsets[k] < 0) return DEXCT_EINVAL;
  const RcclSymbols& r = rccl();
  if (!r.send || !r.recv || !r.group_start || !r.group_end) return DEXCT_ERCCL;
  hipStream_t st = as_stream(stream);
  if (receives && counts[rank] > 0 && gathered + offsets[rank] != local)         // this rank's own shard: a device copy
    DEXCT_HIP_TRY(hipMemcpyAsync(gathered + offsets[rank], local, (size_t)counts[rank] * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (world == 1) return DEXCT_OK;
  int rc = r.group_start();
  for (int k = 0; k < world && rc == 0; ++k) {
    if (k == rank) continue;
    if ((root < 0 || root == k) && counts[rank] > 0)                             // my shard goes to peer k
      rc = r.send(local, (size_t)counts[rank], /* ncclFloat32 */ 7, k, rccl_comm, st);
    if (rc == 0 && receives && counts[k] > 0)                                    // peer k's shard lands where it belongs
      rc = r.recv(gathered + offsets[k], (size_t)counts[k], 7, k, rccl_comm, st);
  }
  const int rc_end = r.group_end();                                              // (always closed, also after a failed call)
  if (rc == 0) rc = rc_end;
  if (rc != 0) {
    g_last_hip_error = rc;
    return DEXCT_ERCCL;
  }
  return DEXCT_OK;
}

int dexct_last_hip_error(void) { return g_last_hip_error; }

// Page-locking a piece of the CALLER'S host array and copying into it (the host boundary of get_basismat_sinos: the result
// array of a first call is locked piece by piece by a helper thread while the kernels and the copies of the pieces before it
// run, instead of in one 0.3 s allocation of page-locked memory in front of everything).  `device`: the HIP device of the
// calling process - a helper thread has no current device of its own.
int dexct_host_pin(void* host, int64_t n_bytes, int32_t device) {
  if (!host || n_bytes <= 0 || device < 0) return DEXCT_EINVAL;
  DEXCT_HIP_TRY(hipSetDevice(device));
  hipError_t e = hipHostRegister(host, (size_t)n_bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {                  // (memory that is locked already, a locked-memory limit: the caller copies without)
    (void)hipGetLastError();              // do not leave the refusal behind for the next launch check of this thread
    return ::dexct::hip_fail(e);
  }
  return DEXCT_OK;
}

int dexct_host_touch(void* host, int64_t n_bytes, int32_t threads) {
  // make the pages of [host, host + n_bytes) resident, `threads` parts at a time: a page fault of fresh memory (zeroing included)
  // is what page-locking it costs; memory that is resident locks in microseconds (tools/probes/pin_resident.py)
  if (!host || n_bytes <= 0 || threads < 1 || threads > 64) return DEXCT_EINVAL;
  const uintptr_t page = 4096, huge = (uintptr_t)1 << 21;
  const uintptr_t first = reinterpret_cast<uintptr_t>(host), end = first + (uintptr_t)n_bytes;
  const uintptr_t lo = first / page * page;
  const uintptr_t hi = (end + page - 1) / page * page;
  auto part = [first, end](uintptr_t b, uintptr_t e) {
    if (e <= b) return;
#ifdef MADV_POPULATE_WRITE
    if (madvise(reinterpret_cast<void*>(b), e - b, MADV_POPULATE_WRITE) == 0) return;
#endif
    // (kernels before 5.14: a read-modify-write of one byte per page - of a byte INSIDE the caller's block: the first and the
    // last page may also hold other objects of the heap, and a byte of theirs rewritten with a stale value is a corrupted heap)
    for (uintptr_t a = b; a < e; a += 4096) {
      const uintptr_t at = a < first ? first : a;
      if (at >= end) break;
      volatile unsigned char* q = reinterpret_cast<volatile unsigned char*>(at);
      *q = *q;
    }
  };
  std::vector<uintptr_t> cut{lo};
  for (int k = 1; k < threads; ++k) {
    uintptr_t c = (lo + (hi - lo) / threads * k) / huge * huge;
    if (c > cut.back() && c < hi) cut.push_back(c);
  }
  cut.push_back(hi);
  std::vector<std::thread> pool;
  for (size_t k = 1; k + 1 < cut.size(); ++k) pool.emplace_back(part, cut[k], cut[k + 1]);
  part(cut[0], cut[1]);
  for (auto& t : pool) t.join();
  return DEXCT_OK;
}

int dexct_host_unpin(void* host, int32_t device) {
  if (!host || device < 0) return DEXCT_EINVAL;
  DEXCT_HIP_TRY(hipSetDevice(device));
  DEXCT_HIP_TRY(hipHostUnregister(host));
  return DEXCT_OK;
}

int dexct_download(void* host, const void* device_src, int64_t n_bytes, void* stream) {
  if (!host || !device_src || n_bytes <= 0) return DEXCT_EINVAL;
  DEXCT_HIP_TRY(hipMemcpyAsync(host, device_src, (size_t)n_bytes, hipMemcpyDeviceToHost, as_stream(stream)));
  return DEXCT_OK;
}

int dexct_volume_ids(const uint8_t* vol, int64_t n_voxels, uint64_t* counts256, void* stream) {
  if (!vol || !counts256 || n_voxels <= 0) return DEXCT_EINVAL;
  if (reinterpret_cast<uintptr_t>(vol) & 15u) return DEXCT_EINVAL;
  hipStream_t st = as_stream(stream);
  DEXCT_HIP_TRY(hipMemsetAsync(counts256, 0, 256 * sizeof(uint64_t), st));
  size_t nblk = ((size_t)n_voxels / 16 + 255) / 256;
  nblk = nblk < 1 ? 1 : (nblk > 4096 ? 4096 : nblk);
  hipLaunchKernelGGL(volume_ids_kernel, dim3((unsigned)nblk), dim3(256), 0, st, vol, (size_t)n_voxels,
                     reinterpret_cast<unsigned long long*>(counts256));
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_volume_remap(uint8_t* vol, int64_t n_voxels, const uint8_t* lut256, void* stream) {
  if (!vol || !lut256 || n_voxels <= 0) return DEXCT_EINVAL;
  if (reinterpret_cast<uintptr_t>(vol) & 3u) return DEXCT_EINVAL;
  Lut256 lut;
  for (int k = 0; k < 256; ++k) lut.v[k] = lut256[k];
  size_t nblk = ((size_t)n_voxels / 4 + 255) / 256;
  nblk = nblk < 1 ? 1 : (nblk > 8192 ? 8192 : nblk);
  hipLaunchKernelGGL(volume_remap_kernel, dim3((unsigned)nblk), dim3(256), 0, as_stream(stream), vol, (size_t)n_voxels, lut);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

int dexct_volume_layouts(const uint8_t* vol, int32_t nx, int32_t ny, int32_t nz, uint8_t* vol_xy, uint8_t* vol_zf,
                         void* stream) {
  if (!vol || nx <= 0 || ny <= 0 || nz <= 0 || (!vol_xy && !vol_zf)) return DEXCT_EINVAL;
  if (nz > 65535 || (ny + 63) / 64 > 65535) return DEXCT_ERANGE;
  hipStream_t st = as_stream(stream);
  if (vol_xy) {
    dim3 grid((nx + 63) / 64, (ny + 63) / 64, nz);
    hipLaunchKernelGGL(transpose_xy_kernel, grid, dim3(256), 0, st, vol, vol_xy, nx, ny);
    DEXCT_LAUNCH_CHECK();
  }
  if (vol_zf) {
    size_t nxy = (size_t)nx * ny;
    dim3 grid((unsigned)((nxy + 63) / 64), (nz + 63) / 64, 1);
    hipLaunchKernelGGL(transpose_z_kernel, grid, dim3(256), 0, st, vol, vol_zf, nxy, nz);
    DEXCT_LAUNCH_CHECK();
  }
  return DEXCT_OK;
}

int dexct_transpose_batched(const void* src, void* dst, int64_t batch, int32_t rows, int32_t cols, int32_t elem_bytes,
                            void* stream) {
  if (!src || !dst || batch <= 0 || rows <= 0 || cols <= 0) return DEXCT_EINVAL;
  if ((rows + 31) / 32 > 65535) return DEXCT_ERANGE;
  hipStream_t st = as_stream(stream);
  switch (elem_bytes) {
    case 4: return launch_transpose<float>(src, dst, batch, rows, cols, st);
    case 8: return launch_transpose<double>(src, dst, batch, rows, cols, st);
    case 16: return launch_transpose<double2>(src, dst, batch, rows, cols, st);
    default: return DEXCT_EINVAL;
  }
}

int dexct_fan_plan(const dexct_fan_geom* geom, const double* view_cs, const double* chan_cs, int32_t view_begin,
                   int32_t view_end, dexct_ray_plan* plan, void* stream) {
  if (!geom || !view_cs || !chan_cs || !plan) return DEXCT_EINVAL;
  if (view_begin < 0 || view_end > geom->n_views || view_end <= view_begin) return DEXCT_EINVAL;
  if (geom->n_channels <= 0 || geom->nx <= 0 || geom->ny <= 0) return DEXCT_EINVAL;
  if (geom->nx > 8192 || geom->ny > 8192) return DEXCT_ERANGE;  // fixed-point range: 13 + 40 bits
  const int nv = view_end - view_begin;
  if (nv > 65535) return DEXCT_ERANGE;
  dim3 grid((geom->n_channels + 255) / 256, nv, 1);
  hipLaunchKernelGGL(fan_plan_kernel, grid, dim3(256), 0, as_stream(stream), *geom, view_cs, chan_cs, view_begin,
                     nv, plan);
  DEXCT_LAUNCH_CHECK();
  return DEXCT_OK;
}

}  // extern "C"
