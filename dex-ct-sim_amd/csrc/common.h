// Shared helpers of the gfx950 kernels (internal; the public surface is include/dexct.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dexct.h"

namespace dexct {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) {
  g_last_hip_error = static_cast<int>(e);
  return DEXCT_EHIP;
}

#define DEXCT_HIP_TRY(expr)                           \
  do {                                                \
    hipError_t e_ = (expr);                           \
    if (e_ != hipSuccess) return ::dexct::hip_fail(e_); \
  } while (0)

#define DEXCT_LAUNCH_CHECK() DEXCT_HIP_TRY(hipGetLastError())

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// More than 64 KB of dynamic LDS for one workgroup (gfx950: 160 KB): say so to the runtime before the launch.
#define DEXCT_ALLOW_LDS(kernel, bytes)                                                                              \
  do {                                                                                                               \
    if ((size_t)(bytes) > 160u * 1024u) return DEXCT_ERANGE;                                                          \
    if ((size_t)(bytes) > 64u * 1024u)                                                                               \
      DEXCT_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                        (int)(bytes)));                                                              \
  } while (0)

// Kernels that keep per-material sums in per-lane LDS columns run 128 lanes per workgroup up to 48 materials (48 KB) and 64
// lanes beyond (2 x 256 x 64 x 4 B = 128 KB at the full uint8 range): the general path, not a fast one.
constexpr int kManyMaterials = 48;

constexpr int kWave = 64;

struct SlabPieces {
  int32_t ja, jb;
  float t;
};

// One slab of the fixed-point DDA (mirror: oracle/dexct_oracle.c dda_slab).
__device__ __forceinline__ SlabPieces dda_slab(long long Va, long long SV, uint32_t smask, float kf) {
  SlabPieces s;
  const long long Vb = Va + SV;
  s.ja = (int32_t)(Va >> DEXCT_FIX_FRAC);
  s.jb = (int32_t)(Vb >> DEXCT_FIX_FRAC);
  const uint32_t fr = (uint32_t)((unsigned long long)Va >> 8);
  const float d = (float)(fr ^ smask);
  s.t = fminf(d * kf, 1.0f);
  return s;
}

// The second output of get_sino (main.py:120-122): ln(air / counts) in float32, as np.log(np.float32(air) / raw)
// computes it: v_rcp_f32 and v_log_f32 (1 ulp each); counts == 0 gives +inf like the NumPy expression.
__device__ __forceinline__ float log_ratio(float air, float c) {
  return __builtin_amdgcn_logf(air * __builtin_amdgcn_rcpf(c)) * 0.693147180559945309f;
}

}  // namespace dexct
