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

constexpr int kWave = 64;

}  // namespace dexct
