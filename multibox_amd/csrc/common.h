// Shared helpers for libmbx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mbx.h"

#include <stdio.h>
// Launch checking: MBX_ENTER() drops any stale error another library left in the runtime's
// per-thread slot; MBX_LAUNCH_CHECK() then sees only this launch's own error.
#define MBX_ENTER() (void)hipGetLastError()
#define MBX_LAUNCH_CHECK()                                                              \
  do {                                                                                  \
    hipError_t mbx_e_ = hipGetLastError();                                              \
    if (mbx_e_ != hipSuccess) {                                                         \
      fprintf(stderr, "[libmbx] %s:%d launch failed: %s\n", __FILE__, __LINE__,         \
              hipGetErrorString(mbx_e_));                                               \
      return MBX_ERR_LAUNCH;                                                            \
    }                                                                                   \
  } while (0)

static inline hipStream_t mbx_s(mbx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ int mbx_lane() { return threadIdx.x & 63; }

// 64-lane butterfly reductions (wave = 64 on gfx950).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
