// Shared helpers of the gfx950 kernels behind include/efm_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/efm_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace efm {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return EFM_E_LAUNCH;
  }
  return EFM_OK;
}

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Wave-wide (64 lanes) sum by xor-shuffles; every lane ends with the total.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace efm

#define EFM_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      efm::set_error(__VA_ARGS__);    \
      return EFM_E_INVALID;           \
    }                                 \
  } while (0)
