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

// Shared by the direct and the Winograd weight-gradient paths (defined in efm_conv.hip).
// reduce_slabs: out[i] (+)= sum over `count` slabs of n4 float4 each, fixed order (two levels above 32 slabs; tmp holds ceil(count/32) slabs).
int reduce_slabs(const float* in, float* tmp, float* out, long n4, int count, int accumulate, hipStream_t s);
// bias_grad: dbias[n_pad16] (+)= column sums of dy (M x cout_p); ws holds bias_grad_ws_floats(d) floats.
size_t bias_grad_ws_floats(const efm_conv_desc* d);
int bias_grad(const efm_conv_desc* d, const float* dy, float* dbias, int accumulate, float* ws, hipStream_t s);

}  // namespace efm

#define EFM_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      efm::set_error(__VA_ARGS__);    \
      return EFM_E_INVALID;           \
    }                                 \
  } while (0)
