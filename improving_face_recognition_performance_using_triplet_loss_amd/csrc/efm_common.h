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

// Unsigned 32-bit division by a launch constant in 5 VALU instructions (round-up multiplier, Granlund & Montgomery; exact for every
// 32-bit dividend): the element-wise kernels turn a flat thread index into (pixel, channel) / (image, row, column) with it — a plain
// `/` on a 64-bit index cost ~200 instructions per division there, more than the loads and stores of the kernel.
struct FastDiv {
  unsigned d, m, s;
};
inline FastDiv fastdiv(unsigned d) {
  FastDiv f{d, 0u, 0u};
  if (d <= 1) return f;  // div() passes the dividend through
  unsigned l = 0;
  while ((1ull << l) < d) ++l;  // ceil(log2 d)
  f.m = (unsigned)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s = l - 1;
  return f;
}
__device__ __forceinline__ unsigned div(unsigned x, const FastDiv& f) {
  const unsigned t = __umulhi(f.m, x);
  const unsigned q = (t + ((x - t) >> 1)) >> f.s;
  return f.d == 1 ? x : q;
}

// Kernel instance name + executed MFMA flops of a Winograd launch (defined in efm_winograd.hip; passes 4..6 of efm_conv_kernel_info).
int wino_kernel_info(const efm_conv_desc* d, int pass, int ways, char* name, size_t len, double* flops);
// One launch: dw (+)= sum of `splits` slabs of n4w float4, dbias (+)= sum of `chunks` partials of n4b float4 (dbias may be null);
// fixed order (defined in efm_conv.hip, used by both weight-gradient forms).
int wgrad_reduce(const float* slabs, float* dw, long n4w, int splits, const float* bpart, float* dbias, long n4b, int chunks, int accumulate,
                 hipStream_t s);
// Weight gradient in Winograd form (efm_wino_wgrad.hip) behind efm_conv_bwd_weight_*: selected by bit 12 of tune_wgrad.
bool wino_wgrad_selected(const efm_conv_desc* d);
size_t wino_wgrad_ws_floats(const efm_conv_desc* d);
int wino_wgrad_slabs(const efm_conv_desc* d, const float* x, const float* dy, int want_bias, void* workspace, size_t workspace_bytes,
                     hipStream_t s);
int wino_wgrad_finish(const efm_conv_desc* d, float* dw_packed, float* dbias, int accumulate, const void* workspace, size_t workspace_bytes,
                      hipStream_t s);
int wino_wgrad_info(const efm_conv_desc* d, char* name, size_t len, double* flops);

// bf16 weight gradient in halo-tile form (efm_convb_wgrad.hip) behind efm_convb_bwd_weight: taken wherever it applies (EFM_WGRAD2=0: never).
bool wgrad2_selected(const efm_conv_desc* d);
int wgrad2_splits(const efm_conv_desc* d);
int wgrad2_bias_chunks(const efm_conv_desc* d);
bool wgrad2_expand_supported(const efm_conv_desc* d, int ways, int pool);
int wgrad2_slabs(const efm_conv_desc* d, const uint16_t* x, const uint16_t* dy, const void* dz, const unsigned char* route, float* slabs,
                 float* bias_part, hipStream_t s);
int wgrad2_info(const efm_conv_desc* d, char* name, size_t len, double* flops);

// Raw-buffer offsets are 32 bits and the kernels use byte offset 2^31 (EFM_OOB) as the "always out of range" address that the
// buffer range check turns into zeros (padding taps, tail rows): every activation tensor a convolution kernel addresses must
// therefore stay BELOW 2^31 bytes, or the sentinel would land inside the tensor and read data instead of zeros.
// esize = 4 (fp32, channels padded to 4) or 2 (bf16, channels padded to 8).  Returns nullptr when fine, else which tensor.
inline const char* conv_tensor_too_large(const efm_conv_desc* d, size_t esize) {
  const size_t cin = esize == 2 ? (size_t)((d->cin + 7) / 8 * 8) : (size_t)d->cin_p;
  const size_t cout = esize == 2 ? (size_t)((d->cout + 7) / 8 * 8) : (size_t)d->cout_p;
  if ((size_t)d->batch * d->hin * d->win * cin * esize >= 0x80000000ull) return "input";
  if ((size_t)d->batch * d->hout * d->wout * cout * esize >= 0x80000000ull) return "output";
  return nullptr;
}

}  // namespace efm

#define EFM_REQUIRE_RANGE(d, esize, what)                                                                                  \
  do {                                                                                                                     \
    const char* which__ = efm::conv_tensor_too_large(d, esize);                                                            \
    if (which__) {                                                                                                         \
      efm::set_error("%s: the %s tensor reaches 2^31 bytes (32-bit buffer offsets; split the batch)", what, which__);      \
      return EFM_E_INVALID;                                                                                                \
    }                                                                                                                      \
  } while (0)

#define EFM_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      efm::set_error(__VA_ARGS__);    \
      return EFM_E_INVALID;           \
    }                                 \
  } while (0)
