// HBM-bound NHWC kernels around the convolutions: layout conversion at the boundary, MFM
// (max / min feature map), 2x2 max pooling.  All tensors fp32, channel stride padded to 4,
// pad channels kept zero.
//
// Replaces SliceChannel/maximum/minimum/Concat (ref: efm_symbol.py:25-30,34-39,55-60,63-64,
// 69-77,96-101; lightcnn.py:22-27,32-37,53-66,123-128) and Pooling (ref: efm_symbol.py:78;
// lightcnn.py:83,89,95,101,107), which MXNet runs as 6+ separate elementwise kernels per MFM.
#include "efm_common.h"

namespace {

// ---- NCHW <-> NHWC(pad4) -------------------------------------------------------------------
// one thread per output pixel-channel-group of 4: reads 4 strided planes, writes 16 B.
__global__ void __launch_bounds__(256) nchw_to_nhwc_k(const float* __restrict__ x, float* __restrict__ y,
                                                      long pixels, int c, int hw, int cp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int ng = cp >> 2;
  if (i >= pixels * ng) return;
  const long pix = i / ng;
  const int g = (int)(i - pix * ng);
  const long b = pix / hw, r = pix - b * hw;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ch = g * 4 + k;
    if (ch < c) v[k] = x[(b * c + ch) * hw + r];
  }
  *reinterpret_cast<f32x4*>(y + pix * cp + g * 4) = v;
}

// ---- NCHW -> row-packed NHWC: y[b][h][w][j*c + ch] = x[b][ch][h][w + j - pad_w] (zero outside the row), j < kw.
// The kw taps of one kernel row become CHANNELS of the pixel, so a kh x kw convolution on c channels turns into a kh x 1
// convolution on kw*c channels with the same weights re-indexed: the first layer's K = kh*kw*c then packs densely
// (5x5 on 3 channels: 5 * pad4(15) = 80 instead of 25 * pad4(3) = 100 -> 112 in fp32; 96 instead of 224 in bf16).
// One thread per (pixel, group of G channels); T = float (G = 4, 16-byte store) or __bf16 (G = 8, 16-byte store).
template <typename T, int G>
__global__ void __launch_bounds__(256) rowpack_nchw_k(const float* __restrict__ x, T* __restrict__ y, unsigned total, efm::FastDiv ngd,
                                                      efm::FastDiv wd, efm::FastDiv cd, int c, int hw, int w, int kw, int pad_w, int cp) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned pix = efm::div(i, ngd);
  const int g = (int)(i - pix * ngd.d);
  const unsigned row = efm::div(pix, wd);            // (b * h + hh)
  const int ww = (int)(pix - row * wd.d);
  const unsigned b = row / (unsigned)(hw / w), hh = row - b * (unsigned)(hw / w);
  T v[G];
#pragma unroll
  for (int k = 0; k < G; ++k) {
    const unsigned s = (unsigned)(g * G + k);
    const unsigned j = efm::div(s, cd);
    const int ch = (int)(s - j * cd.d);
    const int wi = ww + (int)j - pad_w;
    float f = 0.f;
    if ((int)j < kw && (unsigned)wi < (unsigned)w) f = x[((long)b * c + ch) * hw + (long)hh * w + wi];
    v[k] = (T)f;
  }
  T* dst = y + (long)pix * cp + g * G;
#pragma unroll
  for (int k = 0; k < G; ++k) dst[k] = v[k];   // G consecutive elements of one thread: merged into one 16-byte store
}

__global__ void __launch_bounds__(256) nhwc_to_nchw_k(const float* __restrict__ x, float* __restrict__ y,
                                                      long total, int c, int hw, int cp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const long r = i % hw;
  const long t = i / hw;
  const int ch = (int)(t % c);
  const long b = t / c;
  y[i] = x[(b * hw + r) * cp + ch];
}

// ---- MFM -----------------------------------------------------------------------------------
// thread = (row, j) with j < cw = c/ways + (pad channels of y); consecutive threads walk j so
// each of the `ways` slice reads and both writes are coalesced runs.
template <int WAYS, typename T = float>
__global__ void __launch_bounds__(256) mfm_fwd_k(const T* __restrict__ x, T* __restrict__ y, unsigned total,
                                                 int cs, int cp_in, int cp_out, efm::FastDiv cwd) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;  // (row, channel of a slice + pad channel): 32-bit, checked by the launcher
  if (i >= total) return;
  const unsigned rowu = efm::div(i, cwd);
  const int j = (int)(i - rowu * cwd.d);
  const long row = rowu;
  const T* xr = x + row * cp_in;
  T* yr = y + row * cp_out;
  if (j < cs) {
    if (WAYS == 3) {
      const float s0 = (float)xr[j], s1 = (float)xr[cs + j], s2 = (float)xr[2 * cs + j];
      yr[j] = (T)fmaxf(fmaxf(s0, s1), s2);        // max / min of stored values: exact in either element type
      yr[cs + j] = (T)fminf(fminf(s0, s1), s2);
    } else {
      yr[j] = (T)fmaxf((float)xr[j], (float)xr[cs + j]);
    }
  } else {
    const int cout = (WAYS == 3) ? 2 * cs : cs;
    const int pc = cout + (j - cs);
    if (pc < cp_out) yr[pc] = (T)0.f;
  }
}

// MXNet: d maximum(l,r) -> l if l >= r else r ; d minimum(l,r) -> l if l <= r else r.
// ORDER_GROUP: max(max(s0,s1),s2); ORDER_RES: max(s2, max(s0,s1)).
template <int WAYS, typename T = float>
__global__ void __launch_bounds__(256) mfm_bwd_k(const T* __restrict__ x, const T* __restrict__ dy,
                                                 const T* __restrict__ add, T* __restrict__ dx, unsigned total,
                                                 int cs, int cp_in, int cp_out, efm::FastDiv cwd, int order) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i >= total) return;
  const unsigned rowu = efm::div(i, cwd);
  const int j = (int)(i - rowu * cwd.d);
  const long row = rowu;
  const T* xr = x + row * cp_in;
  const T* gr = dy + row * cp_out;
  T* dr = dx + row * cp_in;
  const T* ar = add ? add + row * cp_in : nullptr;
  if (j < cs) {
    if (WAYS == 3) {
      const float s0 = (float)xr[j], s1 = (float)xr[cs + j], s2 = (float)xr[2 * cs + j];
      const float gmax = (float)gr[j], gmin = (float)gr[cs + j];
      int imax = (s0 >= s1) ? 0 : 1;
      int imin = (s0 <= s1) ? 0 : 1;
      const float m1 = fmaxf(s0, s1), n1 = fminf(s0, s1);
      if (order == EFM_MFM_ORDER_GROUP) {
        if (!(m1 >= s2)) imax = 2;
        if (!(n1 <= s2)) imin = 2;
      } else {
        if (s2 >= m1) imax = 2;
        if (s2 <= n1) imin = 2;
      }
      float d0 = (imax == 0 ? gmax : 0.f) + (imin == 0 ? gmin : 0.f);
      float d1 = (imax == 1 ? gmax : 0.f) + (imin == 1 ? gmin : 0.f);
      float d2 = (imax == 2 ? gmax : 0.f) + (imin == 2 ? gmin : 0.f);
      if (ar) { d0 += (float)ar[j]; d1 += (float)ar[cs + j]; d2 += (float)ar[2 * cs + j]; }
      dr[j] = (T)d0; dr[cs + j] = (T)d1; dr[2 * cs + j] = (T)d2;
    } else {
      const float s0 = (float)xr[j], s1 = (float)xr[cs + j];
      const float g = (float)gr[j];
      float d0 = (s0 >= s1) ? g : 0.f, d1 = (s0 >= s1) ? 0.f : g;
      if (ar) { d0 += (float)ar[j]; d1 += (float)ar[cs + j]; }
      dr[j] = (T)d0; dr[cs + j] = (T)d1;
    }
  } else {
    const int pc = WAYS * cs + (j - cs);  // pad channels of x (c % WAYS == 0 => c == WAYS*cs)
    if (pc < cp_in) dr[pc] = (T)0.f;
  }
}

// ---- 2x2 / stride 2 max pooling, floor ------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool2_fwd_k(const float* __restrict__ x, float* __restrict__ y,
                                                      long total, int h, int w, int ho, int wo, int ng) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int g = (int)(i % ng);
  long t = i / ng;
  const int ow = (int)(t % wo); t /= wo;
  const int oh = (int)(t % ho);
  const long b = t / ho;
  const int cp = ng * 4;
  const float* p00 = x + ((b * h + 2 * oh) * w + 2 * ow) * cp + g * 4;
  const f32x4 a = *reinterpret_cast<const f32x4*>(p00);
  const f32x4 bq = *reinterpret_cast<const f32x4*>(p00 + cp);
  const f32x4 cq = *reinterpret_cast<const f32x4*>(p00 + (long)w * cp);
  const f32x4 dq = *reinterpret_cast<const f32x4*>(p00 + (long)w * cp + cp);
  f32x4 m;
#pragma unroll
  for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(a[k], bq[k]), fmaxf(cq[k], dq[k]));
  *reinterpret_cast<f32x4*>(y + i * 4) = m;
}

// Gradient to the first maximum of the window in scan order (h, then w), as MXNet's pooling
// backward does; every window position is written (dx needs no memset when h, w are even; odd
// trailing rows/columns are zeroed by the threads of the last window row/column).
__global__ void __launch_bounds__(256) maxpool2_bwd_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ dx, long total, int h, int w, int ho,
                                                      int wo, int ng) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int g = (int)(i % ng);
  long t = i / ng;
  const int ow = (int)(t % wo); t /= wo;
  const int oh = (int)(t % ho);
  const long b = t / ho;
  const int cp = ng * 4;
  const long o00 = ((b * h + 2 * oh) * w + 2 * ow) * cp + g * 4;
  const long o01 = o00 + cp, o10 = o00 + (long)w * cp, o11 = o10 + cp;
  const f32x4 a = *reinterpret_cast<const f32x4*>(x + o00);
  const f32x4 bq = *reinterpret_cast<const f32x4*>(x + o01);
  const f32x4 cq = *reinterpret_cast<const f32x4*>(x + o10);
  const f32x4 dq = *reinterpret_cast<const f32x4*>(x + o11);
  const f32x4 gq = *reinterpret_cast<const f32x4*>(dy + i * 4);
  f32x4 ra, rb, rc, rd;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int idx = 0;
    float m = a[k];
    if (bq[k] > m) { m = bq[k]; idx = 1; }
    if (cq[k] > m) { m = cq[k]; idx = 2; }
    if (dq[k] > m) { m = dq[k]; idx = 3; }
    ra[k] = idx == 0 ? gq[k] : 0.f;
    rb[k] = idx == 1 ? gq[k] : 0.f;
    rc[k] = idx == 2 ? gq[k] : 0.f;
    rd[k] = idx == 3 ? gq[k] : 0.f;
  }
  *reinterpret_cast<f32x4*>(dx + o00) = ra;
  *reinterpret_cast<f32x4*>(dx + o01) = rb;
  *reinterpret_cast<f32x4*>(dx + o10) = rc;
  *reinterpret_cast<f32x4*>(dx + o11) = rd;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const bool last_w = (ow == wo - 1) && (w & 1), last_h = (oh == ho - 1) && (h & 1);
  if (last_w) {
    *reinterpret_cast<f32x4*>(dx + o01 + cp) = z;
    *reinterpret_cast<f32x4*>(dx + o11 + cp) = z;
  }
  if (last_h) {
    *reinterpret_cast<f32x4*>(dx + o10 + (long)w * cp) = z;
    *reinterpret_cast<f32x4*>(dx + o11 + (long)w * cp) = z;
    if (last_w) *reinterpret_cast<f32x4*>(dx + o11 + (long)w * cp + cp) = z;
  }
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {
// ImageRecordIter's augmentation on the device: decoded uint8 HWC images -> random / centre crop, horizontal mirror, * scale,
// NCHW fp32 (what the iterator emits, ref: train_efm.py:179-181 `scale=1./255, rand_crop=True, rand_mirror=True`).
// One thread per output element; reads are byte gathers from a window of the source image (L2-resident), writes are coalesced.
__global__ void __launch_bounds__(256) crop_mirror_u8_k(const unsigned char* __restrict__ src, const int* __restrict__ crop,
                                                        float* __restrict__ dst, long total, int ih, int iw, int c, int h, int w,
                                                        float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int xo = (int)(i % w);
  long t = i / w;
  const int yo = (int)(t % h);
  t /= h;
  const int ch = (int)(t % c);
  const long b = t / c;
  const int y0 = crop[b * 3], x0 = crop[b * 3 + 1], flip = crop[b * 3 + 2];
  const int xs = x0 + (flip ? (w - 1 - xo) : xo);
  dst[i] = (float)src[((b * ih + (y0 + yo)) * iw + xs) * c + ch] * scale;
}
}  // namespace

extern "C" {

int efm_crop_mirror_u8(const uint8_t* src_hwc, const int32_t* crop, float* dst_nchw, int batch, int ih, int iw, int c, int h, int w,
                       float scale, void* stream) {
  EFM_REQUIRE(src_hwc && crop && dst_nchw && batch > 0 && c > 0 && h > 0 && w > 0 && ih >= h && iw >= w,
              "crop_mirror_u8: bad argument (the crop window must fit the source image)");
  const long total = (long)batch * c * h * w;
  hipLaunchKernelGGL(crop_mirror_u8_k, dim3((unsigned)efm::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, src_hwc, crop, dst_nchw,
                     total, ih, iw, c, h, w, scale);
  return efm::check_launch("crop_mirror_u8");
}

int efm_nchw_to_nhwc(const float* x, float* y, int batch, int c, int h, int w, void* stream) {
  EFM_REQUIRE(x && y && batch > 0 && c > 0 && h > 0 && w > 0, "nchw_to_nhwc: bad argument");
  const int cp = efm_pad4(c);
  const long pixels = (long)batch * h * w;
  const long n = pixels * (cp >> 2);
  hipLaunchKernelGGL(nchw_to_nhwc_k, dim3((unsigned)efm::cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                     pixels, c, h * w, cp);
  return efm::check_launch("nchw_to_nhwc");
}

int efm_rowpack_nchw(const float* x, void* y, int batch, int c, int h, int w, int kw, int pad_w, int bf16, void* stream) {
  EFM_REQUIRE(x && y && batch > 0 && c > 0 && h > 0 && w > 0 && kw > 0 && pad_w >= 0, "rowpack_nchw: bad argument");
  const int cr = kw * c, cp = bf16 ? ((cr + 7) & ~7) : efm_pad4(cr), G = bf16 ? 8 : 4;
  const long pixels = (long)batch * h * w, n = pixels * (cp / G);
  EFM_REQUIRE(n < 0x100000000L && pixels * cp * (bf16 ? 2 : 4) < 0x80000000L, "rowpack_nchw: output reaches 2^31 bytes (split the batch)");
  const dim3 grid((unsigned)efm::cdiv(n, 256));
  if (bf16)
    hipLaunchKernelGGL((rowpack_nchw_k<__bf16, 8>), grid, dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<__bf16*>(y), (unsigned)n,
                       efm::fastdiv(cp / 8), efm::fastdiv(w), efm::fastdiv(c), c, h * w, w, kw, pad_w, cp);
  else
    hipLaunchKernelGGL((rowpack_nchw_k<float, 4>), grid, dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<float*>(y), (unsigned)n,
                       efm::fastdiv(cp / 4), efm::fastdiv(w), efm::fastdiv(c), c, h * w, w, kw, pad_w, cp);
  return efm::check_launch("rowpack_nchw");
}

int efm_nhwc_to_nchw(const float* x, float* y, int batch, int c, int h, int w, void* stream) {
  EFM_REQUIRE(x && y && batch > 0 && c > 0 && h > 0 && w > 0, "nhwc_to_nchw: bad argument");
  const long total = (long)batch * c * h * w;
  hipLaunchKernelGGL(nhwc_to_nchw_k, dim3((unsigned)efm::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                     total, c, h * w, efm_pad4(c));
  return efm::check_launch("nhwc_to_nchw");
}

int efm_mfm_fwd(const float* x, float* y, int64_t rows, int c, int ways, void* stream) {
  EFM_REQUIRE(x && y && rows > 0 && c > 0, "mfm_fwd: bad argument");
  EFM_REQUIRE((ways == 2 || ways == 3) && c % ways == 0, "mfm_fwd: c=%d not divisible by ways=%d", c, ways);
  const int cs = c / ways, cout = (ways == 3) ? 2 * cs : cs;
  const int cp_in = efm_pad4(c), cp_out = efm_pad4(cout);
  const int cw = cs + (cp_out - cout);
  const long n = rows * cw;
  EFM_REQUIRE(n < 0x100000000L, "mfm: more than 2^32 elements");
  dim3 grid((unsigned)efm::cdiv(n, 256));
  if (ways == 3)
    hipLaunchKernelGGL(mfm_fwd_k<3>, grid, dim3(256), 0, (hipStream_t)stream, x, y, (unsigned)((long)rows * cw), c / 3, cp_in, cp_out, efm::fastdiv(cw));
  else
    hipLaunchKernelGGL(mfm_fwd_k<2>, grid, dim3(256), 0, (hipStream_t)stream, x, y, (unsigned)((long)rows * cw), c / 2, cp_in, cp_out, efm::fastdiv(cw));
  return efm::check_launch("mfm_fwd");
}

int efm_mfm_bwd(const float* x, const float* dy, const float* add, float* dx, int64_t rows, int c, int ways,
                int order, void* stream) {
  EFM_REQUIRE(x && dy && dx && rows > 0 && c > 0, "mfm_bwd: bad argument");
  EFM_REQUIRE((ways == 2 || ways == 3) && c % ways == 0, "mfm_bwd: c=%d not divisible by ways=%d", c, ways);
  EFM_REQUIRE(order == EFM_MFM_ORDER_GROUP || order == EFM_MFM_ORDER_RES, "mfm_bwd: bad order %d", order);
  const int cs = c / ways, cout = (ways == 3) ? 2 * cs : cs;
  const int cp_in = efm_pad4(c), cp_out = efm_pad4(cout);
  const int cw = cs + (cp_in - c);
  const long n = rows * cw;
  EFM_REQUIRE(n < 0x100000000L, "mfm: more than 2^32 elements");
  dim3 grid((unsigned)efm::cdiv(n, 256));
  if (ways == 3)
    hipLaunchKernelGGL(mfm_bwd_k<3>, grid, dim3(256), 0, (hipStream_t)stream, x, dy, add, dx, (unsigned)((long)rows * cw), c / 3, cp_in, cp_out, efm::fastdiv(cw), order);
  else
    hipLaunchKernelGGL(mfm_bwd_k<2>, grid, dim3(256), 0, (hipStream_t)stream, x, dy, add, dx, (unsigned)((long)rows * cw), c / 2, cp_in, cp_out, efm::fastdiv(cw), order);
  return efm::check_launch("mfm_bwd");
}

// bf16 activations (channel stride pad8): the stand-alone MFM of the bf16 plan (EFM-29's residual-block inputs)
static inline int pad8c(int c) { return (c + 7) & ~7; }

int efm_mfmb_fwd(const uint16_t* x, uint16_t* y, int64_t rows, int c, int ways, void* stream) {
  EFM_REQUIRE(x && y && rows > 0 && c > 0, "mfmb_fwd: bad argument");
  EFM_REQUIRE((ways == 2 || ways == 3) && c % ways == 0, "mfmb_fwd: c=%d not divisible by ways=%d", c, ways);
  const int cs = c / ways, cout = (ways == 3) ? 2 * cs : cs;
  const int cp_in = pad8c(c), cp_out = pad8c(cout);
  const int cw = cs + (cp_out - cout);
  EFM_REQUIRE(rows * cw < 0x100000000L, "mfmb_fwd: more than 2^32 elements");
  dim3 grid((unsigned)efm::cdiv(rows * cw, 256));
  const __bf16* xb = reinterpret_cast<const __bf16*>(x);
  __bf16* yb = reinterpret_cast<__bf16*>(y);
  if (ways == 3)
    hipLaunchKernelGGL((mfm_fwd_k<3, __bf16>), grid, dim3(256), 0, (hipStream_t)stream, xb, yb, (unsigned)((long)rows * cw), c / 3, cp_in, cp_out, efm::fastdiv(cw));
  else
    hipLaunchKernelGGL((mfm_fwd_k<2, __bf16>), grid, dim3(256), 0, (hipStream_t)stream, xb, yb, (unsigned)((long)rows * cw), c / 2, cp_in, cp_out, efm::fastdiv(cw));
  return efm::check_launch("mfmb_fwd");
}

int efm_mfmb_bwd(const uint16_t* x, const uint16_t* dy, const uint16_t* add, uint16_t* dx, int64_t rows, int c, int ways, int order,
                 void* stream) {
  EFM_REQUIRE(x && dy && dx && rows > 0 && c > 0, "mfmb_bwd: bad argument");
  EFM_REQUIRE((ways == 2 || ways == 3) && c % ways == 0, "mfmb_bwd: c=%d not divisible by ways=%d", c, ways);
  EFM_REQUIRE(order == EFM_MFM_ORDER_GROUP || order == EFM_MFM_ORDER_RES, "mfmb_bwd: bad order %d", order);
  const int cs = c / ways, cout = (ways == 3) ? 2 * cs : cs;
  const int cp_in = pad8c(c), cp_out = pad8c(cout);
  const int cw = cs + (cp_in - c);
  EFM_REQUIRE(rows * cw < 0x100000000L, "mfmb_bwd: more than 2^32 elements");
  dim3 grid((unsigned)efm::cdiv(rows * cw, 256));
  const __bf16 *xb = reinterpret_cast<const __bf16*>(x), *gb = reinterpret_cast<const __bf16*>(dy), *ab = reinterpret_cast<const __bf16*>(add);
  __bf16* db = reinterpret_cast<__bf16*>(dx);
  if (ways == 3)
    hipLaunchKernelGGL((mfm_bwd_k<3, __bf16>), grid, dim3(256), 0, (hipStream_t)stream, xb, gb, ab, db, (unsigned)((long)rows * cw), c / 3, cp_in, cp_out, efm::fastdiv(cw), order);
  else
    hipLaunchKernelGGL((mfm_bwd_k<2, __bf16>), grid, dim3(256), 0, (hipStream_t)stream, xb, gb, ab, db, (unsigned)((long)rows * cw), c / 2, cp_in, cp_out, efm::fastdiv(cw), order);
  return efm::check_launch("mfmb_bwd");
}

int efm_maxpool2_fwd(const float* x, float* y, int batch, int h, int w, int c, void* stream) {
  EFM_REQUIRE(x && y && batch > 0 && h >= 2 && w >= 2 && c > 0, "maxpool2_fwd: bad argument");
  const int ho = h / 2, wo = w / 2, ng = efm_pad4(c) >> 2;
  const long total = (long)batch * ho * wo * ng;
  hipLaunchKernelGGL(maxpool2_fwd_k, dim3((unsigned)efm::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                     total, h, w, ho, wo, ng);
  return efm::check_launch("maxpool2_fwd");
}

int efm_maxpool2_bwd(const float* x, const float* dy, float* dx, int batch, int h, int w, int c, void* stream) {
  EFM_REQUIRE(x && dy && dx && batch > 0 && h >= 2 && w >= 2 && c > 0, "maxpool2_bwd: bad argument");
  const int ho = h / 2, wo = w / 2, ng = efm_pad4(c) >> 2;
  const long total = (long)batch * ho * wo * ng;
  hipLaunchKernelGGL(maxpool2_bwd_k, dim3((unsigned)efm::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, dy,
                     dx, total, h, w, ho, wo, ng);
  return efm::check_launch("maxpool2_bwd");
}

}  // extern "C"
