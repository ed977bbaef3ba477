// Implicit-GEMM convolution for gfx950 (MI355X): forward / data-gradient / weight-gradient.
//
// Replaces mx.symbol.Convolution / nn.Conv2D / FullyConnected call sites of the reference
// (efm_symbol.py:32,41,54,62,65,67,94; lightcnn.py:14-15,47-48,111) — the reference ships no
// kernels of its own, MXNet dispatches these to cuDNN.
//
// Design (fp32 in, fp32 accumulate, exact: v_mfma_f32_16x16x4_f32):
//   GEMM view  C[m][n] = sum_k A[m][k] * W[n][k]
//     m = (b, ho, wo) output pixel, n = output channel, k = (kh, kw, ci) with ci fastest.
//   A is never materialised: each 16-byte piece A[m][k..k+3] is 4 consecutive input channels
//   of one NHWC pixel, fetched straight from x (zero outside the image = padding).
//   Block = 256 threads = 4 waves; block tile = (64*MT) pixels x (16*NT) channels, K step 16.
//   Each wave owns 16*MT pixel rows and all NT channel tiles -> MT*NT 16x16 accumulators.
//   Operands are staged global -> registers -> LDS (double buffered, one barrier per K step);
//   the LDS image is [row][16 floats] with the 16-byte slot XOR-swizzled so that both the
//   staging ds_write_b128 and the fragment ds_read_b128 are bank-conflict free.
//   One ds_read_b128 per lane feeds 4 MFMAs: lane (i = lane&15, q = lane>>4) holds
//   A[i][4q..4q+3]; MFMA step j contracts k = {j, 4+j, 8+j, 12+j} on both operands.
//   The data gradient is the same kernel run on dy with tap-flipped, transposed weights.
//   The weight gradient contracts over pixels: C[n][k] = sum_m dy[m][n] * A[m][k], split over
//   m into workspace slabs and reduced in a fixed order (bitwise reproducible).
#include <algorithm>
#include <string.h>

#include <type_traits>

#include "efm_common.h"

namespace {

__host__ __device__ __forceinline__ int swz_g(int b) { return (0x78 >> (2 * b)) & 3; }  // {0,2,3,1}

struct ConvP {
  const void* x;  // activations / weights / residual / output are float or __bf16 (template parameter T of the kernel)
  const void* w;
  const float* bias;
  const void* res;
  void* y;
  int M;
  int hin, win, cin_p;
  int hout, wout, cout_p;
  int kh, kw, pad_h, pad_w;
  int n_pad16, k_pad, ksteps;
  int nblocks;
  unsigned magic_c, magic_kw;  // ceil(2^32 / cin_p), ceil(2^32 / kw): exact k / cin_p and tap / kw for k < 2^16
  unsigned x_bytes, w_bytes;
  // fused MFM (+ 2x2 max pooling) epilogue
  unsigned char* route;  // per output element: slice (and window pixel) the value came from
  int cout, ways, order, pool, hp, wp;
  int cn;  // fused epilogue with several channel blocks: block nb owns channels [nb*cn, nb*cn + cn) of EVERY slice
  int cpo;      // channel stride of the fused epilogue's output z and of the route bytes
  int out_f32;  // bf16 kernel only: write the fused epilogue's z as float (the layer that feeds the fp32 head)
  // bf16 data gradient: K ordered (32-channel chunk, tap, channel) instead of (tap, channel): the 9 taps of a chunk are 9 consecutive
  // K steps, so the shifted re-reads of the same 64 bytes of a pixel come back from L1 / L2 instead of HBM (see efm_convb_bwd_data)
  int chunk_major;
  unsigned magic_taps;  // ceil(2^32 / (kh*kw))
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define EFM_OOB 0x80000000u  // a byte offset no tensor reaches: the buffer range check then returns zeros

// DMA = true : operands go global -> LDS directly (buffer_load ... lds, no VGPR staging, no ds_write);
// DMA = false: global -> VGPR -> ds_write_b128.  Either way every load is an unconditional buffer load whose
// out-of-image / out-of-matrix lanes carry an out-of-range offset (hardware zero fill): no divergent branch in the
// K loop, so the compiler is free to schedule the loads among the MFMAs.
// (The body lives in a __device__ function: the buffer-resource builtins only exist in the device pass, and a
// __global__ template that names them directly loses its host-side launch stub.)
// T = float : v_mfma_f32_16x16x4_f32, 4 channels per 16-byte piece, K step 16 (4 MFMAs per fragment pair);
// T = __bf16: v_mfma_f32_16x16x32_bf16, 8 channels per piece, K step 32 (1 MFMA per fragment pair); fp32 accumulate.
// Either way an LDS row is 64 bytes = one K step of one pixel / output channel, so staging, swizzle and fragment
// addressing are shared.
template <typename T, int MT, int NT, bool DMA, int EPI>
__device__ __forceinline__ void conv_fwd_body(const ConvP& p, float* smem) {
  constexpr int BM = MT * 64, BN = NT * 16;
  constexpr int PB = (NT * 64 + 255) / 256;
  constexpr int EB = sizeof(T), CH = 16 / EB, KS = 4 * CH;
  constexpr bool BF = (EB == 2);
  static_assert(DMA || !BF, "the bf16 kernel stages through LDS-DMA only");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // XCD-aware bijective remap: consecutive logical blocks (same pixel rows, neighbouring
  // channel blocks / neighbouring pixel rows) share one XCD's L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int mb = lid / p.nblocks, nb = lid - mb * p.nblocks;
  const int m0 = mb * BM, n0 = nb * BN;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);

  // ---- staging coordinates: thread -> row = tid/4 (+64 per pass), LDS slot = tid%4 of that row's 64 bytes.
  // The slot holds K piece kc = slot ^ g(row/4 % 4) (XOR swizzle, see file header).
  const int lrow = tid >> 2, slot = tid & 3;
  const int kc4 = (slot ^ swz_g((lrow >> 2) & 3)) * CH;  // element offset of this thread's piece inside a K step
  int a_hi0[MT], a_wi0[MT], a_base[MT];
  const int hw = p.hout * p.wout;
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = m0 + lrow + 64 * j;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    int b, ho, wo;
    if (EPI == 1 && p.pool) {
      // window-major pixel order: rows 4q..4q+3 are the 2x2 pooling window q = (b, hp, wp), scan order (dh, dw);
      // in the MFMA result layout these are the 4 registers of one lane, so pooling needs no cross-lane traffic.
      const int q = mm >> 2, jw = mm & 3;
      const int hwp = p.hp * p.wp;
      b = q / hwp;
      const int r = q - b * hwp;
      const int hq = r / p.wp, wq = r - hq * p.wp;
      ho = 2 * hq + (jw >> 1);
      wo = 2 * wq + (jw & 1);
    } else {
      b = mm / hw;
      const int r = mm - b * hw;
      ho = r / p.wout;
      wo = r - ho * p.wout;
    }
    a_hi0[j] = ok ? ho - p.pad_h : -(1 << 20);
    a_wi0[j] = wo - p.pad_w;
    a_base[j] = ((b * p.hin + ho - p.pad_h) * p.win + (wo - p.pad_w)) * p.cin_p;
  }
  unsigned b_off[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    int n = n0 + lrow + 64 * j;
    if (EPI == 1) {
      // tile column r <-> slice r / cnb, channel cb + r % cnb: every slice of a channel lands in this block, whatever the
      // number of channel blocks (the weight rows are permuted on the fly, the packed layout stays natural)
      const int csl = p.cout / p.ways, cb = nb * p.cn, cnb = min(p.cn, csl - cb);
      const int r = lrow + 64 * j, sl = r / cnb;
      n = (sl < p.ways) ? sl * csl + cb + (r - sl * cnb) : p.n_pad16;  // beyond the last slice: out of range -> zeros
    }
    b_off[j] = (n < p.n_pad16) ? (unsigned)((n * p.k_pad + kc4) * EB) : EFM_OOB;
  }
  const int taps = p.kh * p.kw;

  u32x4 ra[MT], rb[PB];
  auto a_offset = [&](int t, int j, unsigned kh_, unsigned kw_, int doff, bool tap_ok) -> unsigned {
    const int hi = a_hi0[j] + (int)kh_, wi = a_wi0[j] + (int)kw_;
    const bool v = tap_ok && (unsigned)hi < (unsigned)p.hin && (unsigned)wi < (unsigned)p.win;
    return v ? (unsigned)((a_base[j] + doff) * EB) : EFM_OOB;
  };
  auto load_tile = [&](int t, int buf) {
    unsigned tap;
    int c;
    if (BF && p.chunk_major) {   // K step t = (chunk, tap): 32 channels of one tap
      const unsigned chunk = __umulhi((unsigned)t, p.magic_taps);
      tap = (unsigned)t - chunk * (unsigned)(p.kh * p.kw);
      c = (int)chunk * 32 + kc4;
    } else {
      const unsigned k = (unsigned)(t * KS + kc4);
      tap = __umulhi(k, p.magic_c);
      c = (int)(k - tap * (unsigned)p.cin_p);
    }
    // (kw == 1: ceil(2^32 / 1) does not fit the 32-bit multiplier — kx1 kernels, e.g. the row-packed first convolution)
    const unsigned kh_ = (p.kw == 1) ? tap : __umulhi(tap, p.magic_kw), kw_ = tap - kh_ * (unsigned)p.kw;
    const bool tap_ok = (int)tap < taps;
    const int doff = ((int)kh_ * p.win + (int)kw_) * p.cin_p + c;
    if (DMA) {
      float* As = smem + buf * (BM + BN) * 16;
      float* Bs = As + BM * 16;
#pragma unroll
      for (int j = 0; j < MT; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(As + (64 * j + 16 * wave) * 16),
                                                 16, a_offset(t, j, kh_, kw_, doff, tap_ok), 0, 0, 0);
#pragma unroll
      for (int j = 0; j < PB; ++j)
        if (64 * j + 16 * wave < BN)  // wave-uniform: a wave stages 16 whole rows
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(Bs + (64 * j + 16 * wave) * 16),
                                                   16, b_off[j] == EFM_OOB ? EFM_OOB : b_off[j] + (unsigned)(t * 64), 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < MT; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, a_offset(t, j, kh_, kw_, doff, tap_ok), 0, 0);
#pragma unroll
      for (int j = 0; j < PB; ++j) rb[j] = __builtin_amdgcn_raw_buffer_load_b128(wr, b_off[j] + (unsigned)(t * 64), 0, 0);
    }
  };
  auto store_tile = [&](int buf) {
    if (DMA) return;
    float* As = smem + buf * (BM + BN) * 16;
    float* Bs = As + BM * 16;
#pragma unroll
    for (int j = 0; j < MT; ++j) *reinterpret_cast<u32x4*>(As + ((lrow + 64 * j) * 4 + slot) * 4) = ra[j];
#pragma unroll
    for (int j = 0; j < PB; ++j)
      if (lrow + 64 * j < BN) *reinterpret_cast<u32x4*>(Bs + ((lrow + 64 * j) * 4 + slot) * 4) = rb[j];
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int a = 0; a < MT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fq = lane >> 4;
  const int fsw = fq ^ swz_g(fi >> 2);
  auto compute = [&](int buf) {
    const float* As = smem + buf * (BM + BN) * 16;
    const float* Bs = As + BM * 16;
    if constexpr (BF) {
      bf16x8 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int row = wave * (MT * 16) + mt * 16 + fi;
        a[mt] = *reinterpret_cast<const bf16x8*>(As + (row * 4 + fsw) * 4);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(Bs + ((nt * 16 + fi) * 4 + fsw) * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b, acc[mt][nt], 0, 0, 0);
      }
      return;
    }
    f32x4 a[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = wave * (MT * 16) + mt * 16 + fi;
      a[mt] = *reinterpret_cast<const f32x4*>(As + (row * 4 + fsw) * 4);
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt += 2) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + ((nt * 16 + fi) * 4 + fsw) * 4);
      f32x4 b1 = b0;
      if (nt + 1 < NT) b1 = *reinterpret_cast<const f32x4*>(Bs + (((nt + 1) * 16 + fi) * 4 + fsw) * 4);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][jj], b0[jj], acc[mt][nt], 0, 0, 0);
          if (nt + 1 < NT)
            acc[mt][nt + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][jj], b1[jj], acc[mt][nt + 1], 0, 0, 0);
        }
      }
    }
  };

  load_tile(0, 0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < p.ksteps; ++t) {
    const bool more = t + 1 < p.ksteps;
    if (more) load_tile(t + 1, (t + 1) & 1);
    compute(t & 1);
    if (more) store_tile((t + 1) & 1);
    __syncthreads();  // (with DMA in flight hipcc drains vmcnt(0) here: the next tile has landed)
  }

  if (EPI == 0) {
    if constexpr (BF) {
      // ---- plain epilogue, bf16: a lane's result elements are 2 bytes in 16 different places of 4 rows — stored directly, a wave
      // instruction writes four 32-byte slivers (M*N/64 store instructions per launch: measured, these narrow stores and not the
      // matrix cores bound the short-K layers).  So the rows go through the wave's own LDS region [R rows][BN + 4] as fp32 (the
      // staging buffers are dead: the K loop ended with a block barrier; a wave touches only its region) and come back as
      // (row, 8 consecutive channels) per lane: bias added on the way in, residual (one 16-byte load) on the way out, ONE 16-byte
      // store per lane — 8x fewer, 8x wider stores.
      constexpr int R = (NT <= 8) ? 16 : 8;
      constexpr int ES = BN + 4;
      float* Es = smem + wave * (R * ES);
      const int G = min(BN, p.cout_p - n0) >> 3;  // 8-channel groups of this block's columns that exist in y (cout_p % 8 == 0)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int h = 0; h < 16 / R; ++h) {
          if (R == 16 || (fq >> 1) == h) {
            const int rr = (R == 16) ? 4 * fq : 4 * (fq & 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const int n = n0 + nt * 16 + fi;
              const float bv = (p.bias && n < p.n_pad16) ? p.bias[n] : 0.f;
#pragma unroll
              for (int r = 0; r < 4; ++r) Es[(rr + r) * ES + nt * 16 + fi] = acc[mt][nt][r] + bv;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const int row0 = m0 + wave * (MT * 16) + mt * 16 + h * R;
          for (int it = lane; it < R * G; it += 64) {
            const int row = it / G, g = it - row * G;
            const long m = row0 + row;
            if (m < p.M) {
              const float* e = Es + row * ES + g * 8;
              const f32x4 v0 = *reinterpret_cast<const f32x4*>(e), v1 = *reinterpret_cast<const f32x4*>(e + 4);
              float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
              const long off = m * p.cout_p + n0 + g * 8;
              if (p.res) {
                const bf16x8 rv = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(p.res) + off);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += (float)rv[k];
              }
              bf16x8 o;
#pragma unroll
              for (int k = 0; k < 8; ++k) o[k] = (__bf16)v[k];
              *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.y) + off) = o;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          __builtin_amdgcn_wave_barrier();  // the next pass overwrites the region
        }
      }
      return;
    }
    // ---- plain epilogue: D[row = 4*fq + r][col = fi] per 16x16 tile; + bias (+ residual)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + nt * 16 + fi;
      if (n < p.cout_p) {
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = m0 + wave * (MT * 16) + mt * 16 + fq * 4 + r;
            if (m < p.M) {
              const long off = (long)m * p.cout_p + n;
              float v = acc[mt][nt][r] + bv;
              if (p.res) v += (float)reinterpret_cast<const T*>(p.res)[off];
              reinterpret_cast<T*>(p.y)[off] = (T)v;
            }
          }
        }
      }
    }
    return;
  }

  // ---- fused epilogue: + bias -> MFM (max / min feature map across channel slices) -> optional 2x2 max pooling.
  // The slices of one channel sit in different lanes, so each wave transposes R result rows through its own LDS
  // region [row][channel] and re-reads them with lane = channel; pooling windows are 4 consecutive rows.
  constexpr int R = (NT <= 8) ? 16 : 8;
  constexpr int ES = BN + 4;
  float* Es = smem + wave * (R * ES);
  const int ways = p.ways, cs = p.cout / ways;
  const int co = (ways == 3) ? 2 * cs : cs;   // real output channels
  const int cpo = p.cpo;                      // channel stride of z and of the route bytes
  const bool zf32 = !BF || p.out_f32;
  auto put = [&](long idx, float v) {
    if (zf32) reinterpret_cast<float*>(p.y)[idx] = v;
    else reinterpret_cast<__bf16*>(p.y)[idx] = (__bf16)v;
  };
  const int cb = nb * p.cn, cnb = min(p.cn, cs - cb);  // this block's channels [cb, cb + cnb) of every slice
  auto slice3 = [&](float x0, float x1, float x2, float& vmax, int& imax, float& vmin, int& imin) {
    // MXNet: maximum/minimum(lhs, rhs) backward sends a tie to lhs; ORDER_GROUP = max(max(s0,s1),s2), ORDER_RES = max(s2, max(s0,s1))
    imax = (x0 >= x1) ? 0 : 1;
    imin = (x0 <= x1) ? 0 : 1;
    const float m1 = fmaxf(x0, x1), n1 = fminf(x0, x1);
    if (p.order == EFM_MFM_ORDER_GROUP) {
      if (!(m1 >= x2)) imax = 2;
      if (!(n1 <= x2)) imin = 2;
    } else {
      if (x2 >= m1) imax = 2;
      if (x2 <= n1) imin = 2;
    }
    vmax = fmaxf(m1, x2);
    vmin = fminf(n1, x2);
  };
  // The K loop ended with a block barrier, so the staging buffers are dead.  From here on a wave only touches its own
  // region, and the DS operations of one wave execute in program order: no further barrier is needed.
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int h = 0; h < 16 / R; ++h) {
      if (R == 16 || (fq >> 1) == h) {
        const int rr = (R == 16) ? 4 * fq : 4 * (fq & 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int n = nt * 16 + fi;
          const int sl = n / cnb;
          const float bv = (p.bias && sl < ways) ? p.bias[sl * cs + cb + (n - sl * cnb)] : 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) Es[(rr + r) * ES + n] = acc[mt][nt][r] + bv;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int row0 = m0 + wave * (MT * 16) + mt * 16 + h * R;  // first result row of this pass
      bool done_vec = false;
      if constexpr (BF) {
        // ---- bf16, slice widths on the 8-channel grid (every MFM layer of LightCNN-9 / the deeper CNN): lane = (window | pixel,
        // group of 8 channels): the slices' values come from the region as 16-byte reads, z leaves as ONE 16-byte store and the
        // route bytes as ONE 8-byte store per lane and slice half.  Same comparisons in the same order as the scalar path below
        // (same values, same route bytes) — what changes is 4-8x fewer, 8x wider store instructions: the narrow stores, not the
        // matrix cores, bounded the short-K layers (conv1, the 1x1 convolutions).
        if (!zf32 && ways == 2 && (cnb & 7) == 0 && (cb & 7) == 0 && (cs & 7) == 0) {
          // (two-slice MFM only: the three-slice form would keep twice the state live next to the accumulators and spill.
          //  4 channels per lane: 8 per lane left most of a wave idle on the 8-row passes of the wide layers — 2 windows x 12 groups
          //  = 24 of 64 lanes — and cost the kernel a resident block in registers; stores are 8 bytes of z + 4 route bytes per lane,
          //  the lanes of a window contiguous.)
          done_vec = true;
          const int G = cnb >> 2;
          __bf16* zb = reinterpret_cast<__bf16*>(p.y);
          const int items = (p.pool ? R / 4 : R) * G;
          for (int it = lane; it < items; it += 64) {
            const int u = it / G, g = it - u * G;              // u = window (pooled) or row of this pass
            const int m = row0 + (p.pool ? 4 * u : u);
            if (m >= p.M) continue;
            const float* e = Es + (p.pool ? 4 * u : u) * ES + g * 4;
            f32x4 bmax;
            unsigned rt4 = 0u;                                  // route bytes of the 4 channels
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (j > 0 && !p.pool) break;
              const f32x4 a0 = *reinterpret_cast<const f32x4*>(e + j * ES), b0 = *reinterpret_cast<const f32x4*>(e + j * ES + cnb);
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const unsigned rt = (unsigned)(j * 4) + ((a0[k] >= b0[k]) ? 0u : 1u);   // tie -> slice 0 (MXNet: lhs wins)
                const float vmax = fmaxf(a0[k], b0[k]);
                const bool take = (j == 0) || vmax > bmax[k];                           // first maximum of the window wins
                bmax[k] = take ? vmax : bmax[k];
                rt4 = take ? ((rt4 & ~(0xffu << (8 * k))) | (rt << (8 * k))) : rt4;
              }
            }
            const long q = p.pool ? (long)(m >> 2) : (long)m;
            const long o = q * cpo + cb + g * 4;
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 zv;
#pragma unroll
            for (int k = 0; k < 4; ++k) zv[k] = (__bf16)bmax[k];
            *reinterpret_cast<bf16x4*>(zb + o) = zv;
            *reinterpret_cast<unsigned*>(p.route + o) = rt4;
          }
          // co = cs or 2*cs is a multiple of 8 here, so cpo == co: no pad channels to clear
        }
      }
      if (done_vec) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        continue;
      }
      if (p.pool) {
#pragma unroll
        for (int wnd = 0; wnd < R / 4; ++wnd) {
          const int m = row0 + 4 * wnd;
          if (m < p.M) {
            const long q = m >> 2;  // pooled output pixel (b, hp, wp), linear
            for (int c = lane; c < cnb; c += 64) {
              float bmax = 0.f, bmin = 0.f;
              int rmax = 0, rmin = 0;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float* e = Es + (4 * wnd + j) * ES;
                float vmax, vmin = 0.f;
                int imax, imin = 0;
                if (ways == 3) {
                  slice3(e[c], e[cnb + c], e[2 * cnb + c], vmax, imax, vmin, imin);
                } else {
                  const float x0 = e[c], x1 = e[cnb + c];
                  imax = (x0 >= x1) ? 0 : 1;
                  vmax = fmaxf(x0, x1);
                }
                if (j == 0 || vmax > bmax) { bmax = vmax; rmax = j * 4 + imax; }   // first maximum of the window wins
                if (j == 0 || vmin > bmin) { bmin = vmin; rmin = j * 4 + imin; }
              }
              put(q * cpo + cb + c, bmax);
              p.route[q * cpo + cb + c] = (unsigned char)rmax;
              if (ways == 3) {
                put(q * cpo + cs + cb + c, bmin);
                p.route[q * cpo + cs + cb + c] = (unsigned char)rmin;
              }
            }
            if (nb == 0 && lane < cpo - co) put(q * cpo + co + lane, 0.f);
          }
        }
      } else {
        for (int row = 0; row < R; ++row) {
          const long m = row0 + row;
          if (m < p.M) {
            const float* e = Es + row * ES;
            for (int c = lane; c < cnb; c += 64) {
              if (ways == 3) {
                float vmax, vmin;
                int imax, imin;
                slice3(e[c], e[cnb + c], e[2 * cnb + c], vmax, imax, vmin, imin);
                put(m * cpo + cb + c, vmax);
                put(m * cpo + cs + cb + c, vmin);
                p.route[m * cpo + cb + c] = (unsigned char)imax;
                p.route[m * cpo + cs + cb + c] = (unsigned char)imin;
              } else {
                const float x0 = e[c], x1 = e[cnb + c];
                put(m * cpo + cb + c, fmaxf(x0, x1));
                p.route[m * cpo + cb + c] = (unsigned char)((x0 >= x1) ? 0 : 1);
              }
            }
            if (nb == 0 && lane < cpo - co) put(m * cpo + co + lane, 0.f);
          }
        }
      }
    }
  }
}

#ifndef EFM_FWD_OCC
#define EFM_FWD_OCC 2
#endif
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <typename T, int MT, int NT, bool DMA, int EPI>
__global__ void __launch_bounds__(256, EFM_FWD_OCC) conv_fwd_k(const ConvP p) {
  // K-loop double buffer, re-used by the fused epilogue as 4 per-wave transposition regions of R rows x (BN + 4)
  __shared__ __attribute__((aligned(16))) float smem[cmax(2 * (MT * 64 + NT * 16) * 16,
                                                            (EPI || sizeof(T) == 2) ? 4 * ((NT <= 8) ? 16 : 8) * (NT * 16 + 4) : 0)];
  conv_fwd_body<T, MT, NT, DMA, EPI>(p, smem);
}

// Backward of the fused MFM (+ pooling) epilogue: scatters dz to the conv-output positions recorded in `route` and
// writes EVERY element of dy (zeros elsewhere, pad channels, and the odd trailing row / column that floor pooling drops).
// thread = (window or pixel, channel j); j < cs + pad channels.
template <typename TZ, typename TY>
__global__ void __launch_bounds__(256) mfm_pool_bwd_k(const unsigned char* __restrict__ route, const TZ* __restrict__ dz,
                                                      TY* __restrict__ dy, unsigned total, int h, int w, int c, int cs, int ways,
                                                      int pool, efm::FastDiv cwd, efm::FastDiv wgd, efm::FastDiv hgd, int cp, int cpo) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;  // (item, channel of a slice + pad channel): 32-bit, checked by the launcher
  if (i >= total) return;
  const unsigned itu = efm::div(i, cwd);
  const int j = (int)(i - itu * cwd.d);
  const long it = itu;
  if (!pool) {
    TY* d = dy + it * cp;
    if (j < cs) {
      const float gmax = (float)dz[it * cpo + j];
      const int imax = route[it * cpo + j];
      float o0 = imax == 0 ? gmax : 0.f, o1 = imax == 1 ? gmax : 0.f, o2 = imax == 2 ? gmax : 0.f;
      if (ways == 3) {
        const float gmin = (float)dz[it * cpo + cs + j];
        const int imin = route[it * cpo + cs + j];
        o0 += imin == 0 ? gmin : 0.f; o1 += imin == 1 ? gmin : 0.f; o2 += imin == 2 ? gmin : 0.f;
        d[2 * cs + j] = (TY)o2;
      }
      d[j] = (TY)o0; d[cs + j] = (TY)o1;
    } else if (c + (j - cs) < cp) {
      d[c + (j - cs)] = (TY)0.f;
    }
    return;
  }
  // pooled: item = window of the ceil grid (partial windows at odd edges only write zeros)
  const int hp = h >> 1, wp = w >> 1;
  const unsigned tu = efm::div(itu, wgd);
  const int wq = (int)(itu - tu * wgd.d);
  const unsigned bu = efm::div(tu, hgd);
  const int hq = (int)(tu - bu * hgd.d);
  const long b = bu;
  const bool full = hq < hp && wq < wp;
  float gmax = 0.f, gmin = 0.f;
  int rmax = -1, rmin = -1;
  if (full && j < cs) {
    const long q = (b * hp + hq) * wp + wq;
    gmax = (float)dz[q * cpo + j];
    rmax = route[q * cpo + j];
    if (ways == 3) { gmin = (float)dz[q * cpo + cs + j]; rmin = route[q * cpo + cs + j]; }
  }
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    const int hh = 2 * hq + (px >> 1), ww = 2 * wq + (px & 1);
    if (hh >= h || ww >= w) continue;
    TY* d = dy + ((b * h + hh) * w + ww) * cp;
    if (j < cs) {
      for (int sl = 0; sl < ways; ++sl) {
        float o = (rmax == px * 4 + sl) ? gmax : 0.f;
        if (ways == 3 && rmin == px * 4 + sl) o += gmin;
        d[sl * cs + j] = (TY)o;
      }
    } else if (c + (j - cs) < cp) {
      d[c + (j - cs)] = (TY)0.f;
    }
  }
}



// ------------------------------------------------------------------------------------------
// Weight gradient.  C[n][k] = sum_m dy[m][n] * A[m][k].  MFMA A operand = dy (rows = n),
// B operand = im2col(x) (cols = k) so that a result register holds 16 consecutive k of one n
// = 64 contiguous bytes of the packed gradient.  Block = 4 waves, each owning KPW k-tiles x NTW
// n-tiles; 16 pixels per step (4 MFMA contractions of 4 pixels).
// ------------------------------------------------------------------------------------------
struct WgradP {
  const float* x;
  const float* dy;
  float* ws;
  float* bias_part;  // [splits][n_pad16] partial column sums of dy, or nullptr
  int M;
  int hin, win, cin_p;
  int hout, wout, cout_p;
  int kh, kw, pad_h, pad_w;
  int n_pad16, k_pad;
  int kblocks, nblocks, splits;
  int m_per_split;
  int mma_blocks;
  int ktiles, ntiles;  // 16-wide tiles of the packed gradient: k_pad / 16, n_pad16 / 16
  int bias_rows;       // dy rows per column-sum block
  int dbg;             // ablation builds (-DEFM_ABLATE + EFM_WGRAD_DBG): 1 = no operand loads, 2 = no MFMA loop; always 0 in the product
  unsigned x_bytes, y_bytes;
};

// Exact m / d for 0 <= m < 2^23 via a float reciprocal and a one-step fix-up (branch free).
__device__ __forceinline__ int fdiv(int m, int d, float inv) {
  int q = (int)((float)m * inv);
  int r = m - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

template <int KPW, int NTW>
__device__ __forceinline__ void conv_wgrad_body(const WgradP& p, float* smem) {
  constexpr int BKR = 64 * KPW, BNW = 16 * NTW, BP = 16;
  constexpr int PX = KPW;                     // x pieces per thread per step
  constexpr int PY = (NTW * 64 + 255) / 256;  // dy pieces per thread per step
  constexpr int K4 = BKR / 4;                 // 16-byte pieces per pixel row of the x tile
  constexpr int N4 = BNW / 4;
  constexpr bool SWZ_Y = (NTW % 2 == 0);      // odd NTW: row stride = 16 (mod 32) floats, already conflict free
  constexpr int TILE = BP * (BKR + BNW);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware bijective remap over the matrix-core blocks (workgroups go round-robin over the 8 XCDs, each with its own L2): the
  // kblocks*nblocks tiles of one pixel split read the same x / dy rows, so consecutive logical blocks share one XCD
  int bid = blockIdx.x;
  {
    const int nwg = p.mma_blocks;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  const int tiles = p.kblocks * p.nblocks;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int kb = bid / p.nblocks, nb = bid - kb * p.nblocks;
  // The 16-wide tiles of the gradient are dealt EVENLY to the k / n blocks (a block owns nkt <= 4*KPW k-tiles and nnt <= NTW
  // n-tiles) and, inside a block, k-tile j goes to wave j % 4: tile slots a block does not own cost no MFMA — the matrix pipe of
  // a SIMD is shared with the waves of the other resident blocks, so every skipped MFMA is theirs to use (with 8 or 13 slots per
  // block and e.g. 25 k-tiles, padding the last block with zero tiles cost up to 28 % of a layer's MFMAs).
  const int kt_begin = (int)(((long)kb * p.ktiles) / p.kblocks), nkt = (int)(((long)(kb + 1) * p.ktiles) / p.kblocks) - kt_begin;
  const int nt_begin = (int)(((long)nb * p.ntiles) / p.nblocks), nnt = (int)(((long)(nb + 1) * p.ntiles) / p.nblocks) - nt_begin;
  const int k0 = kt_begin * 16, n0 = nt_begin * 16;
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int hw = p.hout * p.wout;
  const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)p.wout;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.y_bytes, 0x00020000);

  // LDS image: Xs[pixel][BKR], Ys[pixel][BNW], no padding (an LDS-DMA wave instruction writes 1 KiB contiguously);
  // bank conflicts of the 4-pixel fragment reads are removed by XOR-ing the 16-byte piece index with 4*(pixel&1),
  // applied on the SOURCE address here and on the read address below.
  // x pieces: e = tid + 256 j -> pixel e / K4, LDS piece e % K4; (tap, channel) is fixed per thread.
  const int xpiece = tid % K4, xp0 = tid / K4;
  const int xk4 = xpiece ^ (4 * (xp0 & 1));
  const int kglob = k0 + xk4 * 4;
  const int tap = kglob / p.cin_p, xc = kglob - tap * p.cin_p;
  const int xkh = tap / p.kw, xkw = tap - xkh * p.kw;
  const bool xtap_ok = tap < p.kh * p.kw && xk4 * 4 < nkt * 16;
  const int xdoff = ((xkh - p.pad_h) * p.win + (xkw - p.pad_w)) * p.cin_p + xc;

  auto load_tile = [&](int step, int buf) {
    float* Xs = smem + buf * TILE;
    float* Ys = Xs + BP * BKR;
    const int mbase = m_begin + step * BP;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      const int m = mbase + xp0 + (256 / K4) * j;
      const int b = fdiv(m, hw, inv_hw), r = m - b * hw;
      const int ho = fdiv(r, p.wout, inv_w), wo = r - ho * p.wout;
      const int hi = ho - p.pad_h + xkh, wi = wo - p.pad_w + xkw;
      const bool v = xtap_ok && m < m_end && (unsigned)hi < (unsigned)p.hin && (unsigned)wi < (unsigned)p.win;
      const unsigned off = v ? (unsigned)((((b * p.hin + ho) * p.win + wo) * p.cin_p + xdoff) * 4) : EFM_OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(Xs + (256 * j + 64 * wave) * 4),
                                               16, off, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < PY; ++j) {
      if (256 * j + 64 * wave < BP * N4) {  // wave-uniform
        const int e = tid + 256 * j;
        const int pp = e / N4;
        int n4 = e - pp * N4;
        if (SWZ_Y) n4 ^= 4 * (pp & 1);
        const int m = mbase + pp, n = n0 + n4 * 4;
        const bool v = m < m_end && n < p.cout_p && n4 * 4 < nnt * 16;
        const unsigned off = v ? (unsigned)((m * p.cout_p + n) * 4) : EFM_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (__attribute__((address_space(3))) void*)(Ys + (256 * j + 64 * wave) * 4),
                                                 16, off, 0, 0, 0);
      }
    }
  };

  f32x4 acc[KPW][NTW];
#pragma unroll
  for (int a = 0; a < KPW; ++a)
#pragma unroll
    for (int b = 0; b < NTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fi = lane & 15, fq = lane >> 4;
  const int fx = 16 * (fq & 1);  // the read-side half of the XOR swizzle (16 floats = 4 pieces)
  // k-tiles this wave owns: slots kt = 0 .. nkw-1 (slot kt holds the block's tile kt*4 + wave).  The MFMA loop exists once per
  // possible count, fully unrolled and branch-free inside (a per-tile branch costs more than the MFMA it saves); the count is
  // wave-uniform, so picking the loop is one scalar branch per 16-pixel step.
  const int nkw = (nkt > wave) ? (nkt - wave + 3) / 4 : 0;
  // Bias gradient = column sums of dy: the dy fragments pass through the registers of every wave anyway, so wave 0 of the first
  // k-block adds them up on the side (NTW v_add per 4-pixel step, in the shadow of KPW*NTW MFMAs) — no second pass over dy, no
  // column-sum blocks competing for the CUs.  One partial per pixel split, reduced in fixed order by wgrad_reduce_k.
  // The n-tiles are dealt to the 4 waves (wave w sums tiles w, w+4, ...): (NTW+3)/4 registers and adds per wave instead of NTW in one.
  constexpr int NB4 = (NTW + 3) / 4;
  const bool do_bias = p.bias_part != nullptr && kb == 0;
  float bsum[NB4];
#pragma unroll
  for (int j = 0; j < NB4; ++j) bsum[j] = 0.f;
  auto compute = [&](auto nk_tag, int buf) {
    constexpr int NK = decltype(nk_tag)::value;
    const float* Xs = smem + buf * TILE;
    const float* Ys = Xs + BP * BKR;
#pragma unroll
    for (int s = 0; s < BP / 4; ++s) {
      float bx[NK];
#pragma unroll
      for (int kt = 0; kt < NK; ++kt) bx[kt] = Xs[(4 * s + fq) * BKR + (((kt * 4 + wave) * 16 + fi) ^ fx)];
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const float ay = Ys[(4 * s + fq) * BNW + (SWZ_Y ? ((nt * 16 + fi) ^ fx) : (nt * 16 + fi))];
#pragma unroll
        for (int kt = 0; kt < NK; ++kt)
          acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, bx[kt], acc[kt][nt], 0, 0, 0);
      }
    }
  };

  // column sums of the 16-pixel dy tile, after the MFMAs of the step (its own pass over LDS: 4*NB4 ds_reads per wave, only in the
  // k-block-0 blocks; kept out of the unrolled MFMA region, where the registers are scarcest)
  auto bias_acc = [&](int buf) {
    const float* Ys = smem + buf * TILE + BP * BKR;
#pragma unroll
    for (int j = 0; j < NB4; ++j) {
      const int nt = wave + 4 * j, ntc = nt < NTW ? nt : NTW - 1;
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < BP / 4; ++s) v += Ys[(4 * s + fq) * BNW + (SWZ_Y ? ((ntc * 16 + fi) ^ fx) : (ntc * 16 + fi))];
      bsum[j] += nt < NTW ? v : 0.f;
    }
  };

  const int steps = (m_end - m_begin + BP - 1) / BP;
  const bool dl = !(p.dbg & 1), dc = !(p.dbg & 2);
  if (steps > 0 && dl) load_tile(0, 0);
  __syncthreads();
  for (int t = 0; t < steps; ++t) {
    if (t + 1 < steps && dl) load_tile(t + 1, (t + 1) & 1);
    if (!dc) { __syncthreads(); continue; }
    // one scalar branch per 16-pixel step picks the fully unrolled loop for this wave's tile count (wave 0, the bias wave, always
    // owns the block's first k-tile)
    if (nkw == KPW)
      compute(std::integral_constant<int, KPW>{}, t & 1);
    else if (KPW == 2 && nkw == 1)
      compute(std::integral_constant<int, 1>{}, t & 1);
    if (do_bias) bias_acc(t & 1);
    __syncthreads();  // drains the LDS-DMA of tile t+1 (vmcnt(0)) and fences the reads of tile t
  }
  if (do_bias) {  // lane (fi, fq) holds the sums of pixels == fq (mod 4): add the four pixel groups, lanes fq == 0 store
#pragma unroll
    for (int j = 0; j < NB4; ++j) {
      const int nt = wave + 4 * j;
      float v = bsum[j];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (fq == 0 && nt < nnt) p.bias_part[(long)split * p.n_pad16 + n0 + nt * 16 + fi] = v;
    }
  }

  float* ws = p.ws + (long)split * p.n_pad16 * p.k_pad;
#pragma unroll
  for (int kt = 0; kt < KPW; ++kt) {
    const int k = k0 + (kt * 4 + wave) * 16 + fi;
    if (kt < nkw) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        if (nt < nnt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) ws[(long)(n0 + nt * 16 + fq * 4 + r) * p.k_pad + k] = acc[kt][nt][r];
        }
      }
    }
  }
}

constexpr int BIAS_ROWS = 1024;  // dy rows per bias partial (fp32: summed inside the k-block-0 blocks; bf16: column-sum blocks)

template <int KPW, int NTW>
__global__ void __launch_bounds__(256, 3) conv_wgrad_k(const WgradP p) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 16 * (64 * KPW + 16 * NTW)];
  conv_wgrad_body<KPW, NTW>(p, smem);
}

// out[g][i] = sum_{s in group g} ws[s*stride + i] (+ out[i] when accumulating), fixed order (deterministic).
// Block = 64 float4 columns x 4 split-lanes; grid = (ceil(n4/64), groups).  A long list of slabs is reduced in two
// levels (groups of `per_group` slabs -> one slab each -> final) so that the loop a thread runs stays short.
__global__ void __launch_bounds__(256) slab_reduce_k(const float* __restrict__ ws, float* __restrict__ out,
                                                     long n4, long stride4, int splits, int per_group,
                                                     long out_stride4, int accumulate) {
  __shared__ __attribute__((aligned(16))) float red[4][64 * 4];
  const int cx = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + cx;
  const int g = blockIdx.y;
  const int s_begin = g * per_group, s_end = min(splits, s_begin + per_group);
  const f32x4* w = reinterpret_cast<const f32x4*>(ws);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i < n4)
    for (int k = s_begin + sl; k < s_end; k += 4) s += w[(long)k * stride4 + i];
  *reinterpret_cast<f32x4*>(&red[sl][cx * 4]) = s;
  __syncthreads();
  if (sl == 0 && i < n4) {
    f32x4 t = *reinterpret_cast<f32x4*>(&red[0][cx * 4]);
    t += *reinterpret_cast<f32x4*>(&red[1][cx * 4]);
    t += *reinterpret_cast<f32x4*>(&red[2][cx * 4]);
    t += *reinterpret_cast<f32x4*>(&red[3][cx * 4]);
    f32x4* o = reinterpret_cast<f32x4*>(out) + (long)g * out_stride4 + i;
    if (accumulate) t += *o;
    *o = t;
  }
}

// The direct weight gradient's reduction as ONE launch: blocks [0, gx_w) sum the `splits` slabs of the packed gradient (64 float4
// columns x 4 slab lanes per block), blocks [gx_w, ...) sum the `chunks` column-sum partials of the bias gradient (16 columns x 16
// lanes: few columns, up to 3136 partials).  Every output element is a fixed-order sum (lane l takes slabs l, l+L, ..., then the
// lanes are added in index order): bitwise reproducible, no atomics.
template <int COLS, int LANES>
__device__ __forceinline__ void reduce_part(const float* __restrict__ in, float* __restrict__ out, long n4, int count, int accumulate,
                                            int bx, float* red) {
  const int cx = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const long i = (long)bx * COLS + cx;
  const f32x4* w = reinterpret_cast<const f32x4*>(in);
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (i < n4) {
    int k = sl;
    for (; k + LANES < count; k += 2 * LANES) {  // two independent chains keep more loads in flight
      s0 += w[(long)k * n4 + i];
      s1 += w[(long)(k + LANES) * n4 + i];
    }
    if (k < count) s0 += w[(long)k * n4 + i];
  }
  *reinterpret_cast<f32x4*>(red + (sl * COLS + cx) * 4) = s0 + s1;
  __syncthreads();
  if (sl == 0 && i < n4) {
    f32x4 t = *reinterpret_cast<f32x4*>(red + cx * 4);
#pragma unroll
    for (int l = 1; l < LANES; ++l) t += *reinterpret_cast<f32x4*>(red + (l * COLS + cx) * 4);
    f32x4* o = reinterpret_cast<f32x4*>(out) + i;
    if (accumulate) t += *o;
    *o = t;
  }
}

__global__ void __launch_bounds__(256) wgrad_reduce_k(const float* __restrict__ slabs, float* __restrict__ dw, long n4w, int splits,
                                                      const float* __restrict__ bpart, float* __restrict__ dbias, long n4b, int chunks,
                                                      int accumulate, int gx_w) {
  __shared__ __attribute__((aligned(16))) float red[256 * 4];
  if ((int)blockIdx.x < gx_w)
    reduce_part<64, 4>(slabs, dw, n4w, splits, accumulate, (int)blockIdx.x, red);
  else
    reduce_part<16, 16>(bpart, dbias, n4b, chunks, accumulate, (int)blockIdx.x - gx_w, red);
}

// ------------------------------------------------------------------------------------------
// Weight layout transforms (one thread per packed element).
// ------------------------------------------------------------------------------------------
__global__ void pack_w_k(const float* __restrict__ w_oihw, float* __restrict__ wp, efm_conv_desc d) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)d.n_pad16 * d.k_pad;
  if (i >= total) return;
  const int n = (int)(i / d.k_pad), k = (int)(i - (long)n * d.k_pad);
  const int tap = k / d.cin_p, ci = k - tap * d.cin_p;
  float v = 0.f;
  if (n < d.cout && tap < d.kh * d.kw && ci < d.cin) {
    const int kh = tap / d.kw, kw = tap - kh * d.kw;
    v = w_oihw[(((long)n * d.cin + ci) * d.kh + kh) * d.kw + kw];
  }
  wp[i] = v;
}

__global__ void unpack_w_k(const float* __restrict__ wp, float* __restrict__ w_oihw, efm_conv_desc d) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)d.cout * d.cin * d.kh * d.kw;
  if (i >= total) return;
  long r = i;
  const int kw = (int)(r % d.kw); r /= d.kw;
  const int kh = (int)(r % d.kh); r /= d.kh;
  const int ci = (int)(r % d.cin); r /= d.cin;
  const int n = (int)r;
  w_oihw[i] = wp[(long)n * d.k_pad + (kh * d.kw + kw) * d.cin_p + ci];
}

// wd[ci][(KH-1-kh, KW-1-kw, co)] = w[co][(kh, kw, ci)]
__global__ void dgrad_w_k(const float* __restrict__ wp, float* __restrict__ wd, efm_conv_desc d) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)d.dn_pad16 * d.dk_pad;
  if (i >= total) return;
  const int ci = (int)(i / d.dk_pad), k = (int)(i - (long)ci * d.dk_pad);
  const int tap = k / d.cout_p, co = k - tap * d.cout_p;
  float v = 0.f;
  if (ci < d.cin && tap < d.kh * d.kw && co < d.cout) {
    const int fkh = tap / d.kw, fkw = tap - fkh * d.kw;
    const int kh = d.kh - 1 - fkh, kw = d.kw - 1 - fkw;
    v = wp[(long)co * d.k_pad + (kh * d.kw + kw) * d.cin_p + ci];
  }
  wd[i] = v;
}

// ------------------------------------------------------------------------------------------
// Host-side dispatch
// ------------------------------------------------------------------------------------------
int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return s ? atoi(s) : dflt;
}

// Rows per block = 64 * MT.  A bf16 MFMA is 8x shorter than the fp32 one on the same fragment bytes, so the bf16 kernels need
// the B fragments re-used over several row tiles (MT = 2, 4) to get off the LDS-bandwidth bound: MFMA cycles / LDS cycles per
// K step = MT*NT / (2*(MT + NT)).  The accumulators (4*MT*NT registers) cap the product.
constexpr int mt_fit(int mt, int nt) { return (mt * nt <= 44) ? mt : ((2 * nt <= 44) ? 2 : 1); }

template <typename T, int MT, bool DMA>
int launch_fwd_nt(int NT, dim3 grid, hipStream_t s, const ConvP& p) {
  switch (NT) {
#define EFM_CASE(N)                                          \
  case N:                                                    \
    hipLaunchKernelGGL((conv_fwd_k<T, mt_fit(MT, N), N, DMA, 0>), grid, dim3(256), 0, s, p); \
    return EFM_OK;
    EFM_CASE(3) EFM_CASE(5) EFM_CASE(6) EFM_CASE(7) EFM_CASE(8) EFM_CASE(9) EFM_CASE(11) EFM_CASE(13)
#undef EFM_CASE
  }
  efm::set_error("conv_fwd: unsupported NT=%d", NT);
  return EFM_E_INVALID;
}

// fused-epilogue variants: one channel block holds every slice of a channel
template <typename T, int MT>
int launch_fwd_epi(int NT, dim3 grid, hipStream_t s, const ConvP& p) {
  switch (NT) {
#define EFM_CASE(N)                                          \
  case N:                                                    \
    hipLaunchKernelGGL((conv_fwd_k<T, mt_fit(MT, N), N, true, 1>), grid, dim3(256), 0, s, p); \
    return EFM_OK;
    EFM_CASE(3) EFM_CASE(5) EFM_CASE(7) EFM_CASE(9) EFM_CASE(11) EFM_CASE(13) EFM_CASE(17) EFM_CASE(25)
#undef EFM_CASE
  }
  efm::set_error("conv_mfm_fwd: unsupported NT=%d", NT);
  return EFM_E_INVALID;
}

int round_nt_epi(int nt) {
  static const int ok[] = {3, 5, 7, 9, 11, 13, 17, 25};
  for (int v : ok)
    if (v >= nt) return v;
  return -1;
}

int round_nt(int nt) {
  static const int ok[] = {3, 5, 6, 7, 8, 9, 11, 13};
  for (int v : ok)
    if (v >= nt) return v;
  return 13;
}

// bf16: the largest row tile that still leaves >= 3 blocks per CU.  256-row tiles (MT = 4) are built and selectable through the
// tune field, but measured equal or slightly slower than 128 rows on LightCNN-9 (the small-N layers are bound by the 9x im2col
// re-fetch of the A operand through L2 -> LDS, not by B-fragment re-use), so the default stops at 2.
int pick_mt_bf16(long M, int nblocks) {
  const int want = env_int("EFM_CONVB_MIN_BLOCKS", 256);  // swept 256..3072: 128-row tiles as soon as one block per CU remains
  if (env_int("EFM_CONVB_MT_MAX", 2) >= 4 && efm::cdiv(M, 256) * nblocks >= want) return 4;
  if (efm::cdiv(M, 128) * nblocks >= want) return 2;
  return 1;
}

struct FwdTiling {
  int NT, MT, nblocks;
};

// Block shape of the plain forward / data-gradient launch: NT 16-column tiles x 64*MT rows (shared by run_fwd and efm_conv_kernel_info).
FwdTiling fwd_tiling(long M, int n_pad16, int tune, int esize) {
  FwdTiling ft;
  const int tiles = n_pad16 / 16;
  int nblocks = (tiles + 12) / 13;
  if (((tune >> 4) & 15) > nblocks) nblocks = std::min((tune >> 4) & 15, tiles);  // bits 9:8 belong to the Winograd kernels
  const int NT = round_nt((tiles + nblocks - 1) / nblocks);
  ft.nblocks = (tiles + NT - 1) / NT;
  // 64-row tiles (52 accumulator registers at NT = 13 -> 4 blocks per CU) measured equal or better than 128-row
  // tiles for NT >= 7; narrow tiles (NT <= 6) amortise the pixel-tile staging better with 128 rows.
  int MT = (NT >= 7) ? 1 : 2;
  if ((long)efm::cdiv(M, 128) * ft.nblocks < 1024) MT = 1;
  if (esize == 2) MT = pick_mt_bf16(M, ft.nblocks);
  if ((tune & 15) == 1 || (tune & 15) == 2 || ((tune & 15) == 4 && esize == 2)) MT = tune & 15;
  MT = env_int("EFM_CONV_MT", MT);
  if (esize == 4 && MT > 2) MT = 2;
  ft.MT = mt_fit(MT, NT);
  ft.NT = NT;
  return ft;
}

// Generic forward-type launch: y[m][n] = sum_k A(x)[m][k] w[n][k] + bias[n] + res[m][n]
template <typename T>
int run_fwd(const void* x, const void* w, const float* bias, const void* res, void* y, int batch,
            int hin, int win, int cin_p, int hout, int wout, int cout_p, int kh, int kw, int pad_h,
            int pad_w, int n_pad16, int k_pad, int tune, hipStream_t s, int chunk_major = 0) {
  constexpr int KS = 64 / (int)sizeof(T);  // K elements per LDS row / K step
  ConvP p;
  p.chunk_major = chunk_major;
  p.magic_taps = (unsigned)((0x100000000ULL + (unsigned)(kh * kw) - 1) / (unsigned)(kh * kw));
  p.x = x; p.w = w; p.bias = bias; p.res = res; p.y = y;
  p.M = batch * hout * wout;
  p.hin = hin; p.win = win; p.cin_p = cin_p;
  p.hout = hout; p.wout = wout; p.cout_p = cout_p;
  p.kh = kh; p.kw = kw; p.pad_h = pad_h; p.pad_w = pad_w;
  p.n_pad16 = n_pad16; p.k_pad = k_pad; p.ksteps = k_pad / KS;
  p.route = nullptr; p.cout = 0; p.ways = 0; p.order = 0; p.pool = 0; p.hp = 0; p.wp = 0; p.cn = 0; p.cpo = 0; p.out_f32 = 1;
  const FwdTiling ft = fwd_tiling(p.M, n_pad16, tune, (int)sizeof(T));
  const int NT = ft.NT, MT = ft.MT;
  p.nblocks = ft.nblocks;
  const int BM = 64 * MT;
  const long mblocks = efm::cdiv(p.M, BM);
  dim3 grid((unsigned)(mblocks * p.nblocks));
  p.magic_c = (unsigned)((0x100000000ULL + (unsigned)cin_p - 1) / (unsigned)cin_p);
  p.magic_kw = (unsigned)((0x100000000ULL + (unsigned)kw - 1) / (unsigned)kw);
  p.x_bytes = (unsigned)((size_t)batch * hin * win * cin_p * sizeof(T));
  p.w_bytes = (unsigned)((size_t)n_pad16 * k_pad * sizeof(T));
  const bool dma = sizeof(T) == 2 || env_int("EFM_CONV_DMA", 1) != 0;
  int rc;
  if (dma && sizeof(T) == 2 && MT == 4)
    rc = launch_fwd_nt<__bf16, 4, true>(NT, grid, s, p);
  else if (dma)
    rc = (MT == 2) ? launch_fwd_nt<T, 2, true>(NT, grid, s, p) : launch_fwd_nt<T, 1, true>(NT, grid, s, p);
  else
    rc = (MT == 2) ? launch_fwd_nt<float, 2, false>(NT, grid, s, p) : launch_fwd_nt<float, 1, false>(NT, grid, s, p);
  if (rc != EFM_OK) return rc;
  return efm::check_launch("conv_fwd");
}

template <int KPW>
int launch_wgrad_nt(int NTW, dim3 grid, hipStream_t s, const WgradP& p) {
  switch (NTW) {
#define EFM_CASE(N)                                             \
  case N:                                                       \
    hipLaunchKernelGGL((conv_wgrad_k<KPW, N>), grid, dim3(256), 0, s, p); \
    return EFM_OK;
    EFM_CASE(3) EFM_CASE(5) EFM_CASE(6) EFM_CASE(7) EFM_CASE(8) EFM_CASE(9) EFM_CASE(11) EFM_CASE(13)
#undef EFM_CASE
  }
  efm::set_error("conv_wgrad: unsupported NTW=%d", NTW);
  return EFM_E_INVALID;
}

struct WgradPlan {
  int KPW, NTW, kblocks, nblocks, splits, m_per_split, groups, per_group, bias_chunks, bias_groups;
  size_t slab_floats, lvl2_floats, ws_floats;  // wgrad slabs | second-level slabs | + bias partials (both levels)
};

WgradPlan plan_wgrad(const efm_conv_desc* d) {
  WgradPlan pl;
  const int M = d->batch * d->hout * d->wout;
  const int ktiles = d->k_pad / 16, ntiles = d->n_pad16 / 16;
  // 8 k-tiles per block whenever that saves a k-block: dy is re-read once per k-block, and the narrow-K layers
  // (conv1: K = 100, the 1x1 convolutions) are bound by exactly that traffic.
  const int tw = (d->tune_wgrad & 0x1000) ? 0 : d->tune_wgrad;  // bit 12 = the Winograd form's own encoding (heuristics if it is switched off)
  pl.KPW = env_int("EFM_WGRAD_KPW", (ktiles >= 5) ? 2 : 1);
  if ((tw & 15) == 1 || (tw & 15) == 2) pl.KPW = tw & 15;
  if (pl.KPW != 2) pl.KPW = 1;
  pl.kblocks = (ktiles + 4 * pl.KPW - 1) / (4 * pl.KPW);
  const int nb = (ntiles + 12) / 13;
  pl.NTW = round_nt((ntiles + nb - 1) / nb);
  pl.nblocks = (ntiles + pl.NTW - 1) / pl.NTW;
  const int base = pl.kblocks * pl.nblocks;
  // measured on EFM-29 @ B=256: ~10 blocks per CU, but never fewer than 768 pixels (48 K steps) per block
  const int target = (tw >> 4) > 0 ? 64 * (tw >> 4) : env_int("EFM_WGRAD_BLOCKS", 2560);
  int splits = (target + base - 1) / base;
  const int max_splits = (M + 767) / 768;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int mps = (M + splits - 1) / splits;
  mps = (mps + 15) & ~15;
  pl.m_per_split = mps;
  pl.splits = (M + mps - 1) / mps;
  // slabs and bias partials are reduced by ONE launch (wgrad_reduce_k), single level
  pl.per_group = pl.splits;
  pl.groups = 1;
  pl.slab_floats = (size_t)pl.splits * d->n_pad16 * d->k_pad;
  pl.lvl2_floats = 0;
  pl.bias_chunks = pl.splits;  // one column-sum partial of dy per pixel split (written by the k-block-0 blocks)
  pl.bias_groups = 0;
  pl.ws_floats = pl.slab_floats + (size_t)pl.bias_chunks * d->n_pad16;
  return pl;
}


// ==========================================================================================================
// bf16 tensor-core path (BASELINE configs[2]): bf16 NHWC activations (channel stride pad8), bf16 packed weights
// (k = tap*pad8(cin) + ci, row length pad32), fp32 accumulate, fp32 master weights / gradients in the fp32 packed layout.
// ==========================================================================================================
__host__ __device__ __forceinline__ int pad8(int c) { return (c + 7) & ~7; }
__host__ __device__ __forceinline__ int pad32(int c) { return (c + 31) & ~31; }

__global__ void __launch_bounds__(256) nchw_to_nhwc_bf16_k(const float* __restrict__ x, __bf16* __restrict__ y, long pixels,
                                                           int c, int hw, int cp) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int ng = cp >> 3;
  if (i >= pixels * ng) return;
  const long pix = i / ng;
  const int g = (int)(i - pix * ng);
  const long b = pix / hw, r = pix - b * hw;
  bf16x8 v;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ch = g * 8 + k;
    v[k] = (__bf16)((ch < c) ? x[(b * c + ch) * hw + r] : 0.f);
  }
  *reinterpret_cast<bf16x8*>(y + pix * cp + g * 8) = v;
}

// fp32 packed master weight -> bf16 forward weight wb[n][tap*cin_p8 + ci] and bf16 data-gradient weight
// wdb[ci][flip(tap)*cout_p8 + co]; one thread per destination element of either matrix.
// A fully connected layer (kernel = whole map, one output pixel: FullyConnected(513) / Dense, ref: efm_symbol.py:94) has a data
// gradient that is a plain GEMM dx[b][(tap, ci)] = sum_co dy[b][co] w[co][(tap, ci)]; run as a "full correlation" it would contract
// over taps x cout with all but one tap of every output pixel out of range (49x the work for a 7x7 map).  For such layers the
// data-gradient weight is the transpose wdb[(tap, ci)][co] and the layer runs as a 1x1 convolution cout -> taps*cin on a 1x1 map.
__host__ __device__ __forceinline__ bool fc_shaped(const efm_conv_desc& d) {
  return d.hout == 1 && d.wout == 1 && d.pad_h == 0 && d.pad_w == 0 && d.kh == d.hin && d.kw == d.win && d.kh * d.kw > 1;
}

// The bf16 data gradient of a k x k convolution (k > 1) whose output channels fill whole 32-channel K steps runs with its K ordered
// (chunk, tap, channel): measured on conv2 (192 -> 48 back to the input, 56x56, 512 images) the tap-major order re-fetched the 616 MB
// gradient TEN times from HBM (FETCH_SIZE 6.2 GB per launch: the 9 shifted reads of a pixel's 64 bytes were a whole pass over the
// channels apart, and the blocks resident on an XCD cover far more than its 4 MB of L2) — the kernel ran at HBM speed, 7.3 TB/s.
__host__ __device__ __forceinline__ bool dgrad_chunk_major(const efm_conv_desc& d) {
  return !fc_shaped(d) && d.kh * d.kw > 1 && (pad8(d.cout) & 31) == 0;
}

__global__ void __launch_bounds__(256) cast_weights_bf16_k(const float* __restrict__ w32, __bf16* __restrict__ wb,
                                                           __bf16* __restrict__ wdb, efm_conv_desc d, long nf, long nd) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int taps = d.kh * d.kw, cin8 = pad8(d.cin), cout8 = pad8(d.cout);
  if (i < nf) {
    const int kp = pad32(taps * cin8);
    const int n = (int)(i / kp), k = (int)(i - (long)n * kp);
    const int tap = k / cin8, ci = k - tap * cin8;
    float v = 0.f;
    if (n < d.cout && tap < taps && ci < d.cin) v = w32[(long)n * d.k_pad + tap * d.cin_p + ci];
    wb[i] = (__bf16)v;
  } else if (i < nf + nd && wdb != nullptr && fc_shaped(d)) {
    const long e = i - nf;
    const int kp = pad32(cout8);
    const int n = (int)(e / kp), co = (int)(e - (long)n * kp);
    const int tap = n / cin8, ci = n - tap * cin8;
    float v = 0.f;
    if (tap < taps && ci < d.cin && co < d.cout) v = w32[(long)co * d.k_pad + tap * d.cin_p + ci];
    wdb[e] = (__bf16)v;
  } else if (i < nf + nd && wdb != nullptr) {
    const long e = i - nf;
    const int kp = pad32(taps * cout8);
    const int ci = (int)(e / kp), k = (int)(e - (long)ci * kp);
    int tap = k / cout8, co = k - tap * cout8;
    if (dgrad_chunk_major(d)) {   // k = ((chunk * taps) + tap) * 32 + channel of the chunk
      const int t = k >> 5, chunk = t / taps;
      tap = t - chunk * taps;
      co = chunk * 32 + (k & 31);
    }
    float v = 0.f;
    if (ci < d.cin && tap < taps && co < d.cout) {
      const int fkh = tap / d.kw, fkw = tap - fkh * d.kw;
      v = w32[(long)co * d.k_pad + ((d.kh - 1 - fkh) * d.kw + (d.kw - 1 - fkw)) * d.cin_p + ci];
    }
    wdb[e] = (__bf16)v;
  }
}

// ---- bf16 weight gradient: C[n][k] = sum_pixels dy[pix][n] * A[pix][k], contraction (MFMA K) = 32 pixels per step.
// The tiles arrive by LDS-DMA as [pixel][channel] rows; the MFMA wants 8 consecutive PIXELS of one channel per lane, i.e. the
// transposed image: ds_read_b64_tr_b16 delivers exactly that (4 pixels x 16 channels per 16-lane group, column-major).
#ifndef WGB_BP
#define WGB_BP 32  // pixels per contraction step of the bf16 weight gradient (one MFMA K); 64 measured 50 % slower (LightCNN-9 wgrad 2.3 -> 3.5 ms)
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct WgradBP {
  const __bf16* x;
  const __bf16* dy;
  float* ws;         // slabs [split][n_pad16][kb_pad]
  float* bias_part;  // [chunks][n_pad16] or nullptr
  int M, hin, win, cin_p, hout, wout, cout_p, kh, kw, pad_h, pad_w;
  int n_pad16, kb_pad;
  int kblocks, nblocks, splits, m_per_split, mma_blocks;
  unsigned x_bytes, y_bytes;
};

template <int KPW, int NTW>
__device__ __forceinline__ void convb_wgrad_body(const WgradBP& p, __bf16* smem) {
  constexpr int BKR = 64 * KPW, BNW = 16 * NTW, BP = WGB_BP;
  constexpr int K8 = BKR / 8, N8 = BNW / 8;          // 16-byte pieces per pixel row
  constexpr int PX = (BP * K8 + 255) / 256, PY = (BP * N8 + 255) / 256;
  constexpr int TILE = BP * (BKR + BNW);              // bf16 elements per stage
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware bijective remap over the matrix-core blocks (workgroups go round-robin over the 8 XCDs, each with its own L2): the
  // kblocks*nblocks tiles of one pixel split read the same x / dy rows, so consecutive logical blocks share one XCD
  int bid = blockIdx.x;
  {
    const int nwg = p.mma_blocks;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  const int tiles = p.kblocks * p.nblocks;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int kb = bid / p.nblocks, nb = bid - kb * p.nblocks;
  const int k0 = kb * BKR, n0 = nb * BNW;
  const int m_begin = split * p.m_per_split;
  const int m_end = min(p.M, m_begin + p.m_per_split);
  const int hw = p.hout * p.wout;
  const float inv_hw = 1.0f / (float)hw, inv_w = 1.0f / (float)p.wout;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.dy), 0, p.y_bytes, 0x00020000);
  const int taps = p.kh * p.kw;

  auto load_tile = [&](int step, int buf) {
    __bf16* Xs = smem + buf * TILE;
    __bf16* Ys = Xs + BP * BKR;
    const int mbase = m_begin + step * BP;
#pragma unroll
    for (int j = 0; j < PX; ++j) {
      if (256 * j + 64 * wave < BP * K8) {  // wave-uniform
        const int e = tid + 256 * j;
        const int pp = e / K8, piece = e - pp * K8;
        const int kg = k0 + piece * 8;
        const int tap = kg / p.cin_p, c = kg - tap * p.cin_p;
        const int tkh = tap / p.kw, tkw = tap - tkh * p.kw;
        const int m = mbase + pp;
        const int b = fdiv(m, hw, inv_hw), r = m - b * hw;
        const int ho = fdiv(r, p.wout, inv_w), wo = r - ho * p.wout;
        const int hi = ho - p.pad_h + tkh, wi = wo - p.pad_w + tkw;
        const bool v = tap < taps && m < m_end && (unsigned)hi < (unsigned)p.hin && (unsigned)wi < (unsigned)p.win;
        const unsigned off = v ? (unsigned)((((b * p.hin + hi) * p.win + wi) * p.cin_p + c) * 2) : EFM_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(Xs + (256 * j + 64 * wave) * 8), 16, off, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < PY; ++j) {
      if (256 * j + 64 * wave < BP * N8) {
        const int e = tid + 256 * j;
        const int pp = e / N8, piece = e - pp * N8;
        const int m = mbase + pp, n = n0 + piece * 8;
        const bool v = m < m_end && n < p.cout_p;
        const unsigned off = v ? (unsigned)((m * p.cout_p + n) * 2) : EFM_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (__attribute__((address_space(3))) void*)(Ys + (256 * j + 64 * wave) * 8), 16, off, 0, 0, 0);
      }
    }
  };

  f32x4 acc[KPW][NTW];
#pragma unroll
  for (int a = 0; a < KPW; ++a)
#pragma unroll
    for (int b = 0; b < NTW; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // transposed fragment: lane = 16*q + 4*r + c4 supplies the address of pixel row (8q + 4h + r), channels 4*c4..4*c4+3 of the
  // 16-channel tile; it receives channel (lane & 15)'s 4 pixels.  Two reads (h = 0, 1) = the 8 K values of one MFMA operand.
  const int fq = lane >> 4, fr = (lane >> 2) & 3, fc4 = lane & 3, fi = lane & 15;
  auto frag = [&](const __bf16* tile, int row_elems, int ch0) -> bf16x8 {
    const __bf16* a0 = tile + (8 * fq + fr) * row_elems + ch0 + 4 * fc4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * row_elems));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  auto compute = [&](int buf) {
    const __bf16* Xs = smem + buf * TILE;
    const __bf16* Ys = Xs + BP * BKR;
#pragma unroll
    for (int ph = 0; ph < BP / 32; ++ph) {  // 32 pixels (one MFMA K) at a time
      bf16x8 bx[KPW];
#pragma unroll
      for (int kt = 0; kt < KPW; ++kt) bx[kt] = frag(Xs + ph * 32 * BKR, BKR, (wave * KPW + kt) * 16);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const bf16x8 ay = frag(Ys + ph * 32 * BNW, BNW, nt * 16);
#pragma unroll
        for (int kt = 0; kt < KPW; ++kt) acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ay, bx[kt], acc[kt][nt], 0, 0, 0);
      }
    }
  };

  const int steps = (m_end - m_begin + BP - 1) / BP;
  if (steps > 0) load_tile(0, 0);
  __syncthreads();
  for (int t = 0; t < steps; ++t) {
    if (t + 1 < steps) load_tile(t + 1, (t + 1) & 1);
    compute(t & 1);
    __syncthreads();
  }

  float* ws = p.ws + (long)split * p.n_pad16 * p.kb_pad;
  const int rq = lane >> 4;
#pragma unroll
  for (int kt = 0; kt < KPW; ++kt) {
    const int k = k0 + (wave * KPW + kt) * 16 + fi;
    if (k < p.kb_pad) {
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + nt * 16 + rq * 4 + r;
          if (n < p.n_pad16) ws[(long)n * p.kb_pad + k] = acc[kt][nt][r];
        }
      }
    }
  }
}

__device__ __forceinline__ void bias_colsum_bf16_body(const WgradBP& p, float* red, int chunk) {
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int m_begin = chunk * BIAS_ROWS, m_end = min(p.M, m_begin + BIAS_ROWS);
  const int ng = p.cout_p >> 2, ng16 = p.n_pad16 >> 2;  // groups of 4 channels (8 bytes of bf16)
  for (int g0 = 0; g0 < ng16; g0 += 64) {
    const int g = g0 + cx;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (g < ng)
      for (int m = m_begin + ry; m < m_end; m += 4) {
        const s16x4 v = *reinterpret_cast<const s16x4*>(p.dy + (long)m * p.cout_p + g * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += __builtin_bit_cast(float, ((unsigned)(unsigned short)v[k]) << 16);
      }
    *reinterpret_cast<f32x4*>(red + (ry * 64 + cx) * 4) = s;
    __syncthreads();
    if (ry == 0 && g < ng16) {
      f32x4 t = *reinterpret_cast<f32x4*>(red + cx * 4);
      t += *reinterpret_cast<f32x4*>(red + (64 + cx) * 4);
      t += *reinterpret_cast<f32x4*>(red + (128 + cx) * 4);
      t += *reinterpret_cast<f32x4*>(red + (192 + cx) * 4);
      *reinterpret_cast<f32x4*>(p.bias_part + (long)chunk * p.n_pad16 + g * 4) = t;
    }
    __syncthreads();
  }
}

template <int KPW, int NTW>
__global__ void __launch_bounds__(256, 2) convb_wgrad_k(const WgradBP p) {
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * WGB_BP * (64 * KPW + 16 * NTW)];
  if ((int)blockIdx.x >= p.mma_blocks)
    bias_colsum_bf16_body(p, reinterpret_cast<float*>(smem), (int)blockIdx.x - p.mma_blocks);
  else
    convb_wgrad_body<KPW, NTW>(p, smem);
}

// sum the slabs (bf16 K layout, channel stride cin_p8) into the fp32 packed gradient (channel stride cin_p4)
__global__ void __launch_bounds__(256) slab_reduce_remap_k(const float* __restrict__ ws, float* __restrict__ out, efm_conv_desc d,
                                                           int kb_pad, int splits, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)d.n_pad16 * d.k_pad;
  if (i >= total) return;
  const int n = (int)(i / d.k_pad), k = (int)(i - (long)n * d.k_pad);
  const int tap = k / d.cin_p, ci = k - tap * d.cin_p;
  float s = 0.f;
  if (tap < d.kh * d.kw) {
    const long src = (long)n * kb_pad + tap * pad8(d.cin) + ci;
    const long stride = (long)d.n_pad16 * kb_pad;
    for (int sp = 0; sp < splits; ++sp) s += ws[sp * stride + src];
  }
  out[i] = accumulate ? out[i] + s : s;
}

// The bf16 weight gradient's reduction as ONE launch: blocks [0, gx_w) sum the `nslabs` slabs (bf16 K layout, channel stride cin_p8)
// into the fp32 packed gradient (channel stride cin_p4) — 64 outputs x 4 slab lanes per block, lane l takes slabs l, l+4, ... in two
// chains, the lanes are added in index order —, blocks [gx_w, ...) sum the `chunks` bias partials (reduce_part).  Fixed order,
// no atomics.  (Was: a first level over groups of slabs, the remapping second level, and two levels for the bias: 4 launches per
// layer, 28 small launches per LightCNN-9 step on the weight-gradient stream.)
__global__ void __launch_bounds__(256) wgradb_reduce_k(const float* __restrict__ ws, float* __restrict__ out, efm_conv_desc d, int kb_pad,
                                                       int nslabs, const float* __restrict__ bpart, float* __restrict__ dbias, long n4b,
                                                       int chunks, int accumulate, int gx_w) {
  __shared__ __attribute__((aligned(16))) float red[256 * 4];
  if ((int)blockIdx.x >= gx_w) {
    reduce_part<16, 16>(bpart, dbias, n4b, chunks, accumulate, (int)blockIdx.x - gx_w, red);
    return;
  }
  const int cx = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + cx;
  const long total = (long)d.n_pad16 * d.k_pad;
  float s0 = 0.f, s1 = 0.f;
  if (i < total) {
    const int n = (int)(i / d.k_pad), k = (int)(i - (long)n * d.k_pad);
    const int tap = k / d.cin_p, ci = k - tap * d.cin_p;
    if (tap < d.kh * d.kw) {
      const float* src = ws + (long)n * kb_pad + tap * pad8(d.cin) + ci;
      const long stride = (long)d.n_pad16 * kb_pad;
      int sp = sl;
      for (; sp + 4 < nslabs; sp += 8) {
        s0 += src[sp * stride];
        s1 += src[(sp + 4) * stride];
      }
      if (sp < nslabs) s0 += src[sp * stride];
    }
  }
  red[sl * 64 + cx] = s0 + s1;
  __syncthreads();
  if (sl == 0 && i < total) {
    const float t = ((red[cx] + red[64 + cx]) + red[128 + cx]) + red[192 + cx];
    out[i] = accumulate ? out[i] + t : t;
  }
}

template <int KPW>
int launch_wgradb_nt(int NTW, dim3 grid, hipStream_t s, const WgradBP& p) {
  switch (NTW) {
#define EFM_CASE(N)                                              \
  case N:                                                        \
    hipLaunchKernelGGL((convb_wgrad_k<KPW, N>), grid, dim3(256), 0, s, p); \
    return EFM_OK;
    EFM_CASE(3) EFM_CASE(5) EFM_CASE(6) EFM_CASE(7) EFM_CASE(8) EFM_CASE(9) EFM_CASE(11) EFM_CASE(13)
#undef EFM_CASE
  }
  efm::set_error("convb_wgrad: unsupported NTW=%d", NTW);
  return EFM_E_INVALID;
}

struct WgradBPlan {
  int KPW, NTW, kblocks, nblocks, splits, m_per_split, bias_chunks, kb_pad;
  size_t slab_floats, lvl2_floats, ws_floats;
};

WgradBPlan plan_wgradb(const efm_conv_desc* d) {
  WgradBPlan pl;
  const int M = d->batch * d->hout * d->wout;
  pl.kb_pad = pad32(d->kh * d->kw * pad8(d->cin));
  const int ktiles = (pl.kb_pad + 15) / 16, ntiles = d->n_pad16 / 16;
  pl.KPW = (ktiles >= 5) ? 2 : 1;
  pl.kblocks = (ktiles + 4 * pl.KPW - 1) / (4 * pl.KPW);
  const int nb = (ntiles + 12) / 13;
  pl.NTW = round_nt((ntiles + nb - 1) / nb);
  pl.nblocks = (ntiles + pl.NTW - 1) / pl.NTW;
  const int base = pl.kblocks * pl.nblocks;
  int splits = (env_int("EFM_WGRADB_BLOCKS", 512) + base - 1) / base;  // one round of resident blocks (2 per CU): measured best (sweep 256..4096)
  const int max_splits = std::min(1024, (M + 1023) / 1024);  // >= 32 contraction steps per block; > 32 slabs reduce in two levels
  splits = std::max(1, std::min(splits, max_splits));
  int mps = (M + splits - 1) / splits;
  mps = (mps + WGB_BP - 1) / WGB_BP * WGB_BP;
  pl.m_per_split = mps;
  pl.splits = (M + mps - 1) / mps;
  pl.bias_chunks = (M + BIAS_ROWS - 1) / BIAS_ROWS;
  pl.slab_floats = (size_t)pl.splits * d->n_pad16 * pl.kb_pad;
  pl.lvl2_floats = pl.splits > 32 ? (size_t)32 * d->n_pad16 * pl.kb_pad : 0;
  pl.ws_floats = pl.slab_floats + pl.lvl2_floats + (size_t)(pl.bias_chunks + (pl.bias_chunks + 31) / 32) * d->n_pad16;
  return pl;
}

}  // namespace

namespace efm {

int wgrad_reduce(const float* slabs, float* dw, long n4w, int splits, const float* bpart, float* dbias, long n4b, int chunks, int accumulate,
                 hipStream_t s) {
  const int gx_w = (int)efm::cdiv(n4w, 64), gx_b = dbias ? (int)efm::cdiv(n4b, 16) : 0;
  hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)(gx_w + gx_b)), dim3(256), 0, s, slabs, dw, n4w, splits, bpart, dbias, n4b, chunks, accumulate,
                     gx_w);
  return efm::check_launch("conv_wgrad_reduce");
}

}  // namespace efm

extern "C" {

int efm_conv_desc_init(efm_conv_desc* d, int batch, int hin, int win, int cin, int cout, int kh, int kw,
                       int pad_h, int pad_w) {
  EFM_REQUIRE(d != nullptr, "conv_desc_init: null descriptor");
  EFM_REQUIRE(batch > 0 && hin > 0 && win > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && pad_h >= 0 && pad_w >= 0,
              "conv_desc_init: non-positive dimension");
  d->batch = batch;
  d->hin = hin; d->win = win; d->cin = cin; d->cin_p = efm_pad4(cin);
  d->kh = kh; d->kw = kw; d->pad_h = pad_h; d->pad_w = pad_w;
  d->hout = hin + 2 * pad_h - kh + 1;
  d->wout = win + 2 * pad_w - kw + 1;
  EFM_REQUIRE(d->hout > 0 && d->wout > 0, "conv_desc_init: kernel larger than padded input");
  d->cout = cout; d->cout_p = efm_pad4(cout);
  d->n_pad16 = efm_pad16(cout);
  d->k_pad = efm_pad16(kh * kw * d->cin_p);
  d->dn_pad16 = efm_pad16(cin);
  d->dk_pad = efm_pad16(kh * kw * d->cout_p);
  d->tune_fwd = 0;
  d->tune_dgrad = 0;
  d->tune_wgrad = 0;
  EFM_REQUIRE((long)batch * hin * win * d->cin_p < (1L << 30) && (long)batch * d->hout * d->wout * d->cout_p < (1L << 30),
              "conv_desc_init: tensor exceeds 2^30 elements (the launches additionally require < 2^31 BYTES per tensor)");
  EFM_REQUIRE(d->k_pad < 65536 && d->dk_pad < 65536, "conv_desc_init: K = kh*kw*channels must stay below 65536");
  EFM_REQUIRE((long)batch * d->hout * d->wout < (1L << 23) && (long)batch * hin * win < (1L << 23),
              "conv_desc_init: batch*H*W must stay below 2^23 pixels");
  return EFM_OK;
}

size_t efm_conv_weight_elems(const efm_conv_desc* d) { return (size_t)d->n_pad16 * d->k_pad; }
size_t efm_conv_dgrad_weight_elems(const efm_conv_desc* d) { return (size_t)d->dn_pad16 * d->dk_pad; }
size_t efm_conv_wgrad_workspace_bytes(const efm_conv_desc* d) {
  if (efm::wino_wgrad_selected(d)) return efm::wino_wgrad_ws_floats(d) * sizeof(float);
  return plan_wgrad(d).ws_floats * sizeof(float);
}

int efm_conv_pack_weights(const efm_conv_desc* d, const float* w_oihw, float* w_packed, void* stream) {
  EFM_REQUIRE(d && w_oihw && w_packed, "conv_pack_weights: null argument");
  const long total = (long)d->n_pad16 * d->k_pad;
  hipLaunchKernelGGL(pack_w_k, dim3((unsigned)efm::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, w_packed, *d);
  return efm::check_launch("conv_pack_weights");
}

int efm_conv_unpack_weights(const efm_conv_desc* d, const float* w_packed, float* w_oihw, void* stream) {
  EFM_REQUIRE(d && w_oihw && w_packed, "conv_unpack_weights: null argument");
  const long total = (long)d->cout * d->cin * d->kh * d->kw;
  hipLaunchKernelGGL(unpack_w_k, dim3((unsigned)efm::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_packed, w_oihw, *d);
  return efm::check_launch("conv_unpack_weights");
}

int efm_conv_make_dgrad_weights(const efm_conv_desc* d, const float* w_packed, float* wd_packed, void* stream) {
  EFM_REQUIRE(d && wd_packed && w_packed, "conv_make_dgrad_weights: null argument");
  const long total = (long)d->dn_pad16 * d->dk_pad;
  hipLaunchKernelGGL(dgrad_w_k, dim3((unsigned)efm::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_packed, wd_packed, *d);
  return efm::check_launch("conv_make_dgrad_weights");
}

int efm_conv_fwd(const efm_conv_desc* d, const float* x, const float* w_packed, const float* bias,
                 const float* residual, float* y, void* stream) {
  EFM_REQUIRE(d && x && w_packed && y, "conv_fwd: null argument");
  EFM_REQUIRE_RANGE(d, 4, "conv_fwd");
  return run_fwd<float>(x, w_packed, bias, residual, y, d->batch, d->hin, d->win, d->cin_p, d->hout, d->wout, d->cout_p,
                 d->kh, d->kw, d->pad_h, d->pad_w, d->n_pad16, d->k_pad, d->tune_fwd, (hipStream_t)stream);
}

}  // extern "C"

namespace {
struct EpiTiling {
  int nsplit, cn, NT, MT;
};

// Channel blocks of the fused conv -> MFM (-> pool) forward: each owns cn channels of every slice (ways * cn columns).  Default
// (measured on EFM-29): one block up to 13 column tiles, two above (387- and 261-channel layers: 13- / 9-tile blocks beat one
// 25- / 17-tile block).  Shared by efm_conv_mfm_fwd and efm_conv_kernel_info.
EpiTiling epi_tiling(const efm_conv_desc* d, int ways) {
  EpiTiling et;
  const int cs_all = d->cout / ways;
  int nsplit = env_int("EFM_EPI_NSPLIT", (d->tune_fwd >> 4) & 15);
  if (nsplit <= 0) nsplit = (ways * cs_all > 13 * 16) ? 2 : 1;
  nsplit = std::min(nsplit, cs_all);
  int cn = (cs_all + nsplit - 1) / nsplit;
  nsplit = (cs_all + cn - 1) / cn;
  int NT = round_nt_epi((ways * cn + 15) / 16);
  while (NT < 0 && cn > 1) {  // too wide for one block: split further
    ++nsplit;
    cn = (cs_all + nsplit - 1) / nsplit;
    nsplit = (cs_all + cn - 1) / cn;
    NT = round_nt_epi((ways * cn + 15) / 16);
  }
  et.nsplit = nsplit; et.cn = cn; et.NT = NT;
  // 128-row tiles (tune_fwd & 15 == 2) halve the weight-tile traffic per pixel: pays for the short-K layers (conv1, the 1x1s)
  et.MT = NT > 0 ? mt_fit(((d->tune_fwd & 15) == 2) ? env_int("EFM_EPI_MT", 2) : env_int("EFM_EPI_MT", 1), NT) : 1;
  return et;
}

}  // namespace

extern "C" {

int efm_conv_mfm_supported(const efm_conv_desc* d) { return d != nullptr; }

int efm_conv_mfm_fwd(const efm_conv_desc* d, const float* x, const float* w_packed, const float* bias, float* z,
                     unsigned char* route, int ways, int order, int pool, void* stream) {
  EFM_REQUIRE(d && x && w_packed && z && route, "conv_mfm_fwd: null argument");
  EFM_REQUIRE_RANGE(d, 4, "conv_mfm_fwd");
  EFM_REQUIRE((ways == 2 || ways == 3) && d->cout % ways == 0, "conv_mfm_fwd: cout=%d not divisible by ways=%d", d->cout, ways);
  EFM_REQUIRE(order == EFM_MFM_ORDER_GROUP || order == EFM_MFM_ORDER_RES, "conv_mfm_fwd: bad order %d", order);
  EFM_REQUIRE(!pool || (d->hout >= 2 && d->wout >= 2), "conv_mfm_fwd: pooling needs a map of at least 2x2");
  // channel blocks: each owns cn channels of every slice (ways * cn columns).  Default (measured on EFM-29): one block up
  // to 13 column tiles, two above (387- and 261-channel layers: 13- / 9-tile blocks beat one 25- / 17-tile block).
  const int cs_all = d->cout / ways;
  const EpiTiling et = epi_tiling(d, ways);
  const int nsplit = et.nsplit, cn = et.cn, NT = et.NT;
  EFM_REQUIRE(NT > 0, "conv_mfm_fwd: no tiling for %d output channels", d->cout);
  ConvP p;
  p.x = x; p.w = w_packed; p.bias = bias; p.res = nullptr; p.y = z;
  p.hin = d->hin; p.win = d->win; p.cin_p = d->cin_p;
  p.hout = d->hout; p.wout = d->wout; p.cout_p = d->cout_p;
  p.kh = d->kh; p.kw = d->kw; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
  p.n_pad16 = d->n_pad16; p.k_pad = d->k_pad; p.ksteps = d->k_pad / 16;
  p.nblocks = nsplit;
  p.cn = cn;
  p.chunk_major = 0; p.magic_taps = 0;
  p.cpo = efm_pad4((ways == 3) ? 2 * cs_all : cs_all);
  p.out_f32 = 1;
  p.route = route; p.cout = d->cout; p.ways = ways; p.order = order; p.pool = pool ? 1 : 0;
  p.hp = d->hout / 2; p.wp = d->wout / 2;
  p.M = pool ? d->batch * p.hp * p.wp * 4 : d->batch * d->hout * d->wout;
  p.magic_c = (unsigned)((0x100000000ULL + (unsigned)d->cin_p - 1) / (unsigned)d->cin_p);
  p.magic_kw = (unsigned)((0x100000000ULL + (unsigned)d->kw - 1) / (unsigned)d->kw);
  p.x_bytes = (unsigned)((size_t)d->batch * d->hin * d->win * d->cin_p * sizeof(float));
  p.w_bytes = (unsigned)((size_t)d->n_pad16 * d->k_pad * sizeof(float));
  // 128-row tiles (tune_fwd & 15 == 2) halve the weight-tile traffic per pixel: pays for the short-K layers (conv1, the 1x1s)
  const int MT = et.MT;
  dim3 grid((unsigned)(efm::cdiv(p.M, 64 * MT) * nsplit));
  int rc = (MT == 2) ? launch_fwd_epi<float, 2>(NT, grid, (hipStream_t)stream, p) : launch_fwd_epi<float, 1>(NT, grid, (hipStream_t)stream, p);
  if (rc != EFM_OK) return rc;
  return efm::check_launch("conv_mfm_fwd");
}

int efm_mfm_pool_bwd(const unsigned char* route, const float* dz, float* dy, int batch, int h, int w, int c, int ways,
                     int pool, void* stream) {
  EFM_REQUIRE(route && dz && dy && batch > 0 && h > 0 && w > 0, "mfm_pool_bwd: bad argument");
  EFM_REQUIRE((ways == 2 || ways == 3) && c % ways == 0, "mfm_pool_bwd: c=%d not divisible by ways=%d", c, ways);
  const int cs = c / ways, cw = cs + (efm_pad4(c) - c);
  const long items = pool ? (long)batch * ((h + 1) / 2) * ((w + 1) / 2) : (long)batch * h * w;
  const int co = (ways == 3) ? 2 * cs : cs;
  EFM_REQUIRE(items * cw < 0x100000000L, "mfm_pool_bwd: more than 2^32 elements");
  hipLaunchKernelGGL((mfm_pool_bwd_k<float, float>), dim3((unsigned)efm::cdiv(items * cw, 256)), dim3(256), 0, (hipStream_t)stream, route,
                     dz, dy, (unsigned)(items * cw), h, w, c, cs, ways, pool ? 1 : 0, efm::fastdiv(cw), efm::fastdiv((w + 1) / 2),
                     efm::fastdiv((h + 1) / 2), efm_pad4(c), efm_pad4(co));
  return efm::check_launch("mfm_pool_bwd");
}

int efm_conv_bwd_data(const efm_conv_desc* d, const float* dy, const float* wd_packed, const float* add,
                      float* dx, void* stream) {
  EFM_REQUIRE(d && dy && wd_packed && dx, "conv_bwd_data: null argument");
  EFM_REQUIRE_RANGE(d, 4, "conv_bwd_data");
  // full correlation of dy with the flipped kernel: pad' = k - 1 - pad
  return run_fwd<float>(dy, wd_packed, nullptr, add, dx, d->batch, d->hout, d->wout, d->cout_p, d->hin, d->win, d->cin_p,
                 d->kh, d->kw, d->kh - 1 - d->pad_h, d->kw - 1 - d->pad_w, d->dn_pad16, d->dk_pad, d->tune_dgrad, (hipStream_t)stream);
}

// The weight gradient is two launches — the split-over-pixels matrix-core kernel that writes slabs (+ bias partials) into the
// workspace, and the fixed-order reduction of those slabs — exposed separately so that a caller can put the reduction (pure
// streaming work) on another stream, under the next layer's matrix-core kernel; efm_conv_bwd_weight runs both back to back.
int efm_conv_bwd_weight_slabs(const efm_conv_desc* d, const float* x, const float* dy, int want_bias, void* workspace,
                              size_t workspace_bytes, void* stream) {
  EFM_REQUIRE(d && x && dy, "conv_bwd_weight: null argument");
  EFM_REQUIRE_RANGE(d, 4, "conv_bwd_weight");
  if (efm::wino_wgrad_selected(d)) return efm::wino_wgrad_slabs(d, x, dy, want_bias, workspace, workspace_bytes, (hipStream_t)stream);
  const WgradPlan pl = plan_wgrad(d);
  if (!workspace || workspace_bytes < pl.ws_floats * sizeof(float)) {
    efm::set_error("conv_bwd_weight: workspace %zu B < required %zu B", workspace_bytes, pl.ws_floats * sizeof(float));
    return EFM_E_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  WgradP p;
  p.x = x; p.dy = dy; p.ws = (float*)workspace;
  p.M = d->batch * d->hout * d->wout;
  p.hin = d->hin; p.win = d->win; p.cin_p = d->cin_p;
  p.hout = d->hout; p.wout = d->wout; p.cout_p = d->cout_p;
  p.kh = d->kh; p.kw = d->kw; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
  p.n_pad16 = d->n_pad16; p.k_pad = d->k_pad;
  p.kblocks = pl.kblocks; p.nblocks = pl.nblocks; p.splits = pl.splits; p.m_per_split = pl.m_per_split;
  p.x_bytes = (unsigned)((size_t)d->batch * d->hin * d->win * d->cin_p * sizeof(float));
  p.y_bytes = (unsigned)((size_t)d->batch * d->hout * d->wout * d->cout_p * sizeof(float));
  float* slabs = (float*)workspace;
  float* bpart = slabs + pl.slab_floats + pl.lvl2_floats;
  p.bias_part = want_bias ? bpart : nullptr;
  p.ktiles = d->k_pad / 16; p.ntiles = d->n_pad16 / 16; p.bias_rows = BIAS_ROWS;
#ifdef EFM_ABLATE  // measurement builds only (-DEFM_ABLATE): a production library never skips operand loads or MFMAs, whatever the environment says
  p.dbg = env_int("EFM_WGRAD_DBG", 0);
#else
  p.dbg = 0;
#endif
  p.mma_blocks = pl.kblocks * pl.nblocks * pl.splits;
  dim3 grid((unsigned)p.mma_blocks);  // the bias gradient rides in the k-block-0 blocks: no column-sum blocks
  int rc = (pl.KPW == 2) ? launch_wgrad_nt<2>(pl.NTW, grid, s, p) : launch_wgrad_nt<1>(pl.NTW, grid, s, p);
  if (rc != EFM_OK) return rc;
  return efm::check_launch("conv_wgrad");
}

int efm_conv_bwd_weight_finish(const efm_conv_desc* d, float* dw_packed, float* dbias, int accumulate, const void* workspace,
                               size_t workspace_bytes, void* stream) {
  EFM_REQUIRE(d && dw_packed, "conv_bwd_weight_finish: null argument");
  if (efm::wino_wgrad_selected(d)) return efm::wino_wgrad_finish(d, dw_packed, dbias, accumulate, workspace, workspace_bytes, (hipStream_t)stream);
  const WgradPlan pl = plan_wgrad(d);
  if (!workspace || workspace_bytes < pl.ws_floats * sizeof(float)) {
    efm::set_error("conv_bwd_weight_finish: workspace %zu B < required %zu B", workspace_bytes, pl.ws_floats * sizeof(float));
    return EFM_E_WORKSPACE;
  }
  const float* slabs = (const float*)workspace;
  const float* bpart = slabs + pl.slab_floats + pl.lvl2_floats;
  return efm::wgrad_reduce(slabs, dw_packed, (long)d->n_pad16 * d->k_pad / 4, pl.splits, bpart, dbias, d->n_pad16 / 4, pl.bias_chunks, accumulate,
                           (hipStream_t)stream);
}

int efm_conv_bwd_weight(const efm_conv_desc* d, const float* x, const float* dy, float* dw_packed, float* dbias,
                        int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  EFM_REQUIRE(d && x && dy && dw_packed, "conv_bwd_weight: null argument");
  const int rc = efm_conv_bwd_weight_slabs(d, x, dy, dbias != nullptr, workspace, workspace_bytes, stream);
  if (rc != EFM_OK) return rc;
  return efm_conv_bwd_weight_finish(d, dw_packed, dbias, accumulate, workspace, workspace_bytes, stream);
}


/* Diagnostic for roofline accounting (bench.py): which kernel instance a launch of `pass` resolves to under the descriptor's tuning
 * fields, and the matrix-core flops it EXECUTES (padded tiles; Winograd: the 16 transformed-domain GEMMs) — as opposed to the
 * algorithmic 2*M*cout*cin*kh*kw.  pass: 0 forward, 1 forward + fused MFM(ways) epilogue (`pool`), 2 data gradient, 3 weight
 * gradient, 4 / 5 / 6 = Winograd forward / fused forward / data gradient. */
int efm_conv_kernel_info(const efm_conv_desc* d, int pass, int ways, int pool, char* name, size_t name_len, double* mfma_flops) {
  EFM_REQUIRE(d && pass >= 0 && pass <= 6, "conv_kernel_info: bad argument");
  if (pass >= 4) {
    EFM_REQUIRE(efm::wino_kernel_info(d, pass, ways, name, name_len, mfma_flops) == EFM_OK, "conv_kernel_info: not a Winograd geometry");
    return EFM_OK;
  }
  const long M = (long)d->batch * d->hout * d->wout;
  double fl = 0.0;
  char buf[96];
  if (pass == 0 || pass == 2) {
    const long rows = pass == 0 ? M : (long)d->batch * d->hin * d->win;
    const FwdTiling ft = pass == 0 ? fwd_tiling(rows, d->n_pad16, d->tune_fwd, 4) : fwd_tiling(rows, d->dn_pad16, d->tune_dgrad, 4);
    const long bm = 64 * ft.MT;
    fl = 2.0 * (double)(efm::cdiv(rows, bm) * bm) * (double)(ft.nblocks * ft.NT * 16) * (double)(pass == 0 ? d->k_pad : d->dk_pad);
    snprintf(buf, sizeof(buf), "conv_fwd_k<float, %d, %d, true, 0>", ft.MT, ft.NT);
  } else if (pass == 1) {
    EFM_REQUIRE((ways == 2 || ways == 3) && d->cout % ways == 0, "conv_kernel_info: cout=%d not divisible by ways=%d", d->cout, ways);
    const EpiTiling et = epi_tiling(d, ways);
    EFM_REQUIRE(et.NT > 0, "conv_kernel_info: no fused tiling");
    const long rows = pool ? (long)d->batch * (d->hout / 2) * (d->wout / 2) * 4 : M;
    const long bm = 64 * et.MT;
    fl = 2.0 * (double)(efm::cdiv(rows, bm) * bm) * (double)(et.nsplit * et.NT * 16) * (double)d->k_pad;
    snprintf(buf, sizeof(buf), "conv_fwd_k<float, %d, %d, true, 1>", et.MT, et.NT);
  } else if (efm::wino_wgrad_selected(d)) {
    efm::wino_wgrad_info(d, buf, sizeof(buf), &fl);
  } else {
    const WgradPlan pl = plan_wgrad(d);
    fl = 2.0 * (double)((long)pl.splits * pl.m_per_split) * (double)(pl.nblocks * pl.NTW * 16) * (double)d->k_pad;
    snprintf(buf, sizeof(buf), "conv_wgrad_k<%d, %d>", pl.KPW, pl.NTW);
  }
  if (name && name_len) snprintf(name, name_len, "%s", buf);
  if (mfma_flops) *mfma_flops = fl;
  return EFM_OK;
}


// ---------------------------------------------------------------------------------------- bf16 path
size_t efm_convb_weight_elems(const efm_conv_desc* d) { return (size_t)d->n_pad16 * pad32(d->kh * d->kw * pad8(d->cin)); }
size_t efm_convb_dgrad_weight_elems(const efm_conv_desc* d) {
  if (fc_shaped(*d)) return (size_t)efm_pad16(d->kh * d->kw * pad8(d->cin)) * pad32(pad8(d->cout));  // transposed weight of the GEMM form
  return (size_t)d->dn_pad16 * pad32(d->kh * d->kw * pad8(d->cout));
}
// slabs | second-level slabs (more than 32 slabs reduce in two levels) | bias partials (both levels) of a bf16 weight gradient with
// `splits` slabs and `chunks` bias partials
static size_t wgradb_ws_floats(const efm_conv_desc* d, int kb_pad, int splits, int chunks) {
  const size_t slab = (size_t)d->n_pad16 * kb_pad;
  return slab * splits + (splits > 32 ? 32 * slab : 0) + (size_t)(chunks + (chunks + 31) / 32) * d->n_pad16;
}
size_t efm_convb_wgrad_workspace_bytes(const efm_conv_desc* d) {
  if (efm::wgrad2_selected(d)) {
    return wgradb_ws_floats(d, pad32(d->kh * d->kw * pad8(d->cin)), efm::wgrad2_splits(d), efm::wgrad2_bias_chunks(d)) * sizeof(float);
  }
  return plan_wgradb(d).ws_floats * sizeof(float);
}

int efm_nchw_to_nhwc_bf16(const float* x, uint16_t* y, int batch, int c, int h, int w, void* stream) {
  EFM_REQUIRE(x && y && batch > 0 && c > 0 && h > 0 && w > 0, "nchw_to_nhwc_bf16: bad argument");
  const long pixels = (long)batch * h * w;
  const int cp = pad8(c);
  hipLaunchKernelGGL(nchw_to_nhwc_bf16_k, dim3((unsigned)efm::cdiv(pixels * (cp >> 3), 256)), dim3(256), 0, (hipStream_t)stream, x,
                     reinterpret_cast<__bf16*>(y), pixels, c, h * w, cp);
  return efm::check_launch("nchw_to_nhwc_bf16");
}

int efm_convb_cast_weights(const efm_conv_desc* d, const float* w_packed, uint16_t* wb, uint16_t* wdb, void* stream) {
  EFM_REQUIRE(d && w_packed && wb, "convb_cast_weights: null argument");
  const long nf = (long)efm_convb_weight_elems(d), nd = wdb ? (long)efm_convb_dgrad_weight_elems(d) : 0;
  hipLaunchKernelGGL(cast_weights_bf16_k, dim3((unsigned)efm::cdiv(nf + nd, 256)), dim3(256), 0, (hipStream_t)stream, w_packed,
                     reinterpret_cast<__bf16*>(wb), reinterpret_cast<__bf16*>(wdb), *d, nf, nd);
  return efm::check_launch("convb_cast_weights");
}

int efm_convb_fwd(const efm_conv_desc* d, const uint16_t* x, const uint16_t* wb, const float* bias, const uint16_t* residual,
                  uint16_t* y, void* stream) {
  EFM_REQUIRE(d && x && wb && y, "convb_fwd: null argument");
  EFM_REQUIRE_RANGE(d, 2, "convb_fwd");
  return run_fwd<__bf16>(x, wb, bias, residual, y, d->batch, d->hin, d->win, pad8(d->cin), d->hout, d->wout, pad8(d->cout), d->kh, d->kw,
                         d->pad_h, d->pad_w, d->n_pad16, pad32(d->kh * d->kw * pad8(d->cin)), d->tune_fwd, (hipStream_t)stream);
}

int efm_convb_bwd_data(const efm_conv_desc* d, const uint16_t* dy, const uint16_t* wdb, const uint16_t* add, uint16_t* dx, void* stream) {
  EFM_REQUIRE(d && dy && wdb && dx, "convb_bwd_data: null argument");
  EFM_REQUIRE_RANGE(d, 2, "convb_bwd_data");
  if (fc_shaped(*d)) {  // fully connected: dx[b][(tap, ci)] = dy[b][:] . wdb[(tap, ci)][:] — a 1x1 convolution cout -> taps*cin on a 1x1 map
    const int nrow = d->kh * d->kw * pad8(d->cin);
    return run_fwd<__bf16>(dy, wdb, nullptr, add, dx, d->batch, 1, 1, pad8(d->cout), 1, 1, nrow, 1, 1, 0, 0, efm_pad16(nrow), pad32(pad8(d->cout)),
                           d->tune_dgrad, (hipStream_t)stream);
  }
  return run_fwd<__bf16>(dy, wdb, nullptr, add, dx, d->batch, d->hout, d->wout, pad8(d->cout), d->hin, d->win, pad8(d->cin), d->kh, d->kw,
                         d->kh - 1 - d->pad_h, d->kw - 1 - d->pad_w, d->dn_pad16, pad32(d->kh * d->kw * pad8(d->cout)), d->tune_dgrad,
                         (hipStream_t)stream, dgrad_chunk_major(*d) ? 1 : 0);
}

int efm_convb_mfm_fwd(const efm_conv_desc* d, const uint16_t* x, const uint16_t* wb, const float* bias, void* z, unsigned char* route,
                      int ways, int order, int pool, int out_f32, void* stream) {
  EFM_REQUIRE(d && x && wb && z && route, "convb_mfm_fwd: null argument");
  EFM_REQUIRE_RANGE(d, 2, "convb_mfm_fwd");
  EFM_REQUIRE((ways == 2 || ways == 3) && d->cout % ways == 0, "convb_mfm_fwd: cout=%d not divisible by ways=%d", d->cout, ways);
  EFM_REQUIRE(order == EFM_MFM_ORDER_GROUP || order == EFM_MFM_ORDER_RES, "convb_mfm_fwd: bad order %d", order);
  const int cs_all = d->cout / ways;
  int nsplit = (d->tune_fwd >> 4) & 15;
  if (nsplit <= 0) nsplit = (ways * cs_all > 13 * 16) ? 2 : 1;
  nsplit = std::min(nsplit, cs_all);
  int cn = (cs_all + nsplit - 1) / nsplit;
  nsplit = (cs_all + cn - 1) / cn;
  int NT = round_nt_epi((ways * cn + 15) / 16);
  while (NT < 0 && cn > 1) {
    ++nsplit;
    cn = (cs_all + nsplit - 1) / nsplit;
    nsplit = (cs_all + cn - 1) / cn;
    NT = round_nt_epi((ways * cn + 15) / 16);
  }
  EFM_REQUIRE(NT > 0, "convb_mfm_fwd: no tiling for %d output channels", d->cout);
  const int cin8 = pad8(d->cin), kp = pad32(d->kh * d->kw * cin8), co = (ways == 3) ? 2 * cs_all : cs_all;
  ConvP p;
  p.x = x; p.w = wb; p.bias = bias; p.res = nullptr; p.y = z;
  p.hin = d->hin; p.win = d->win; p.cin_p = cin8;
  p.hout = d->hout; p.wout = d->wout; p.cout_p = pad8(d->cout);
  p.kh = d->kh; p.kw = d->kw; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
  p.n_pad16 = d->n_pad16; p.k_pad = kp; p.ksteps = kp / 32;
  p.nblocks = nsplit; p.cn = cn;
  p.chunk_major = 0; p.magic_taps = 0;
  p.cpo = out_f32 ? efm_pad4(co) : pad8(co);
  p.out_f32 = out_f32 ? 1 : 0;
  p.route = route; p.cout = d->cout; p.ways = ways; p.order = order; p.pool = pool ? 1 : 0;
  p.hp = d->hout / 2; p.wp = d->wout / 2;
  p.M = pool ? d->batch * p.hp * p.wp * 4 : d->batch * d->hout * d->wout;
  p.magic_c = (unsigned)((0x100000000ULL + (unsigned)cin8 - 1) / (unsigned)cin8);
  p.magic_kw = (unsigned)((0x100000000ULL + (unsigned)d->kw - 1) / (unsigned)d->kw);
  p.x_bytes = (unsigned)((size_t)d->batch * d->hin * d->win * cin8 * 2);
  p.w_bytes = (unsigned)((size_t)d->n_pad16 * kp * 2);
  int MT = pick_mt_bf16(p.M, nsplit);
  if ((d->tune_fwd & 15) == 1 || (d->tune_fwd & 15) == 2 || (d->tune_fwd & 15) == 4) MT = d->tune_fwd & 15;
  MT = mt_fit(env_int("EFM_CONV_MT", MT), NT);
  dim3 grid((unsigned)(efm::cdiv(p.M, 64 * MT) * nsplit));
  int rc = (MT == 4)   ? launch_fwd_epi<__bf16, 4>(NT, grid, (hipStream_t)stream, p)
           : (MT == 2) ? launch_fwd_epi<__bf16, 2>(NT, grid, (hipStream_t)stream, p)
                       : launch_fwd_epi<__bf16, 1>(NT, grid, (hipStream_t)stream, p);
  if (rc != EFM_OK) return rc;
  return efm::check_launch("convb_mfm_fwd");
}

}  // extern "C"

namespace {
// Vector form of mfm_pool_bwd_k for bf16 gradients whose slice width is a multiple of 8 channels (every MFM2 / MFM3 layer of
// LightCNN-9, the deeper CNN and most of EFM-29's bf16 plan): thread = (pooling window | pixel, group of 8 channels): one 8-byte
// route load and one 16-byte (bf16) / 32-byte (fp32) dz load per slice-half, then a 16-byte store per (window pixel, slice) —
// the scalar kernel's 2-byte stores ran this pure streaming kernel at ~1 TB/s.  Same arithmetic, same bits.
__device__ __forceinline__ void load8(const __bf16* p, float* out) {  // 16-byte aligned
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) out[e] = (float)v[e];
}
__device__ __forceinline__ void load8(const float* p, float* out) {  // 16-byte aligned (32 bytes in two pieces)
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { out[e] = a[e]; out[4 + e] = b[e]; }
}

template <typename TZ>
__global__ void __launch_bounds__(256) mfm_pool_bwd_v8_k(const unsigned char* __restrict__ route, const TZ* __restrict__ dz,
                                                         __bf16* __restrict__ dy, long items, int h, int w, int c, int ways, int pool,
                                                         int cp, int cpo) {
  const int cs = c / ways, g8 = cs >> 3;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= items * g8) return;
  const long it = i / g8;
  const int j = (int)(i - it * g8) * 8;
  float gmax[8], gmin[8];
  unsigned char rmax[8], rmin[8];
  auto load = [&](long q) {
    const uint2 r = *reinterpret_cast<const uint2*>(route + q * cpo + j);
    *reinterpret_cast<uint2*>(rmax) = r;
    load8(dz + q * cpo + j, gmax);
    if (ways == 3) {
      *reinterpret_cast<uint2*>(rmin) = *reinterpret_cast<const uint2*>(route + q * cpo + cs + j);
      load8(dz + q * cpo + cs + j, gmin);
    }
  };
  const bf16x8 zero = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
  if (!pool) {
    load(it);
    __bf16* d = dy + it * cp;
    for (int sl = 0; sl < ways; ++sl) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = (rmax[e] == sl) ? gmax[e] : 0.f;
        if (ways == 3 && rmin[e] == sl) v += gmin[e];
        o[e] = (__bf16)v;
      }
      *reinterpret_cast<bf16x8*>(d + sl * cs + j) = o;
    }
    return;
  }
  const int hg = (h + 1) >> 1, wg = (w + 1) >> 1, hp = h >> 1, wp = w >> 1;
  const int wq = (int)(it % wg);
  const long t = it / wg;
  const int hq = (int)(t % hg);
  const long b = t / hg;
  const bool full = hq < hp && wq < wp;
  if (full) load((b * hp + hq) * wp + wq);
#pragma unroll
  for (int px = 0; px < 4; ++px) {
    const int hh = 2 * hq + (px >> 1), ww = 2 * wq + (px & 1);
    if (hh >= h || ww >= w) continue;
    __bf16* d = dy + ((b * h + hh) * w + ww) * cp;
    for (int sl = 0; sl < ways; ++sl) {
      bf16x8 o = zero;
      if (full) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = (rmax[e] == px * 4 + sl) ? gmax[e] : 0.f;
          if (ways == 3 && rmin[e] == px * 4 + sl) v += gmin[e];
          o[e] = (__bf16)v;
        }
      }
      *reinterpret_cast<bf16x8*>(d + sl * cs + j) = o;
    }
  }
}
}  // namespace

extern "C" {

int efm_convb_mfm_pool_bwd(const unsigned char* route, const void* dz, int dz_f32, uint16_t* dy, int batch, int h, int w, int c, int ways,
                           int pool, void* stream) {
  EFM_REQUIRE(route && dz && dy && batch > 0 && h > 0 && w > 0, "convb_mfm_pool_bwd: bad argument");
  EFM_REQUIRE((ways == 2 || ways == 3) && c % ways == 0, "convb_mfm_pool_bwd: c=%d not divisible by ways=%d", c, ways);
  const int cs = c / ways, cp = pad8(c), cw = cs + (cp - c), co = (ways == 3) ? 2 * cs : cs;
  const long items = pool ? (long)batch * ((h + 1) / 2) * ((w + 1) / 2) : (long)batch * h * w;
  if (cs % 8 == 0 && env_int("EFM_POOLBWD_V8", 1)) {  // c = ways*cs is then a multiple of 8 too: no pad channels, 16-byte stores
    dim3 gv((unsigned)efm::cdiv(items * (cs / 8), 256));
    if (dz_f32)
      hipLaunchKernelGGL((mfm_pool_bwd_v8_k<float>), gv, dim3(256), 0, (hipStream_t)stream, route, (const float*)dz,
                         reinterpret_cast<__bf16*>(dy), items, h, w, c, ways, pool ? 1 : 0, cp, efm_pad4(co));
    else
      hipLaunchKernelGGL((mfm_pool_bwd_v8_k<__bf16>), gv, dim3(256), 0, (hipStream_t)stream, route, (const __bf16*)dz,
                         reinterpret_cast<__bf16*>(dy), items, h, w, c, ways, pool ? 1 : 0, cp, pad8(co));
    return efm::check_launch("convb_mfm_pool_bwd");
  }
  EFM_REQUIRE(items * cw < 0x100000000L, "convb_mfm_pool_bwd: more than 2^32 elements");
  dim3 grid((unsigned)efm::cdiv(items * cw, 256));
  const efm::FastDiv cwd = efm::fastdiv(cw), wgd = efm::fastdiv((w + 1) / 2), hgd = efm::fastdiv((h + 1) / 2);
  if (dz_f32)
    hipLaunchKernelGGL((mfm_pool_bwd_k<float, __bf16>), grid, dim3(256), 0, (hipStream_t)stream, route, (const float*)dz,
                       reinterpret_cast<__bf16*>(dy), (unsigned)(items * cw), h, w, c, cs, ways, pool ? 1 : 0, cwd, wgd, hgd, cp, efm_pad4(co));
  else
    hipLaunchKernelGGL((mfm_pool_bwd_k<__bf16, __bf16>), grid, dim3(256), 0, (hipStream_t)stream, route, (const __bf16*)dz,
                       reinterpret_cast<__bf16*>(dy), (unsigned)(items * cw), h, w, c, cs, ways, pool ? 1 : 0, cwd, wgd, hgd, cp, pad8(co));
  return efm::check_launch("convb_mfm_pool_bwd");
}

static int convb_bwd_weight_impl(const efm_conv_desc* d, const uint16_t* x, const uint16_t* dy, const void* dz, const unsigned char* route,
                                 float* dw_packed, float* dbias, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const bool halo = efm::wgrad2_selected(d);   // the halo-tile form (efm_convb_wgrad.hip) wherever it applies
  WgradBPlan pl;
  if (halo) {
    pl.splits = efm::wgrad2_splits(d);
    pl.bias_chunks = efm::wgrad2_bias_chunks(d);
    pl.kb_pad = pad32(d->kh * d->kw * pad8(d->cin));
    pl.slab_floats = (size_t)pl.splits * d->n_pad16 * pl.kb_pad;
    pl.lvl2_floats = pl.splits > 32 ? (size_t)32 * d->n_pad16 * pl.kb_pad : 0;
    pl.ws_floats = wgradb_ws_floats(d, pl.kb_pad, pl.splits, pl.bias_chunks);
  } else {
    pl = plan_wgradb(d);
  }
  if (!workspace || workspace_bytes < pl.ws_floats * sizeof(float)) {
    efm::set_error("convb_bwd_weight: workspace %zu B < required %zu B", workspace_bytes, pl.ws_floats * sizeof(float));
    return EFM_E_WORKSPACE;
  }
  float* lvl2 = (float*)workspace + pl.slab_floats;
  float* bpart = lvl2 + pl.lvl2_floats;
  float* bpart2 = bpart + (size_t)pl.bias_chunks * d->n_pad16;
  int rc;
  if (halo) {
    rc = efm::wgrad2_slabs(d, x, dy, dz, route, (float*)workspace, dbias ? bpart : nullptr, s);
    if (rc != EFM_OK) return rc;
  } else {
    WgradBP p;
    p.x = reinterpret_cast<const __bf16*>(x); p.dy = reinterpret_cast<const __bf16*>(dy); p.ws = (float*)workspace;
    p.M = d->batch * d->hout * d->wout;
    p.hin = d->hin; p.win = d->win; p.cin_p = pad8(d->cin);
    p.hout = d->hout; p.wout = d->wout; p.cout_p = pad8(d->cout);
    p.kh = d->kh; p.kw = d->kw; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    p.n_pad16 = d->n_pad16; p.kb_pad = pl.kb_pad;
    p.kblocks = pl.kblocks; p.nblocks = pl.nblocks; p.splits = pl.splits; p.m_per_split = pl.m_per_split;
    p.x_bytes = (unsigned)((size_t)d->batch * d->hin * d->win * p.cin_p * 2);
    p.y_bytes = (unsigned)((size_t)d->batch * d->hout * d->wout * p.cout_p * 2);
    p.bias_part = dbias ? bpart : nullptr;
    p.mma_blocks = pl.kblocks * pl.nblocks * pl.splits;
    dim3 grid((unsigned)(p.mma_blocks + (dbias ? pl.bias_chunks : 0)));
    rc = (pl.KPW == 2) ? launch_wgradb_nt<2>(pl.NTW, grid, s, p) : launch_wgradb_nt<1>(pl.NTW, grid, s, p);
    if (rc != EFM_OK) return rc;
    rc = efm::check_launch("convb_wgrad");
    if (rc != EFM_OK) return rc;
  }
  const long total = (long)d->n_pad16 * d->k_pad;
  (void)lvl2; (void)bpart2;
  const int gx_w = (int)efm::cdiv(total, 64), gx_b = dbias ? (int)efm::cdiv((long)d->n_pad16 / 4, 16) : 0;
  hipLaunchKernelGGL(wgradb_reduce_k, dim3((unsigned)(gx_w + gx_b)), dim3(256), 0, s, (const float*)workspace, dw_packed, *d, pl.kb_pad, pl.splits,
                     (const float*)bpart, dbias, (long)d->n_pad16 / 4, pl.bias_chunks, accumulate, gx_w);
  return efm::check_launch("convb_wgrad_reduce");
}

int efm_convb_bwd_weight(const efm_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw_packed, float* dbias, int accumulate,
                         void* workspace, size_t workspace_bytes, void* stream) {
  EFM_REQUIRE(d && x && dy && dw_packed, "convb_bwd_weight: null argument");
  EFM_REQUIRE_RANGE(d, 2, "convb_bwd_weight");
  return convb_bwd_weight_impl(d, x, dy, nullptr, nullptr, dw_packed, dbias, accumulate, workspace, workspace_bytes, stream);
}

int efm_convb_mfm_bwd_weight_supported(const efm_conv_desc* d, int ways, int pool) {
  return d && efm::wgrad2_expand_supported(d, ways, pool) ? 1 : 0;
}

int efm_convb_mfm_bwd_weight(const efm_conv_desc* d, const uint16_t* x, const unsigned char* route, const uint16_t* dz, int ways, int pool,
                             float* dw_packed, float* dbias, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  EFM_REQUIRE(d && x && route && dz && dw_packed, "convb_mfm_bwd_weight: null argument");
  EFM_REQUIRE_RANGE(d, 2, "convb_mfm_bwd_weight");
  EFM_REQUIRE(efm::wgrad2_expand_supported(d, ways, pool), "convb_mfm_bwd_weight: layer / epilogue not supported (ask efm_convb_mfm_bwd_weight_supported)");
  return convb_bwd_weight_impl(d, x, nullptr, dz, route, dw_packed, dbias, accumulate, workspace, workspace_bytes, stream);
}

}  // extern "C"
