// bf16 weight gradient, halo-tile form (gfx950): dw[n][(tap, ci)] = sum_pixels dy[pix][n] * x[pix + tap][ci]
// (the weight gradient of mx.symbol.Convolution, ref: efm_symbol.py:32,41,54,62,65,67 — MXNet dispatches it to cuDNN).
//
// Why a second form next to convb_wgrad_k (efm_conv.hip).  A bf16 MFMA retires 8x the MACs of the fp32 one per operand byte, so
// everything AROUND the matrix cores decides: the im2col form stages a [32 pixels][128 k-columns] tile per 26 MFMAs of a wave — the x
// operand is fetched 9x (once per tap), dy once per k-block (4x on conv2), every 32-pixel step pays a block barrier and ~5 LDS-DMA
// instructions per wave (~100 issue cycles each) for 416 cycles of MFMA.  Measured 240-490 TFLOP/s (10-20 % of peak).  Here:
//   * a block owns a WHOLE [n chunks][k chunks] tile of the gradient (up to 12 x 28 or 8 x 36 tiles of 16 x 16: all 9 taps of
//     conv2, one kernel row of the wider layers) in the accumulators of its 8 waves (2 x 4 over n x k, 144-168 registers each),
//     and streams over its share of the pixels: dy is read once per tap group, x once per n part;
//   * a stage = 4 x 16 output pixels of one image: the input arrives ONCE as a halo tile ((4 + rows - 1) x (16 + kw - 1) pixels), the
//     taps are formed by shifted reads from LDS; per stage and wave 2 x (TN x TK) MFMAs (>= 1000 cycles) per barrier;
//   * LDS images are [16-channel chunk][pixel row][32 B], filled by LDS-DMA (16 B per lane, source address per lane, padding and
//     image borders by the buffer range check).  The MFMA wants 8 consecutive PIXELS of one channel per lane: ds_read_b64_tr_b16
//     transposes on the way out.  A 32-lane half of such a read touches pixels {0..3, 8..11} (+4, +16) of one row: the pixel -> LDS
//     row map swaps bits 2 and 3 of the column so that those 8 rows are 8 consecutive 32-byte slots — conflict free;
//   * the bias gradient (column sums of dy) rides on the dy fragments of the k-group-0 waves (v_dot2_f32_bf16 against ones);
//   * slabs and bias partials have convb_wgrad_k's layout: the fixed-order reductions (slab_reduce_k / slab_reduce_remap_k) are shared;
//   * small gradients (the row-packed first convolution: 6 x 5 tiles) do not split over the waves — a wave with a few tiles reads an
//     LDS fragment per MFMA or two — but over the PIXELS: each of the 8 waves holds the whole tile and takes 2 rows of a 16-row
//     stage (template WN x WK waves over the tile, 8 / (WN*WK) over the rows; every pixel group writes its own slab);
//   * EXPAND: the first convolution's output gradient is never materialised.  Its only consumer is this kernel (the network input
//     needs no gradient), and it is the backward of conv -> MFM2 -> 2x2 max pooling: one non-zero among the 8 (slice, window pixel)
//     positions of every (window, channel), named by the route byte.  The kernel reads dz (1/8 of dy's bytes) + the route bytes,
//     zero-fills the stage's dy image in LDS and scatters the dz values to their positions (ds_write_b16): 1.2 GB that
//     efm_convb_mfm_pool_bwd wrote and this kernel read back (LightCNN-9, 512 images) become 0.23 GB read once.
#include <algorithm>

#include "efm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define EFM_OOB 0x80000000u

constexpr int W2_THREADS = 512;
constexpr int W2_STAGE_BYTES = 61440;  // per stage: x halo image (rounded up to 1 KiB) + dy image; two stages = 120 KiB of the 160

struct Wg2P {
  const __bf16* x;
  const __bf16* dy;
  float* ws;         // slabs [split][n_pad16][kb_pad]
  float* bias_part;  // [split][n_pad16] or nullptr
  int batch, h, w, cin8, cout8;
  int kh, kw, pad_h, pad_w;
  int n_pad16, kb_pad;
  int nparts, tgroups;  // the gradient is cut into nparts (n) x tgroups (kernel rows) block tiles
  int nch, cch;         // 16-channel chunks of a block's n range / of cin
  int rows_g;           // kernel rows per tap group
  int hh, hwp;          // halo tile: rows, row pitch (16 + kw - 1)
  int x_img_bytes;      // bytes of the x image rounded up to 1 KiB (the dy image follows)
  int px, py;           // LDS-DMA pieces (16 B) of the x / dy image
  int tiles_x, tiles_y, stages, stages_per_split, splits;
  unsigned x_bytes, y_bytes;
  // EXPAND: dy = backward of the fused MFM2 + 2x2 pooling epilogue, formed in LDS from dz [b][hp][wp][cpo] + route bytes (same layout)
  const __bf16* dz;
  const unsigned char* route;
  int hp, wp, cs, cpo;
};

__device__ __forceinline__ int bitswap23(int v) { return (v & ~12) | ((v & 4) << 1) | ((v & 8) >> 1); }

// TN x TK: 16 x 16 tiles of the gradient per wave; WN x WK waves over the block's tile, WM = 8 / (WN*WK) wave groups over the rows of a
// stage; S: 32-pixel steps (2 rows) per wave and stage -> a stage is TH = 2*S*WM rows x 16 columns.
template <int TN, int TK, int WN, int WK, int S, bool EXPAND>
__device__ __forceinline__ void wg2_body(const Wg2P& p, char* smem) {
  constexpr int WM = 8 / (WN * WK), TH = 2 * S * WM;
  constexpr int YPLANE = 16 * TH * 32;   // bytes of one 16-channel chunk of the dy image
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it (tile ranges, gradient columns) stays in SGPRs
  const int wn = wave % WN, wk = (wave / WN) % WK, wm = wave / (WN * WK);
  // block -> (split, n part, tap group); consecutive LOGICAL blocks = the tiles of one split: they read the same pixels (dy once per tap
  // group, x once per n part), so they must share an L2 — workgroups go round-robin over the 8 XCDs, hence the bijective XCD-aware remap
  // (measured before it: conv3's 6 tiles per split read 1.38 GB from HBM per launch against 0.39 GB of operands)
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  }
  const int tiles = p.nparts * p.tgroups;
  const int split = bid / tiles;
  bid -= split * tiles;
  const int npart = bid / p.tgroups, tg = bid - npart * p.tgroups;
  const int row0 = tg * p.rows_g;                       // first kernel row of this tap group
  const int rows = min(p.rows_g, p.kh - row0);
  const int kch = rows * p.kw * p.cch;                  // k chunks of this block
  const int n_off = npart * p.nch * 16;                 // first output channel of this block
  const int st_begin = split * p.stages_per_split, st_end = min(p.stages, st_begin + p.stages_per_split);

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.dy), 0, p.y_bytes, 0x00020000);

  // ---- staging tables: piece e = tid + 512 j of an image -> what it loads, relative to the stage origin (fixed per thread)
  constexpr int NJX = (TN == 6 && WN == 2) ? 3 : 5;   // x pieces per thread: up to 1536 for the 12-chunk form (few input channels per tap group), 2560 else
  constexpr int NJY = EXPAND ? 1 : (12 * 32 * TH + W2_THREADS - 1) / W2_THREADS;
  int xinfo[NJX], yinfo[NJY];  // hy | hx << 8 | channel offset << 16, or -1
#pragma unroll
  for (int j = 0; j < NJX; ++j) {
    const int e = tid + W2_THREADS * j;
    xinfo[j] = -1;
    if (e < p.px) {
      const int per = p.hh * p.hwp * 2;
      const int chunk = e / per, rem = e - chunk * per;
      const int row = rem >> 1, half = rem & 1;
      const int hy = row / p.hwp, pr = row - hy * p.hwp;
      const int hx = pr < 16 ? bitswap23(pr) : pr;
      xinfo[j] = hy | (hx << 8) | ((chunk * 16 + half * 8) << 16);
    }
  }
#pragma unroll
  for (int j = 0; j < NJY; ++j) {
    const int e = tid + W2_THREADS * j;
    yinfo[j] = -1;
    if (!EXPAND && e < p.py) {
      const int chunk = e / (32 * TH), rem = e - chunk * (32 * TH);
      const int pix = bitswap23(rem >> 1), half = rem & 1;
      yinfo[j] = (pix >> 4) | ((pix & 15) << 8) | ((chunk * 16 + half * 8) << 16);
    }
  }
  const int per_img = p.tiles_x * p.tiles_y;
  auto stage_dma = [&](int st, int buf) {
    char* xs = smem + buf * W2_STAGE_BYTES;
    char* ys = xs + p.x_img_bytes;
    const int b = st / per_img, r = st - b * per_img;
    const int tyi = r / p.tiles_x, txi = r - tyi * p.tiles_x;
    const int y0 = tyi * TH, x0 = txi * 16;
    const bool live = st < st_end;
#pragma unroll
    for (int j = 0; j < NJX; ++j) {
      if (W2_THREADS * j + 64 * wave < p.px) {  // wave-uniform
        const int inf = xinfo[j];
        const int yy = y0 + row0 - p.pad_h + (inf & 255), xx = x0 - p.pad_w + ((inf >> 8) & 255);
        const bool v = live && inf >= 0 && (unsigned)yy < (unsigned)p.h && (unsigned)xx < (unsigned)p.w;
        const unsigned off = v ? (unsigned)((((b * p.h + yy) * p.w + xx) * p.cin8 + (inf >> 16)) * 2) : EFM_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void*)(xs + (W2_THREADS * j + 64 * wave) * 16), 16, off, 0, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < NJY; ++j) {
      if (!EXPAND && W2_THREADS * j + 64 * wave < p.py) {
        const int inf = yinfo[j];
        const int yy = y0 + (inf & 255), xx = x0 + ((inf >> 8) & 255);
        const bool v = live && inf >= 0 && yy < p.h && xx < p.w;
        const unsigned off = v ? (unsigned)((((b * p.h + yy) * p.w + xx) * p.cout8 + n_off + (inf >> 16)) * 2) : EFM_OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (__attribute__((address_space(3))) void*)(ys + (W2_THREADS * j + 64 * wave) * 16), 16, off, 0, 0, 0);
      }
    }
  };

  // ---- EXPAND: item = (window of the stage, group of 8 channels of a slice): its dz values and route bytes are loaded ahead of the
  // compute of the previous stage and scattered into the dy image of the next buffer after it.
  constexpr int EW = (TH / 2) * 8;            // pooling windows of a stage
  bf16x8 ex_dz[EXPAND ? 1 : 1];
  uint2 ex_rt = make_uint2(0u, 0u);
  bool ex_ok = false;
  auto expand_load = [&](int st) {
    if constexpr (EXPAND) {
      const int G = p.cs >> 3;
      const int w = tid / G, g = tid - w * G;
      const int b = st / per_img, r = st - b * per_img;
      const int tyi = r / p.tiles_x, txi = r - tyi * p.tiles_x;
      const int gy = tyi * (TH / 2) + (w >> 3), gx = txi * 8 + (w & 7);
      ex_ok = st < st_end && w < EW && gy < p.hp && gx < p.wp;
      if (ex_ok) {
        const long o = ((long)(b * p.hp + gy) * p.wp + gx) * p.cpo + g * 8;
        ex_dz[0] = *reinterpret_cast<const bf16x8*>(p.dz + o);
        ex_rt = *reinterpret_cast<const uint2*>(p.route + o);
      }
    }
  };
  auto expand_store = [&](int buf) {
    if constexpr (EXPAND) {
      const int G = p.cs >> 3;
      const int w = tid / G, g = tid - w * G;
      if (w < EW) {
        char* ys = smem + buf * W2_STAGE_BYTES + p.x_img_bytes;
        const int wy = w >> 3, wx = w & 7;
        const int cpl = p.cs >> 4;                       // chunks per slice
        const int coff = (g >> 1) * YPLANE + (g & 1) * 16;
        int rowoff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rowoff[j] = bitswap23((2 * wy + (j >> 1)) * 16 + 2 * wx + (j & 1)) * 32;
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(ys + sl * cpl * YPLANE + coff + rowoff[j]) = zero;
        if (ex_ok) {
          const s16x8 v = __builtin_bit_cast(s16x8, ex_dz[0]);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const unsigned rt = ((k < 4 ? ex_rt.x : ex_rt.y) >> (8 * (k & 3))) & 0xffu;   // (window pixel) * 4 + slice
            const int j = (int)(rt >> 2), sl = (int)(rt & 1u);
            const int ro = j == 0 ? rowoff[0] : (j == 1 ? rowoff[1] : (j == 2 ? rowoff[2] : rowoff[3]));
            *reinterpret_cast<short*>(ys + sl * cpl * YPLANE + coff + ro + k * 2) = v[k];
          }
        }
      }
    }
  };

  // ---- fragment addressing.  ds_read_b64_tr_b16: lane = 16 q + 4 r + c4 supplies the address of pixel (8 q + 4 hsel + r) of the
  // 32-pixel step, channels 4 c4 .. 4 c4 + 3 of the 16-channel chunk, and receives channel (lane & 15)'s 4 pixels; hsel = 0, 1 give
  // the 8 k values of one MFMA operand.  Pixel 8 q + 4 hsel + r of step s = row 2 s + (q >> 1), column 8 (q & 1) + 4 hsel + r.
  const int fq = lane >> 4, fr = (lane >> 2) & 3, fc4 = lane & 3, fi = lane & 15;
  const int ntw = max(0, min(TN, p.nch - wn * TN));                          // n chunks this wave owns
  const int kbeg = (wk * kch) / WK, ktw = ((wk + 1) * kch) / WK - kbeg;      // k chunks [kbeg, kbeg + ktw): dealt evenly to the WK k waves
  const int wrow = wm * 2 * S;                                               // first stage row of this wave's pixels
  int ylane[2];    // byte offset inside a dy chunk plane, hsel = 0 / 1 (step s adds 32 rows)
#pragma unroll
  for (int hs = 0; hs < 2; ++hs) {
    const int pix = 16 * (fq >> 1) + 8 * (fq & 1) + 4 * hs + fr;
    ylane[hs] = bitswap23(pix) * 32 + fc4 * 8;
  }
  int xlane[TK][2];  // byte offset inside the x image for k chunk kt (its channel chunk plane + tap shift), hsel = 0 / 1
  int kcol[TK];      // column of the chunk in the packed gradient: tap * cin8 + channel chunk * 16
#pragma unroll
  for (int kt = 0; kt < TK; ++kt) {
    const int kc = kbeg + min(kt, max(ktw - 1, 0));
    const int tap_l = kc / p.cch, cc = kc - tap_l * p.cch;
    const int tr = tap_l / p.kw, tc = tap_l - tr * p.kw;
    kcol[kt] = ((row0 + tr) * p.kw + tc) * p.cin8 + cc * 16;
#pragma unroll
    for (int hs = 0; hs < 2; ++hs) {
      const int hx = 8 * (fq & 1) + 4 * hs + fr + tc;
      const int prow = ((fq >> 1) + tr) * p.hwp + (hx < 16 ? bitswap23(hx) : hx);
      xlane[kt][hs] = cc * (p.hh * p.hwp * 32) + prow * 32 + fc4 * 8;
    }
  }
  auto tr_frag = [&](const char* a0, const char* a1) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  f32x4 acc[TN][TK];
#pragma unroll
  for (int a = 0; a < TN; ++a)
#pragma unroll
    for (int b = 0; b < TK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.bias_part != nullptr && tg == 0 && wk == 0;
  float bsum[TN];
#pragma unroll
  for (int a = 0; a < TN; ++a) bsum[a] = 0.f;
  const bf16x2 ones = {(__bf16)1.0f, (__bf16)1.0f};

  auto compute = [&](int buf) {
    const char* xs = smem + buf * W2_STAGE_BYTES;
    const char* ys = xs + p.x_img_bytes;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      bf16x8 a[TN];
#pragma unroll
      for (int nt = 0; nt < TN; ++nt) {
        const int nc = wn * TN + min(nt, max(ntw - 1, 0));   // (a wave with fewer chunks re-reads its last one; the result is dropped)
        const char* base = ys + nc * YPLANE + (wrow + 2 * s) * 512;
        a[nt] = tr_frag(base + ylane[0], base + ylane[1]);
      }
      if (do_bias) {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const bf2 v = {a[nt][2 * k], a[nt][2 * k + 1]};
            bsum[nt] = __builtin_amdgcn_fdot2_f32_bf16(v, ones, bsum[nt], false);
          }
        }
      }
      // k chunks: straight-line code, the fragment of chunk kt + 1 is read while the MFMAs of chunk kt run.  No per-chunk guard: a
      // wave with fewer than TK chunks repeats its last one (the clamp in xlane) into accumulators that are never stored — a
      // branch per chunk kept the compiler from moving any read ahead of the previous chunk's MFMAs: every chunk waited out its own
      // LDS latency (measured: matrix pipe 28 % busy).  TK is picked per layer (ceil(chunks / 4)), so few slots are wasted.
      const char* xb = xs + (wrow + 2 * s) * (p.hwp * 32);
      bf16x8 b = tr_frag(xb + xlane[0][0], xb + xlane[0][1]);
#pragma unroll
      for (int kt = 0; kt < TK; ++kt) {
        bf16x8 bn = b;
        if (kt + 1 < TK) bn = tr_frag(xb + xlane[kt + 1][0], xb + xlane[kt + 1][1]);
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[nt], b, acc[nt][kt], 0, 0, 0);
        b = bn;
      }
    }
  };

  if (st_begin < st_end) {
    stage_dma(st_begin, 0);
    expand_load(st_begin);
    expand_store(0);
  }
  __syncthreads();
  for (int st = st_begin; st < st_end; ++st) {
    const int buf = (st - st_begin) & 1;
    if (st + 1 < st_end) {
      stage_dma(st + 1, buf ^ 1);
      expand_load(st + 1);     // global loads in flight under the MFMAs of this stage
    }
    compute(buf);
    if (st + 1 < st_end) expand_store(buf ^ 1);   // the other buffer: last read before the previous barrier
    __syncthreads();  // drains the LDS-DMA of the next stage (vmcnt(0)) and fences the reads of this one
  }

  // ---- WM > 1: the wave groups hold partial sums of the SAME tile over different rows: add them through LDS in a fixed order (a
  // binary tree: the upper half of the groups writes, the lower half adds), so that the block writes ONE slab.  The staging buffers
  // are free (the loop ended with a barrier); TN*TK tiles x 1 KiB per wave, at most 4 writers: 4 x 30 KiB = exactly the 120 KiB.
  if constexpr (WM > 1) {
    static_assert(WN == 1 && WK == 1 && TN * TK * 1024 * (WM / 2) <= 2 * W2_STAGE_BYTES, "in-block reduction sized for whole-tile waves");
    f32x4* red = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int half = WM / 2; half >= 1; half >>= 1) {
      if (wm >= half && wm < 2 * half) {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int kt = 0; kt < TK; ++kt) red[((wm - half) * TN * TK + nt * TK + kt) * 64 + lane] = acc[nt][kt];
      }
      __syncthreads();
      if (wm < half) {
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
          for (int kt = 0; kt < TK; ++kt) acc[nt][kt] += red[(wm * TN * TK + nt * TK + kt) * 64 + lane];
      }
      __syncthreads();
    }
  }
  // ---- slab: C[row = 4 fq + r -> n][col = fi -> k]
  float* ws = p.ws + (long)split * p.n_pad16 * p.kb_pad;
  if (wm == 0) {
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      if (nt < ntw) {
        const int n = n_off + (wn * TN + nt) * 16 + fq * 4;
#pragma unroll
        for (int kt = 0; kt < TK; ++kt) {
          if (kt < ktw) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ws[(long)(n + r) * p.kb_pad + kcol[kt] + fi] = acc[nt][kt][r];
          }
        }
      }
    }
  }
  const int slab = split * WM + wm;   // bias partials: one per split and pixel group of waves
  if (do_bias) {  // lane (fi = channel, fq = pixel quarter): add the quarters, lanes fq == 0 store
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
      float v = bsum[nt];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (fq == 0 && nt < ntw) p.bias_part[(long)slab * p.n_pad16 + n_off + (wn * TN + nt) * 16 + fi] = v;
    }
  }
}

template <int TN, int TK, int WN, int WK, int S, bool EXPAND>
__global__ void __launch_bounds__(W2_THREADS, 1) convb_wgrad2_k(const Wg2P p) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * W2_STAGE_BYTES];
  wg2_body<TN, TK, WN, WK, S, EXPAND>(p, smem);
}

struct Wg2Plan {
  bool ok;
  int cfg;  // 0: <6,TK,2,4,2>   1: <4,TK,2,4,2>   2: <6,5,1,1,1> (whole tile per wave, waves over the rows)
  int TN, TK, wm, th, nparts, tgroups, nch, cch, rows_g, hh, hwp, x_img_bytes, px, py, tiles_x, tiles_y, stages, stages_per_split, splits, kb_pad;
};

int pad8i(int c) { return (c + 7) & ~7; }

Wg2Plan plan_wg2(const efm_conv_desc* d) {
  Wg2Plan pl;
  pl.ok = false;
  static const bool off = [] { const char* e = getenv("EFM_WGRAD2"); return e && atoi(e) == 0; }();
  if (off) return pl;
  const int cin8 = pad8i(d->cin), cout8 = pad8i(d->cout);
  if ((cin8 & 15) || (cout8 & 15) || cout8 != d->n_pad16) return pl;
  if (d->hout != d->hin || d->wout != d->win || d->kw > 3 || d->kh > 7) return pl;           // 'same' convolutions, stride 1
  if (d->hout == 1 && d->wout == 1) return pl;                                                  // fully connected: a plain GEMM, the im2col form serves it
  const int nchunks = cout8 / 16;
  pl.cch = cin8 / 16;
  const int kall = d->kh * d->kw * pl.cch;
  // candidates in order of preference: the small-gradient form (every wave the whole tile, waves over the rows of a 16-row stage) where
  // the tile and its stage fit; else n parts of 12 chunks (TN = 6) or 8 (TN = 4) with tap groups = whole kernel rows, as many as the
  // k capacity (4 waves x TK) holds
  bool fit = false;
  for (int cand = 0; cand < 2 && !fit; ++cand) {
    pl.wm = 1;
    if (cand == 0) {
      if (!(nchunks <= 6 && kall <= 5)) continue;
      pl.cfg = 2; pl.TN = 6; pl.TK = 5; pl.nch = nchunks; pl.wm = 8;
    } else if (nchunks % 12 == 0) { pl.cfg = 0; pl.TN = 6; pl.TK = 7; pl.nch = 12; }
    else if (nchunks % 8 == 0) { pl.cfg = 1; pl.TN = 4; pl.TK = 9; pl.nch = 8; }
    else if (nchunks <= 12 && nchunks > 8) { pl.cfg = 0; pl.TN = 6; pl.TK = 7; pl.nch = nchunks; }
    else if (nchunks <= 8) { pl.cfg = 1; pl.TN = 4; pl.TK = 9; pl.nch = nchunks; }
    else return pl;
    pl.nparts = (nchunks + pl.nch - 1) / pl.nch;
    const int cap = (pl.cfg == 2 ? 1 : 4) * pl.TK, per_row = d->kw * pl.cch;
    if (per_row > cap) continue;
    pl.rows_g = std::min(d->kh, cap / per_row);
    while (d->kh % pl.rows_g) --pl.rows_g;  // equal groups
    pl.tgroups = d->kh / pl.rows_g;
    if (pl.cfg != 2) pl.TK = std::max(1, (pl.rows_g * per_row + 3) / 4);   // k chunks per wave: the instance without idle slots
    pl.th = pl.cfg == 2 ? 16 : 4;
    pl.hh = pl.th + pl.rows_g - 1;
    pl.hwp = 16 + d->kw - 1;
    pl.px = pl.cch * pl.hh * pl.hwp * 2;
    pl.py = pl.nch * 32 * pl.th;
    pl.x_img_bytes = (pl.px * 16 + 1023) & ~1023;
    const int y_img_bytes = (pl.py * 16 + 1023) & ~1023;
    fit = pl.x_img_bytes + y_img_bytes <= W2_STAGE_BYTES && pl.px <= (pl.cfg == 0 ? 3 : 5) * W2_THREADS && pl.hh <= 255;
  }
  if (!fit) return pl;
  pl.tiles_x = (d->wout + 15) / 16;
  pl.tiles_y = (d->hout + pl.th - 1) / pl.th;
  pl.stages = d->batch * pl.tiles_x * pl.tiles_y;
  static const int target = [] { const char* e = getenv("EFM_WGRAD2_BLOCKS"); return e ? atoi(e) : 256; }();  // one block per CU
  int splits = std::max(1, target / (pl.nparts * pl.tgroups));
  splits = std::min(splits, std::max(1, pl.stages / 8));  // >= 8 stages per block
  pl.stages_per_split = (pl.stages + splits - 1) / splits;
  pl.splits = (pl.stages + pl.stages_per_split - 1) / pl.stages_per_split;
  pl.kb_pad = (d->kh * d->kw * cin8 + 31) & ~31;
  pl.ok = true;
  return pl;
}

// EXPAND form: the gradient of conv -> MFM2 -> 2x2 pooling consumed as dz + route bytes (first convolution: small gradient, cfg 2)
bool wg2_expand_ok(const efm_conv_desc* d, const Wg2Plan& pl, int ways, int pool) {
  const int cs = d->cout / 2;
  return pl.ok && pl.cfg == 2 && ways == 2 && pool && (d->cout % 2) == 0 && (cs % 16) == 0 && (pl.th / 2) * 8 * (cs / 8) <= W2_THREADS;
}

int launch_wg2(const efm_conv_desc* d, const Wg2Plan& pl, const uint16_t* x, const uint16_t* dy, const void* dz, const unsigned char* route,
               float* slabs, float* bias_part, hipStream_t s) {
  Wg2P p;
  p.x = reinterpret_cast<const __bf16*>(x); p.dy = reinterpret_cast<const __bf16*>(dy); p.ws = slabs; p.bias_part = bias_part;
  p.batch = d->batch; p.h = d->hin; p.w = d->win; p.cin8 = pad8i(d->cin); p.cout8 = pad8i(d->cout);
  p.kh = d->kh; p.kw = d->kw; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
  p.n_pad16 = d->n_pad16; p.kb_pad = pl.kb_pad;
  p.nparts = pl.nparts; p.tgroups = pl.tgroups; p.nch = pl.nch; p.cch = pl.cch; p.rows_g = pl.rows_g; p.hh = pl.hh; p.hwp = pl.hwp;
  p.x_img_bytes = pl.x_img_bytes; p.px = pl.px; p.py = pl.py;
  p.tiles_x = pl.tiles_x; p.tiles_y = pl.tiles_y; p.stages = pl.stages; p.stages_per_split = pl.stages_per_split; p.splits = pl.splits;
  p.x_bytes = (unsigned)((size_t)d->batch * d->hin * d->win * p.cin8 * 2);
  p.y_bytes = (unsigned)((size_t)d->batch * d->hout * d->wout * p.cout8 * 2);
  p.dz = reinterpret_cast<const __bf16*>(dz); p.route = route;
  p.hp = d->hout / 2; p.wp = d->wout / 2; p.cs = d->cout / 2; p.cpo = pad8i(d->cout / 2);
  const dim3 grid((unsigned)(pl.splits * pl.nparts * pl.tgroups));
  if (pl.cfg == 2 && dz)
    hipLaunchKernelGGL((convb_wgrad2_k<6, 5, 1, 1, 1, true>), grid, dim3(W2_THREADS), 0, s, p);
  else if (pl.cfg == 2)
    hipLaunchKernelGGL((convb_wgrad2_k<6, 5, 1, 1, 1, false>), grid, dim3(W2_THREADS), 0, s, p);
  else {
#define EFM_W2_CASE(TN_, TK_) \
  else if (pl.TN == TN_ && pl.TK == TK_) hipLaunchKernelGGL((convb_wgrad2_k<TN_, TK_, 2, 4, 2, false>), grid, dim3(W2_THREADS), 0, s, p);
    if (false) {}
    EFM_W2_CASE(6, 1) EFM_W2_CASE(6, 2) EFM_W2_CASE(6, 3) EFM_W2_CASE(6, 4) EFM_W2_CASE(6, 5) EFM_W2_CASE(6, 6) EFM_W2_CASE(6, 7)
    EFM_W2_CASE(4, 1) EFM_W2_CASE(4, 2) EFM_W2_CASE(4, 3) EFM_W2_CASE(4, 4) EFM_W2_CASE(4, 5) EFM_W2_CASE(4, 6) EFM_W2_CASE(4, 7) EFM_W2_CASE(4, 8)
    EFM_W2_CASE(4, 9)
    else {
      efm::set_error("convb_wgrad2: no instance for TN=%d TK=%d", pl.TN, pl.TK);
      return EFM_E_INVALID;
    }
#undef EFM_W2_CASE
  }
  return efm::check_launch("convb_wgrad2");
}

}  // namespace

namespace efm {

bool wgrad2_selected(const efm_conv_desc* d) { return plan_wg2(d).ok; }
int wgrad2_splits(const efm_conv_desc* d) { return plan_wg2(d).splits; }      // slabs: one per block
int wgrad2_bias_chunks(const efm_conv_desc* d) {                                // bias partials: one per block and pixel group of waves
  const Wg2Plan pl = plan_wg2(d);
  return pl.splits * pl.wm;
}
bool wgrad2_expand_supported(const efm_conv_desc* d, int ways, int pool) { return wg2_expand_ok(d, plan_wg2(d), ways, pool); }

// slabs [splits][n_pad16][kb_pad] + bias partials [splits][n_pad16] (bias_part may be null) into the caller's workspace.
// dz != null: the EXPAND form (dy is not read; dz / route = the fused epilogue's output gradient and route bytes).
int wgrad2_slabs(const efm_conv_desc* d, const uint16_t* x, const uint16_t* dy, const void* dz, const unsigned char* route, float* slabs,
                 float* bias_part, hipStream_t s) {
  const Wg2Plan pl = plan_wg2(d);
  if (!pl.ok || (dz && !wg2_expand_ok(d, pl, 2, 1))) {
    efm::set_error("convb_wgrad2: layer not supported by the halo-tile form");
    return EFM_E_INVALID;
  }
  return launch_wg2(d, pl, x, dy, dz, route, slabs, bias_part, s);
}

int wgrad2_info(const efm_conv_desc* d, char* name, size_t len, double* flops) {
  const Wg2Plan pl = plan_wg2(d);
  if (!pl.ok) return EFM_E_INVALID;
  if (name) snprintf(name, len, "convb_wgrad2_k<%d, %d>", pl.TN, pl.TK);
  // executed: every stage = 16*th pixel slots x (nparts*nch*16) x (kh*kw*cin8) MACs
  if (flops) *flops = 2.0 * 16.0 * pl.th * (double)pl.stages * (double)(pl.nparts * pl.nch * 16) * (double)(d->kh * d->kw * pad8i(d->cin));
  return EFM_OK;
}

}  // namespace efm
