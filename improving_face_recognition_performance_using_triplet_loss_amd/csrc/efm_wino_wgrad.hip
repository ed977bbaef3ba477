// Weight gradient of the 3x3 / pad 1 / stride 1 convolutions in Winograd form, F(3x3 filter gradient from 2x2 tiles of dy and
// their 4x4 input patches), fp32 on v_mfma_f32_16x16x4_f32 (ref: the weight gradients of efm_symbol.py:32,41,54,65,67's convolutions):
//
//   dw = G^T [ sum_tiles (A dy A^T) .* (B^T d B) ] G        16 independent GEMMs  dU_xi[co][ci] = sum_tile P_xi[tile][co] V_xi[tile][ci]
//
// 2.25x fewer multiplies than the direct weight gradient.  What made the first version of this kernel lose to the direct one was
// the transformed operands: they are 4x the raw data, so an LDS stage held 4 tiles and every 48 MFMAs paid two transforms and a
// barrier.  Here the LDS holds RAW pixels and the transform happens on the way from LDS to the MFMA operand registers:
//   * a chunk = a 4x4 group of tiles of one image (<= 16 tiles = 4 MFMA k steps): its 10x10 input pixels x 16*CIT channels and 8x8
//     dy pixels x 16*COT channels arrive by LDS-DMA (16 bytes per lane, whole pixel rows, no registers, no transform pass), two
//     stages, each followed by a few zero pixels that a tile past the chunk's last one reads as dy;
//   * block = 8 waves, one per (row i of the transform, column pair {0,1} or {2,3}): wave (i, par) owns xi = (i, 2par), (i, 2par+1),
//     2 x COT x CIT accumulator tiles.  An operand fragment of xi is a +- combination of 4 (input) / 1..4 (dy) raw pixels of the
//     lane's tile.  Which pixels and which signs is compile-time per wave: the main loop is instantiated 8 times (switch on the
//     wave index; the chunk barriers pair up across the instances), so pixel offsets are immediates, the combinations are plain
//     adds / subtracts and zero coefficients cost nothing;
//   * MFMA tile e, row m  <->  channel N*m + e: the N values a lane needs of one pixel are N consecutive floats at one address;
//     where the lane's tile of a k step starts in the stage comes from a 512-byte LDS table (chunk class x k step x k lane);
//   * per k step and wave: ~20 LDS reads, ~40 VALU, 2*COT*CIT MFMAs; one barrier per chunk, placed before the MFMAs of the chunk's
//     last k step so that the DMA of the chunk after next is issued under them;
//   * the bias gradient rides along (the first input-channel block of a split sums the dy pixels it stages anyway);
//   * epilogue: G^T . G across the waves through LDS, slabs [split][n_pad16][k_pad] + bias partials exactly as the direct kernel
//     writes them: efm_conv_bwd_weight_{slabs,finish} dispatch here on bit 12 of tune_wgrad and share wgrad_reduce_k.
// Built with -fno-slp-vectorize (build.py): packed fp32 VALU is slower than scalar VALU beside MFMAs.  Measurements, SQ counters and
// the ablation builds behind the EFM_WW_* macros below: profiles/round2_wino_wgrad.md, DESIGN.md section 3c.
#include <algorithm>
#include <string.h>

#include "efm_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lds_ptr;
#define EFM_OOB 0x80000000u

constexpr int WG_R = 4;                          // tile rows of a chunk (its tile columns WG_T are a kernel parameter: 4)
constexpr int XR = 2 * WG_R + 2, YR = 2 * WG_R;  // staged input rows (halo of 1) / dy rows

struct WinoWP {
  const float* x;
  const float* dy;
  float* ws;         // slabs [split][n_pad16][k_pad]
  float* bias_part;  // [split][n_pad16] column sums of dy (written by the first input-channel block of every split), or null
  int batch, h, w, cin_p, cout_p;
  int th, tw;              // tiles per image column / row
  int gyn, gxn, nchunks;   // chunk grid per image, chunks in the batch
  int cob, cib, splits, cps;  // channel blocks, splits over the chunks, chunks per split
  int n_pad16, k_pad;
  unsigned x_bytes, y_bytes;
  int dbg;
};

// MFMA tile e, row / column m of it  <->  channel N*m + e of the block's 16N channels: the N values a lane needs of one staged pixel are
// N consecutive floats (one ds_read_b128 for N = 4, b64s for even N, b32s otherwise) at ONE per-lane address, and both sides of the
// epilogue use the same map.
template <int N>
__device__ __forceinline__ void load_px(const float* px, float (&f)[N]) {
#ifdef EFM_WW_NOLOAD  // ablation build (tools/ww_ablate.sh): operands without LDS reads
#pragma unroll
  for (int e = 0; e < N; ++e) {
    float v = 1.0f;
    asm volatile("" : "+v"(v));
    f[e] = v;
  }
  return;
#endif
  if constexpr (N % 4 == 0) {
#pragma unroll
    for (int g = 0; g < N / 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(px + 4 * g);
      f[4 * g] = v[0]; f[4 * g + 1] = v[1]; f[4 * g + 2] = v[2]; f[4 * g + 3] = v[3];
    }
  } else if constexpr (N % 2 == 0) {
#pragma unroll
    for (int g = 0; g < N / 2; ++g) {
      const f32x2 v = *reinterpret_cast<const f32x2*>(px + 2 * g);
      f[2 * g] = v[0]; f[2 * g + 1] = v[1];
    }
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) f[e] = px[e];
  }
}

// Geometry shared by the kernel and its per-wave main loop.
template <int COT, int CIT, int WG_T>
struct WG {
  static constexpr int XW = 2 * WG_T + 2, YW = 2 * WG_T;  // staged pixels per row
  static constexpr int XPIX = XW * XR, YPIX = YW * YR, ZPIX = YW + 2;
  static constexpr int CHX = 16 * CIT, CHY = 16 * COT;
  // a stage: input pixels | dy pixels | ZPIX zero pixels (what a tile past the chunk's last one reads as dy)
  static constexpr int XST = XPIX * CHX, YST = YPIX * CHY, ZST = ZPIX * CHY, STAGE = XST + YST + ZST;
  static constexpr int KS = WG_R * WG_T / 4;  // k steps of a full chunk
  // tile offsets: [chunk class 4][k step KS][fq 4] x (input offset, dy offset) in bytes from the stage
  static constexpr int TBL = 2 * STAGE, TBL_N = 4 * KS * 4 * 2, MAIN = TBL + TBL_N;
  static constexpr int ES = CHX + 4, EPI = 4 * CHY * ES;
  static constexpr int SMEM = MAIN > EPI ? MAIN : EPI;
};

struct WWCtx {
  int cbeg, cend, tid, lane, wave, fi, fq;
  int th, tw, gyn, gxn;
  bool do_bias;
};

// The main loop of wave (WI, PAR): xi = (WI, 2 PAR) -> acc[0], (WI, 2 PAR + 1) -> acc[1].  Which raw pixels make an operand and with
// which signs is compile-time here (8 instances of the loop, one per wave; the barriers pair up across them):
//   B^T rows (input):  d0 - d2, d1 + d2, d2 - d1, d1 - d3          columns: PAR 0: r0 - r2, r1 + r2;  PAR 1: r2 - r1, r1 - r3
//   A rows (dy):       y0, y0 + y1, y0 - y1, (-) y1                columns: PAR 0: s0, s0 + s1;      PAR 1: s0 - s1, (-) s1
// (the two minus signs in brackets are carried into the epilogue's coefficients).
template <int COT, int CIT, int WG_T, int WI, int PAR, class Dma>
__device__ __forceinline__ void ww_main_loop(const WWCtx& c, float* smem, f32x4 (&acc)[2][COT][CIT], f32x4& bsum, Dma&& stage_chunk) {
  using G = WG<COT, CIT, WG_T>;
  constexpr int XA0 = (WI == 0) ? 0 : (WI == 2 ? 2 : 1), XA1 = (WI <= 1) ? 2 : (WI == 2 ? 1 : 3);
  constexpr bool XADD = (WI == 1);
  constexpr int XROW0 = XA0 * G::XW * G::CHX, XROW1 = XA1 * G::XW * G::CHX;
  constexpr int C0 = PAR, C1 = PAR + 1, C2 = PAR + 2;  // the three input columns this wave combines
  constexpr int YA = (WI == 3) ? 1 : 0;                // the single dy row of WI = 0 / 3
  constexpr bool YTWO = (WI == 1 || WI == 2);
  constexpr int GY = G::CHY / 4, PB = 512 / GY;
  const int bpl = c.tid / GY, bg = c.tid - bpl * GY;
  const char* sbase = reinterpret_cast<const char*>(smem);
  const unsigned lx = (unsigned)(CIT * c.fi * 4), ly = (unsigned)(COT * c.fi * 4);
  const unsigned ltab = (unsigned)(G::TBL * 4 + c.fq * 8);
  const int r_last = c.th - WG_R * (c.gyn - 1), t_last = c.tw - WG_T * (c.gxn - 1);

  float va[CIT], vb[CIT], pa[COT], pb[COT];  // operands of the current k step
  // operands of k step s of the chunk in stage `buf` (class cls: bit 1 = last chunk row, bit 0 = last chunk column of the image)
  auto operands = [&](int cls, int s, int buf) {
    const u32x2 off = *reinterpret_cast<const u32x2*>(sbase + ltab + (unsigned)((cls * G::KS + s) * 32));
    const unsigned sb = (unsigned)(buf * G::STAGE * 4);
    const float* xb = reinterpret_cast<const float*>(sbase + (off[0] + lx + sb));
    const float* yb = reinterpret_cast<const float*>(sbase + (off[1] + ly + sb));
    {
      float d0[3][CIT], d1[3][CIT];
      load_px<CIT>(xb + XROW0 + C0 * G::CHX, d0[0]); load_px<CIT>(xb + XROW1 + C0 * G::CHX, d1[0]);
      load_px<CIT>(xb + XROW0 + C1 * G::CHX, d0[1]); load_px<CIT>(xb + XROW1 + C1 * G::CHX, d1[1]);
      load_px<CIT>(xb + XROW0 + C2 * G::CHX, d0[2]); load_px<CIT>(xb + XROW1 + C2 * G::CHX, d1[2]);
#pragma unroll
      for (int n = 0; n < CIT; ++n) {
#ifdef EFM_WW_NOXFORM  // ablation build: operands = raw pixels, no +- combinations
        va[n] = d0[0][n]; vb[n] = d1[1][n];
        continue;
#endif
        const float r0 = XADD ? d0[0][n] + d1[0][n] : d0[0][n] - d1[0][n];
        const float r1 = XADD ? d0[1][n] + d1[1][n] : d0[1][n] - d1[1][n];
        const float r2 = XADD ? d0[2][n] + d1[2][n] : d0[2][n] - d1[2][n];
        if (PAR == 0) { va[n] = r0 - r2; vb[n] = r1 + r2; }   // columns 0 1 2: V0 = r0 - r2, V1 = r1 + r2
        else          { va[n] = r1 - r0; vb[n] = r0 - r2; }   // columns 1 2 3: V2 = r2 - r1, V3 = r1 - r3
      }
    }
    {
      float s0[COT], s1[COT];
      if (YTWO) {
        float y0[COT], y1[COT];
        load_px<COT>(yb, y0); load_px<COT>(yb + G::YW * G::CHY, y1);
#pragma unroll
        for (int m = 0; m < COT; ++m) s0[m] = (WI == 1) ? y0[m] + y1[m] : y0[m] - y1[m];
        load_px<COT>(yb + G::CHY, y0); load_px<COT>(yb + (G::YW + 1) * G::CHY, y1);
#pragma unroll
        for (int m = 0; m < COT; ++m) s1[m] = (WI == 1) ? y0[m] + y1[m] : y0[m] - y1[m];
      } else {
        load_px<COT>(yb + YA * G::YW * G::CHY, s0);
        load_px<COT>(yb + (YA * G::YW + 1) * G::CHY, s1);
      }
#pragma unroll
      for (int m = 0; m < COT; ++m) {
#ifdef EFM_WW_NOXFORM
        pa[m] = s0[m]; pb[m] = s1[m];
        continue;
#endif
        if (PAR == 0) { pa[m] = s0[m]; pb[m] = s0[m] + s1[m]; }
        else          { pa[m] = s0[m] - s1[m]; pb[m] = s1[m]; }
      }
    }
  };
  auto mfmas = [&]() {
#ifndef EFM_WW_NOMFMA
#pragma unroll
    for (int m = 0; m < COT; ++m)
#pragma unroll
      for (int n = 0; n < CIT; ++n) {
        acc[0][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[m], va[n], acc[0][m][n], 0, 0, 0);
        acc[1][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(pb[m], vb[n], acc[1][m][n], 0, 0, 0);
      }
#else  // ablation build: operands formed, no matrix instructions
#pragma unroll
    for (int m = 0; m < COT; ++m) asm volatile("" ::"v"(pa[m]), "v"(pb[m]));
#pragma unroll
    for (int n = 0; n < CIT; ++n) asm volatile("" ::"v"(va[n]), "v"(vb[n]));
#endif
  };
  auto bias_rows = [&](int buf) {  // column sums of the staged dy pixels (pixels outside the image are zeros)
    if (c.do_bias && c.tid < PB * GY) {
      const float* Ys = smem + buf * G::STAGE + G::XST;
#pragma unroll
      for (int px = 0; px < G::YPIX; px += PB)
        if (px + bpl < G::YPIX) bsum += *reinterpret_cast<const f32x4*>(Ys + (px + bpl) * G::CHY + 4 * bg);
    }
  };

  // the k steps of the block's chunks as one stream: the chunk barrier sits BEFORE the MFMAs of a chunk's last step, so that the DMA
  // of the chunk after next is issued under them
  int gy, gx;  // the chunk staged last
#ifdef EFM_WW_STAMPS  // diagnostic build (tools/ww_ablate.sh): cycles per phase by s_memtime, printed by the waves of block 64
  unsigned long long t_mfma = 0, t_opnd = 0, t_bar = 0, t_dma = 0, n_steps = 0, n_chunks = 0;
  auto stamp = [&]() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  };
  const unsigned long long t_begin = stamp();
#define EFM_STAMPED(acc_var, stmt) { const unsigned long long t0_ = stamp(); stmt; acc_var += stamp() - t0_; }
#else
#define EFM_STAMPED(acc_var, stmt) { stmt; }
#endif
  stage_chunk(0, gy, gx);
  int ks = ((gy == c.gyn - 1 ? r_last : WG_R) * (gx == c.gxn - 1 ? t_last : WG_T) + 3) >> 2;
  int cls = (gy == c.gyn - 1 ? 2 : 0) + (gx == c.gxn - 1 ? 1 : 0);
  __syncthreads();
  if (c.cbeg + 1 < c.cend) stage_chunk(1, gy, gx);
  int ch = c.cbeg, s = 0, buf = 0;
  bias_rows(0);
  operands(cls, 0, 0);
  for (;;) {
    // a tight inner loop over the k steps of a chunk (one compare and one backward branch per step); after a chunk switch s = -1,
    // so its first pass runs the MFMAs of the previous chunk's last step and forms the new chunk's first operands
    while (s + 1 < ks) {
      ++s;
      EFM_STAMPED(t_mfma, mfmas());
      EFM_STAMPED(t_opnd, operands(cls, s, buf));
#ifdef EFM_WW_STAMPS
      ++n_steps;
#endif
    }
    if (ch + 1 >= c.cend) break;
#ifndef EFM_WW_NOBARRIER  // (ablation build: races, timing only)
    EFM_STAMPED(t_bar, __syncthreads());  // the next chunk's stage has landed, and every wave holds its last operands of this one
#endif
    ks = ((gy == c.gyn - 1 ? r_last : WG_R) * (gx == c.gxn - 1 ? t_last : WG_T) + 3) >> 2;
    cls = (gy == c.gyn - 1 ? 2 : 0) + (gx == c.gxn - 1 ? 1 : 0);
    ++ch;
    if (ch + 1 < c.cend) EFM_STAMPED(t_dma, stage_chunk(buf, gy, gx));
    buf ^= 1;
    s = -1;
    bias_rows(buf);
#ifdef EFM_WW_STAMPS
    ++n_chunks;
#endif
  }
  mfmas();
#ifdef EFM_WW_STAMPS
  const unsigned long long t_total = stamp() - t_begin;
  if (blockIdx.x == 64 && c.lane == 0)
    printf("wave %d (row %d, columns %d): total %llu cyc; %llu k steps: mfmas %llu (%.0f/step) operands %llu (%.0f/step); %llu chunk switches: "
           "barrier %llu (%.0f) dma %llu (%.0f)\n",
           c.wave, WI, PAR, t_total, n_steps, t_mfma, (double)t_mfma / (double)n_steps, t_opnd, (double)t_opnd / (double)n_steps, n_chunks, t_bar,
           (double)t_bar / (double)n_chunks, t_dma, (double)t_dma / (double)n_chunks);
#endif
#undef EFM_STAMPED
}

template <int COT, int CIT, int WG_T>
__global__ void __launch_bounds__(512, 1) wino_wgrad_k(const WinoWP p) {
  using G = WG<COT, CIT, WG_T>;
  constexpr int XW = G::XW, YW = G::YW, CHX = G::CHX, CHY = G::CHY, STAGE = G::STAGE, XST = G::XST;
  constexpr int GX = CHX / 4, GY = CHY / 4, PXI = 64 / GX, PYI = 64 / GY;
  constexpr int ES = G::ES;
  __shared__ __attribute__((aligned(16))) float smem[G::SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fi = lane & 15, fq = lane >> 4;
  // consecutive logical blocks (the channel blocks of one split: same pixels) share an XCD and its L2
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int base = p.cob * p.cib;
  const int split = lid / base, rblk = lid - split * base;
  const int cb = rblk / p.cib, ib = rblk - cb * p.cib;
  const int co0 = cb * CHY, ci0 = ib * CHX;
  const int cbeg = split * p.cps, cend = min(p.nchunks, cbeg + p.cps);

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, p.y_bytes, 0x00020000);

  // ---- staging: a DMA instruction moves PXI (PYI) pixels of one staged row = 16 bytes per lane, consecutive in LDS.  Wave v owns the
  // staged input rows v and v + 8 and the dy row v of every chunk: the row part of an address is a handful of SALU instructions, per
  // lane there is the column check, one add and the select.
  constexpr int NXB = (XW + PXI - 1) / PXI, NYB = (YW + PYI - 1) / PYI;
  static_assert(YR == 8 && XR <= 16, "row ownership below assumes 8 dy rows and at most 16 input rows");
  const int lpx = lane / GX, lgx = lane - lpx * GX, lpy = lane / GY, lgy = lane - lpy * GY;
  const bool lx_ch = ci0 + 4 * lgx < p.cin_p, ly_ch = co0 + 4 * lgy < p.cout_p;
  const int lx_off = (lpx * p.cin_p + ci0 + 4 * lgx) * 4, ly_off = (lpy * p.cout_p + co0 + 4 * lgy) * 4;
  auto dma_x_row = [&](int rr, int row0, int y0, int x0, float* Xs) {
    const int iy = y0 - 1 + rr;
    const bool rok = (unsigned)iy < (unsigned)p.h;
    const int sb = ((row0 + iy) * p.w + x0 - 1) * p.cin_p * 4;
#pragma unroll
    for (int cbk = 0; cbk < NXB; ++cbk) {
      if (lane < PXI * GX && (XW % PXI == 0 || cbk * PXI + lpx < XW)) {
        const bool ok = rok && lx_ch && (unsigned)(x0 - 1 + cbk * PXI + lpx) < (unsigned)p.w;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr)(Xs + (rr * XW + cbk * PXI) * CHX), 16,
                                                 ok ? (unsigned)(sb + cbk * PXI * p.cin_p * 4 + lx_off) : EFM_OOB, 0, 0, 0);
      }
    }
  };
  auto dma = [&](int b, int gy, int gx, int buf) {
    float* Xs = smem + buf * STAGE;
    float* Ys = Xs + XST;
    const int y0 = 2 * WG_R * gy, x0 = 2 * WG_T * gx, row0 = b * p.h;
    dma_x_row(wave, row0, y0, x0, Xs);
    if (wave + 8 < XR) dma_x_row(wave + 8, row0, y0, x0, Xs);
    const int oy = y0 + wave;
    const bool rok = oy < p.h;
    const int sb = ((row0 + oy) * p.w + x0) * p.cout_p * 4;
#pragma unroll
    for (int cbk = 0; cbk < NYB; ++cbk) {
      if (lane < PYI * GY && (YW % PYI == 0 || cbk * PYI + lpy < YW)) {
        const bool ok = rok && ly_ch && x0 + cbk * PYI + lpy < p.w;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lds_ptr)(Ys + (wave * YW + cbk * PYI) * CHY), 16,
                                                 ok ? (unsigned)(sb + cbk * PYI * p.cout_p * 4 + ly_off) : EFM_OOB, 0, 0, 0);
      }
    }
  };

  f32x4 acc[2][COT][CIT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int m = 0; m < COT; ++m)
#pragma unroll
      for (int n = 0; n < CIT; ++n) acc[a][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  // bias gradient: the first input-channel block of a split sums the dy pixels it stages anyway (thread = 4 channels x every PB-th pixel)
  constexpr int PB = 512 / GY;
  const bool do_bias = p.bias_part != nullptr && ib == 0;
  const int bpl = tid / GY, bg = tid - bpl * GY;
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  // zero pixels of both stages, and the tile table: entry (class, k step s, fq) = where tile q = 4 s + fq of a chunk with
  // R x T tiles starts in the stage (input: its patch's top-left pixel; dy: its top-left pixel, or the zero pixels past the last tile)
  for (int i = tid; i < 2 * G::ZST; i += 512) smem[(i / G::ZST) * STAGE + XST + G::YST + i % G::ZST] = 0.f;
  if (tid < 4 * G::KS * 4) {
    const int cls = tid / (G::KS * 4), s = (tid / 4) % G::KS, k = tid & 3;
    const int R = (cls & 2) ? p.th - WG_R * (p.gyn - 1) : WG_R, T = (cls & 1) ? p.tw - WG_T * (p.gxn - 1) : WG_T;
    const int q = 4 * s + k, r = q / T, t = q - r * T;
    const bool valid = q < R * T;
    u32x2 e;
    e[0] = valid ? (unsigned)((2 * r * XW + 2 * t) * CHX * 4) : 0u;
    e[1] = valid ? (unsigned)((XST + (2 * r * YW + 2 * t) * CHY) * 4) : (unsigned)((XST + G::YST) * 4);
    reinterpret_cast<u32x2*>(smem + G::TBL)[tid] = e;
  }

  if (cbeg < cend) {
    // (nb, ngy, ngx): the next chunk to stage
    int nb = cbeg / (p.gyn * p.gxn), rem = cbeg - nb * (p.gyn * p.gxn);
    int ngy = rem / p.gxn, ngx = rem - ngy * p.gxn;
    auto stage_chunk = [&](int buf, int& gy, int& gx) {
      if (!(p.dbg & 1)) dma(nb, ngy, ngx, buf);
      gy = ngy; gx = ngx;
      if (++ngx == p.gxn) {
        ngx = 0;
        if (++ngy == p.gyn) { ngy = 0; ++nb; }
      }
    };
    WWCtx c;
    c.cbeg = cbeg; c.cend = cend; c.tid = tid; c.lane = lane; c.wave = wave; c.fi = fi; c.fq = fq;
    c.th = p.th; c.tw = p.tw; c.gyn = p.gyn; c.gxn = p.gxn; c.do_bias = do_bias;
    switch (wave) {
#define EFM_WAVE(W) case W: ww_main_loop<COT, CIT, WG_T, (W >> 1), (W & 1)>(c, smem, acc, bsum, stage_chunk); break;
      EFM_WAVE(0) EFM_WAVE(1) EFM_WAVE(2) EFM_WAVE(3) EFM_WAVE(4) EFM_WAVE(5) EFM_WAVE(6)
      default: ww_main_loop<COT, CIT, WG_T, 3, 1>(c, smem, acc, bsum, stage_chunk); break;
#undef EFM_WAVE
    }
  }
  __syncthreads();

  // ---- epilogue: dw[pp][q] = sum_ij G[i][pp] dU[i][j] G[j][q], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].  Over this wave's two j in
  // registers, over the partner wave's two j and then over i through LDS, one filter column q per pass.
  const int wi = wave >> 1, par = wave & 1;
  float* Es = smem;  // [i][co CHY][ES]
  if (do_bias) {
    if (tid < PB * GY) *reinterpret_cast<f32x4*>(Es + bpl * CHY + 4 * bg) = bsum;
    __syncthreads();
    if (tid < CHY && co0 + tid < p.n_pad16) {
      float t = 0.f;
      for (int l = 0; l < PB; ++l) t += Es[l * CHY + tid];
      p.bias_part[(long)split * p.n_pad16 + co0 + tid] = t;
    }
    __syncthreads();
  }
  float* slab = p.ws + (long)split * p.n_pad16 * p.k_pad;
  const float sga = (wi == 3) ? -1.f : 1.f, sgb = par ? -sga : sga;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    // G[j][q] of j = 2 par and 2 par + 1
    const float gq_a = sga * (par ? (q == 1 ? -0.5f : 0.5f) : (q == 0 ? 1.f : 0.f));
    const float gq_b = sgb * (par ? (q == 2 ? 1.f : 0.f) : 0.5f);
    auto exchange = [&](bool add) {
#pragma unroll
      for (int m = 0; m < COT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // accumulator row 4 fq + r of tile m = output channel COT (4 fq + r) + m; its CIT tiles hold input channels CIT fi + n
          float* dst = Es + (wi * CHY + COT * (4 * fq + r) + m) * ES + CIT * fi;
#pragma unroll
          for (int n = 0; n < CIT; ++n) {
            const float v = gq_a * acc[0][m][n][r] + gq_b * acc[1][m][n][r];
            dst[n] = add ? v + dst[n] : v;
          }
        }
    };
    if (par) exchange(false);
    __syncthreads();
    if (!par) exchange(true);  // same lanes, same addresses as the partner wave wrote
    __syncthreads();
    for (int it = tid; it < CHY * GX; it += 512) {
      const int co = it / GX, cq = it - co * GX;
      const int n = co0 + co, ci = ci0 + cq * 4;
      if (n >= p.n_pad16 || ci >= p.cin_p) continue;
      const f32x4 e0 = *reinterpret_cast<const f32x4*>(Es + (0 * CHY + co) * ES + cq * 4);
      const f32x4 e1 = *reinterpret_cast<const f32x4*>(Es + (1 * CHY + co) * ES + cq * 4);
      const f32x4 e2 = *reinterpret_cast<const f32x4*>(Es + (2 * CHY + co) * ES + cq * 4);
      const f32x4 e3 = *reinterpret_cast<const f32x4*>(Es + (3 * CHY + co) * ES + cq * 4);
      float* row = slab + (long)n * p.k_pad + ci;
      *reinterpret_cast<f32x4*>(row + (0 * 3 + q) * p.cin_p) = e0 + 0.5f * (e1 + e2);
      *reinterpret_cast<f32x4*>(row + (1 * 3 + q) * p.cin_p) = 0.5f * (e1 - e2);
      *reinterpret_cast<f32x4*>(row + (2 * 3 + q) * p.cin_p) = 0.5f * (e1 + e2) + e3;
    }
    __syncthreads();
  }
  // the K padding of the packed layout (k in [9*cin_p, k_pad)) is written as zeros by the first input-channel block
  if (ib == 0) {
    const int kz = p.k_pad - 9 * p.cin_p;
    for (int it = tid; it < CHY * kz; it += 512) {
      const int co = it / kz, k = it - co * kz;
      if (co0 + co < p.n_pad16) slab[(long)(co0 + co) * p.k_pad + 9 * p.cin_p + k] = 0.f;
    }
  }
}

// ---- host side
struct Shape { int cot, cit, tt; };
// accumulator tiles per wave and xi: COT*CIT <= 24 (2 xi x 4 registers each, 256 registers per lane with two waves per SIMD);
// tt = tile columns of a chunk (4: chunks of 8 columns measured the same speed at twice the LDS)
constexpr Shape kShapes[] = {{6, 4, 4}, {4, 6, 4}, {7, 3, 4}, {5, 4, 4}, {4, 5, 4}, {6, 3, 4}, {5, 3, 4}, {3, 5, 4}, {4, 4, 4}};
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

struct WinoWPlan {
  int shape;
  int cob, cib, splits, cps, gyn, gxn, nchunks;
  size_t slab_floats, ws_floats;
};

int env_shape() {
  static const int forced = [] {
    const char* e = getenv("EFM_WINO_WGRAD_SHAPE");  // "6x4"
    int a = 0, b = 0;
    if (!e || sscanf(e, "%dx%d", &a, &b) != 2) return -1;
    for (int i = 0; i < kNumShapes; ++i)
      if (kShapes[i].cot == a && kShapes[i].cit == b && kShapes[i].tt == (strstr(e, "w8") ? 8 : 4)) return i;
    return -1;
  }();
  return forced;
}

// tune_wgrad of a Winograd weight gradient: bit 12 set; bits 3:0 = 1 + index into kShapes (0: the shape that pads the layer's
// channel tiles least); bits 9:4 = blocks to aim for / 64 (0: one block per CU)
int pick_shape(const efm_conv_desc* d) {
  const int t = d->tune_wgrad & 15;
  if ((d->tune_wgrad & 0x1000) && t >= 1 && t <= kNumShapes) return t - 1;
  if (env_shape() >= 0) return env_shape();
  const int tco = d->n_pad16 / 16, tci = (d->cin_p + 15) / 16;
  int best = 0;
  long best_cost = -1;
  for (int i = 0; i < kNumShapes; ++i) {
    if (kShapes[i].tt != 4) continue;
    const int cot = kShapes[i].cot, cit = kShapes[i].cit;
    const long padded = (long)((tco + cot - 1) / cot * cot) * ((tci + cit - 1) / cit * cit);
    const long cost = padded * 100 - cot * cit;  // padded MFMA work first, then the larger block (fewer operand bytes per MFMA)
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = i; }
  }
  return best;
}

WinoWPlan plan_wino_wgrad(const efm_conv_desc* d) {
  WinoWPlan pl;
  pl.shape = pick_shape(d);
  const int cot = kShapes[pl.shape].cot, cit = kShapes[pl.shape].cit;
  const int th = (d->hin + 1) / 2, tw = (d->win + 1) / 2;
  const int WG_T = kShapes[pl.shape].tt;
  pl.gyn = (th + WG_R - 1) / WG_R;
  pl.gxn = (tw + WG_T - 1) / WG_T;
  pl.nchunks = d->batch * pl.gyn * pl.gxn;
  pl.cob = (d->n_pad16 + 16 * cot - 1) / (16 * cot);
  pl.cib = (d->cin_p + 16 * cit - 1) / (16 * cit);
  const int base = pl.cob * pl.cib;
  static const int env_target = [] { const char* e = getenv("EFM_WINO_WGRAD_BLOCKS"); return e ? atoi(e) : 256; }();
  const int tb = ((d->tune_wgrad & 0x1000) ? (d->tune_wgrad >> 4) & 63 : 0) * 64;
  const int target = tb ? tb : env_target;
  int splits = std::max(1, target / base);
  splits = std::min(splits, std::max(1, pl.nchunks / 8));  // at least 8 chunks per block
  pl.cps = (pl.nchunks + splits - 1) / splits;
  pl.splits = (pl.nchunks + pl.cps - 1) / pl.cps;
  pl.slab_floats = (size_t)pl.splits * d->n_pad16 * d->k_pad;
  pl.ws_floats = pl.slab_floats + (size_t)pl.splits * d->n_pad16;  // slabs, then the bias partials [split][n_pad16]
  return pl;
}

template <int COT, int CIT, int TT>
void launch(const WinoWP& p, int blocks, hipStream_t s) {
  hipLaunchKernelGGL((wino_wgrad_k<COT, CIT, TT>), dim3((unsigned)blocks), dim3(512), 0, s, p);
}

}  // namespace

namespace efm {

// The Winograd form behind efm_conv_bwd_weight_{workspace_bytes,slabs,finish} (efm_conv.hip dispatches here when tune_wgrad has
// bit 12): same two-launch protocol and workspace contract as the direct kernel.
bool wino_wgrad_selected(const efm_conv_desc* d) {
  static const bool off = [] { const char* e = getenv("EFM_WINO_WGRAD"); return e && atoi(e) == 0; }();  // EFM_WINO_WGRAD=0: direct kernel always
  return !off && (d->tune_wgrad & 0x1000) && efm_wino_supported(d);
}

size_t wino_wgrad_ws_floats(const efm_conv_desc* d) { return plan_wino_wgrad(d).ws_floats; }

int wino_wgrad_slabs(const efm_conv_desc* d, const float* x, const float* dy, int want_bias, void* workspace, size_t workspace_bytes,
                     hipStream_t s) {
  const WinoWPlan pl = plan_wino_wgrad(d);
  if (!workspace || workspace_bytes < pl.ws_floats * sizeof(float)) {
    efm::set_error("wino_bwd_weight: workspace %zu B < required %zu B", workspace_bytes, pl.ws_floats * sizeof(float));
    return EFM_E_WORKSPACE;
  }
  WinoWP p;
  p.x = x; p.dy = dy; p.ws = (float*)workspace;
  p.bias_part = want_bias ? p.ws + pl.slab_floats : nullptr;
  p.batch = d->batch; p.h = d->hin; p.w = d->win; p.cin_p = d->cin_p; p.cout_p = d->cout_p;
  p.th = (d->hin + 1) / 2; p.tw = (d->win + 1) / 2;
  p.gyn = pl.gyn; p.gxn = pl.gxn; p.nchunks = pl.nchunks;
  p.n_pad16 = d->n_pad16; p.k_pad = d->k_pad;
  p.cob = pl.cob; p.cib = pl.cib; p.splits = pl.splits; p.cps = pl.cps;
  p.x_bytes = (unsigned)((size_t)d->batch * d->hin * d->win * d->cin_p * 4);
  p.y_bytes = (unsigned)((size_t)d->batch * d->hout * d->wout * d->cout_p * 4);
#ifdef EFM_ABLATE  // measurement builds only (-DEFM_ABLATE): a production library never skips loads or MFMAs, whatever the environment says
  { const char* e = getenv("EFM_WINO_DBG"); p.dbg = e ? atoi(e) : 0; }
#else
  p.dbg = 0;
#endif
  const int blocks = pl.cob * pl.cib * pl.splits;
  switch (kShapes[pl.shape].cot * 256 + kShapes[pl.shape].cit * 16 + kShapes[pl.shape].tt) {
#define EFM_CASE(A, B, T) case A * 256 + B * 16 + T: launch<A, B, T>(p, blocks, s); break;
    EFM_CASE(6, 4, 4) EFM_CASE(4, 6, 4) EFM_CASE(7, 3, 4) EFM_CASE(5, 4, 4) EFM_CASE(4, 5, 4) EFM_CASE(6, 3, 4) EFM_CASE(5, 3, 4) EFM_CASE(3, 5, 4)
    EFM_CASE(4, 4, 4)
#undef EFM_CASE
    default:
      efm::set_error("wino_bwd_weight: no kernel for shape %dx%d", kShapes[pl.shape].cot, kShapes[pl.shape].cit);
      return EFM_E_INVALID;
  }
  return efm::check_launch("wino_wgrad");
}

int wino_wgrad_finish(const efm_conv_desc* d, float* dw_packed, float* dbias, int accumulate, const void* workspace, size_t workspace_bytes,
                      hipStream_t s) {
  const WinoWPlan pl = plan_wino_wgrad(d);
  if (!workspace || workspace_bytes < pl.ws_floats * sizeof(float)) {
    efm::set_error("wino_bwd_weight: workspace %zu B < required %zu B", workspace_bytes, pl.ws_floats * sizeof(float));
    return EFM_E_WORKSPACE;
  }
  const float* slabs = (const float*)workspace;
  return efm::wgrad_reduce(slabs, dw_packed, (long)d->n_pad16 * d->k_pad / 4, pl.splits, slabs + pl.slab_floats, dbias, d->n_pad16 / 4, pl.splits,
                           accumulate, s);
}

int wino_wgrad_info(const efm_conv_desc* d, char* name, size_t len, double* flops) {
  const WinoWPlan pl = plan_wino_wgrad(d);
  const int cot = kShapes[pl.shape].cot, cit = kShapes[pl.shape].cit;
  const int WG_T = kShapes[pl.shape].tt;
  if (name) snprintf(name, len, "wino_wgrad_k<%d, %d, %d>", cot, cit, WG_T);
  if (flops) {
    // executed: 16 xi x (k steps of 4 tiles, chunk by chunk) x padded channel blocks
    const int th = (d->hin + 1) / 2, tw = (d->win + 1) / 2;
    long ksteps = 0;
    for (int gy = 0; gy < pl.gyn; ++gy)
      for (int gx = 0; gx < pl.gxn; ++gx) ksteps += (std::min(WG_R, th - WG_R * gy) * std::min(WG_T, tw - WG_T * gx) + 3) / 4;
    *flops = 2.0 * 16 * 4 * (double)ksteps * d->batch * (16.0 * cot * pl.cob) * (16.0 * cit * pl.cib);
  }
  return EFM_OK;
}

}  // namespace efm

extern "C" {

size_t efm_wino_wgrad_workspace_bytes(const efm_conv_desc* d) { return plan_wino_wgrad(d).ws_floats * sizeof(float); }

int efm_wino_bwd_weight(const efm_conv_desc* d, const float* x, const float* dy, float* dw_packed, float* dbias, int accumulate,
                        void* workspace, size_t workspace_bytes, void* stream) {
  EFM_REQUIRE(efm_wino_supported(d) && x && dy && dw_packed, "wino_bwd_weight: unsupported descriptor or null argument");
  EFM_REQUIRE_RANGE(d, 4, "wino_bwd_weight");
  int rc = efm::wino_wgrad_slabs(d, x, dy, dbias != nullptr, workspace, workspace_bytes, (hipStream_t)stream);
  if (rc != EFM_OK) return rc;
  return efm::wino_wgrad_finish(d, dw_packed, dbias, accumulate, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
