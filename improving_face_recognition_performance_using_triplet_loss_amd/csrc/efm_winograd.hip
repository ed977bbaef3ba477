// Winograd F(2x2, 3x3) convolution for gfx950 (MI355X), fp32, fused in one kernel: forward / data gradient of the 3x3, pad 1,
// stride 1 convolutions (93 % of EFM-29's flops; efm_symbol.py:32,41,54,65,67 are the call sites it serves).
//
//   y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      d = 4x4 input patch of a 2x2 output tile, g = 3x3 filter
//
// 2.25x fewer multiplies than the direct form, and the 16 element-wise products are 16 independent GEMMs
//   M_xi[tile][co] = sum_ci V_xi[tile][ci] * U_xi[co][ci],   xi = 4i + j in 0..15
// which run on v_mfma_f32_16x16x4_f32 exactly like the direct kernel (fp32 in, fp32 accumulate).
//
// Nothing but x, U and y touches HBM:
//   * U = G g G^T is made once per step from the packed fp32 master weights (wino_u_k);
//   * block = 512 threads = 8 waves = 32 tiles x (16*NTB) output channels x all 16 xi.  Per 8-channel K chunk, 256 threads
//     load the 4x4 patches (thread = tile, 4-channel group, patch column; 4 x 16-byte loads), apply B^T . B in registers (the
//     column step pulls the neighbouring columns out of the 4-lane quad by DPP) and write V to LDS; U chunks arrive by LDS-DMA;
//   * wave (i, h) owns xi = 4i..4i+3 for tile rows 16h..16h+15: 4*NTB accumulator tiles.  One ds_read_b64 per fragment
//     feeds the 2 MFMAs of a chunk (k permuted identically on both operands);
//   * epilogue: the j-combination (M A) happens in registers, the i-combination (A^T .) through LDS, then bias / residual
//     and coalesced float4 stores of the 2x2 output pixels.
#include <algorithm>

#include "efm_common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define EFM_OOB 0x80000000u

constexpr int TB = 64;              // tiles per block
constexpr int KC = 8;               // channels per K chunk
constexpr int V_STAGE = 16 * TB * KC;  // floats: V[xi][row][8], row = tile ^ ((xi & 3) << 1) (spreads the 4 planes a quad writes over the banks)

struct WinoP {
  const float* x;
  const float* u;     // [channel block][K chunk][16 xi][NB rows][KC]  (chunk-major: wino_u_k)
  const float* bias;  // may be null
  const float* res;   // may be null: added to y (residual / skip gradient)
  float* y;
  int batch, h, w, cin_p, cout_p;
  int th, tw, tiles;  // tile grid per image, total tiles
  // ceil(2^32 / (th*tw)), ceil(2^32 / tw): tile -> (image, row, column) by one v_mul_hi_u32 against an SGPR each.  A plain `/` made
  // the compiler keep two per-lane reciprocals alive from the prologue to the epilogue — the two VGPRs wino4_k<3> (192 accumulators)
  // spilled.  Exact while tiles * th*tw < 2^32 (checked on the host).
  unsigned magic_per, magic_tw;  // 0 when the divisor is 1 (2^32 does not fit): one_per / one_tw = 1 then adds n back
  int one_per, one_tw;
  int kpad, chunks, nblocks;
  unsigned x_bytes, u_bytes;
  int dbg;
  // fused bias -> MFM (-> 2x2 max pooling) epilogue (ways > 0): y = z (pooled / MFM output, channel stride cpo), route bytes as
  // efm_conv_mfm_fwd writes them.  Block nb owns channels [nb*cn, nb*cn + cn) of EVERY slice: column r of its tile = slice r / cnb,
  // channel nb*cn + r % cnb (wino_u_k permutes the rows of U accordingly).
  unsigned char* route;
  int cout, ways, order, pool, cn, cpo;
};

// ---- final stage of one output column bb (0 / 1) of the 2x2 tiles, shared by both kernel variants.  Rs = [i 0..3][tile 64][RS]
// holds R_i[bb] = sum_j M[i][j] At[bb][j];  y[a][bb] = sum_i At[a][i] R_i[bb]  (At = [1 1 1 0; 0 1 -1 -1]).
template <int NB, int NT>
__device__ __forceinline__ void wino_final(const WinoP& p, const float* Rs, float* state, int bb, int t0, int nb, int tid) {
  constexpr int RS = NB + 4;
  const int per = p.th * p.tw;
  const int n0 = nb * NB;
  if (p.ways == 0) {
    constexpr int NQ = NB / 4;
    for (int it = tid; it < TB * 2 * NQ; it += NT) {
      const int cq = it % NQ, rest = it / NQ;
      const int a = rest & 1, tl = rest >> 1;
      const int tile = t0 + tl, n = n0 + cq * 4;
      if (tile >= p.tiles || n >= p.cout_p) continue;
      const int b = (int)__umulhi((unsigned)tile, p.magic_per) + tile * p.one_per, r = tile - b * per;
      const int ty = (int)__umulhi((unsigned)r, p.magic_tw) + r * p.one_tw, tx = r - ty * p.tw;
      const int oy = 2 * ty + a, ox = 2 * tx + bb;
      if (oy >= p.h || ox >= p.w) continue;
      const f32x4 r1 = *reinterpret_cast<const f32x4*>(Rs + (1 * TB + tl) * RS + cq * 4);
      const f32x4 r2 = *reinterpret_cast<const f32x4*>(Rs + (2 * TB + tl) * RS + cq * 4);
      f32x4 v;
      if (a == 0) v = *reinterpret_cast<const f32x4*>(Rs + (0 * TB + tl) * RS + cq * 4) + r1 + r2;
      else v = r1 - r2 - *reinterpret_cast<const f32x4*>(Rs + (3 * TB + tl) * RS + cq * 4);
      const long off = ((long)(b * p.h + oy) * p.w + ox) * p.cout_p + n;
      if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + n);
      if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + off);
      *reinterpret_cast<f32x4*>(p.y + off) = v;
    }
    return;
  }
  // ---- fused epilogue: thread = (tile, channel j of this block), all slices and both rows a of the column
  const int ways = p.ways, cs = p.cout / ways;
  const int cb = nb * p.cn, cnb = min(p.cn, cs - cb);
  const int co = (ways == 3) ? 2 * cs : cs, cpo = p.cpo;
  const int hp = p.h >> 1, wp = p.w >> 1;
  for (int it = tid; it < TB * cnb; it += NT) {
    const int j = it % cnb, tl = it / cnb;
    const int tile = t0 + tl;
    if (tile >= p.tiles) continue;
    const int b = (int)__umulhi((unsigned)tile, p.magic_per) + tile * p.one_per, r = tile - b * per;
    const int ty = (int)__umulhi((unsigned)r, p.magic_tw) + r * p.one_tw, tx = r - ty * p.tw;
    float ya[2][3];
#pragma unroll
    for (int sl = 0; sl < 3; ++sl) {
      if (sl < ways) {
        const int col = sl * cnb + j;
        const float r0 = Rs[(0 * TB + tl) * RS + col], r1 = Rs[(1 * TB + tl) * RS + col];
        const float r2 = Rs[(2 * TB + tl) * RS + col], r3 = Rs[(3 * TB + tl) * RS + col];
        const float bv = p.bias ? p.bias[sl * cs + cb + j] : 0.f;
        ya[0][sl] = (r0 + r1 + r2) + bv;
        ya[1][sl] = (r1 - r2 - r3) + bv;
      } else {
        ya[0][sl] = ya[1][sl] = 0.f;
      }
    }
    float bmax = 0.f, bmin = 0.f;
    int rmax = 0, rmin = 0;
    if (p.pool && bb == 1) {
      bmax = state[it * 4 + 0]; bmin = state[it * 4 + 1];
      rmax = __builtin_bit_cast(int, state[it * 4 + 2]); rmin = __builtin_bit_cast(int, state[it * 4 + 3]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const float x0 = ya[a][0], x1 = ya[a][1], x2 = ya[a][2];
      // MXNet tie rules (as conv_fwd_body's fused epilogue): maximum / minimum(lhs, rhs) backward sends a tie to lhs
      int imax = (x0 >= x1) ? 0 : 1, imin = (x0 <= x1) ? 0 : 1;
      float vmax = fmaxf(x0, x1), vmin = fminf(x0, x1);
      if (ways == 3) {
        if (p.order == EFM_MFM_ORDER_GROUP) {
          if (!(vmax >= x2)) imax = 2;
          if (!(vmin <= x2)) imin = 2;
        } else {
          if (x2 >= vmax) imax = 2;
          if (x2 <= vmin) imin = 2;
        }
        vmax = fmaxf(vmax, x2);
        vmin = fminf(vmin, x2);
      }
      if (p.pool) {
        // window scan order is (a, bb) = (0,0) (0,1) (1,0) (1,1) and its FIRST maximum wins; the passes arrive bb-major
        const int jw = a * 2 + bb;
        const bool first = (bb == 0 && a == 0);
        if (first || vmax > bmax || (vmax == bmax && jw * 4 < (rmax & ~3))) { bmax = vmax; rmax = jw * 4 + imax; }
        if (first || vmin > bmin || (vmin == bmin && jw * 4 < (rmin & ~3))) { bmin = vmin; rmin = jw * 4 + imin; }
      } else {
        const int oy = 2 * ty + a, ox = 2 * tx + bb;
        if (oy < p.h && ox < p.w) {
          const long m = (long)(b * p.h + oy) * p.w + ox;
          p.y[m * cpo + cb + j] = vmax;
          p.route[m * cpo + cb + j] = (unsigned char)imax;
          if (ways == 3) {
            p.y[m * cpo + cs + cb + j] = vmin;
            p.route[m * cpo + cs + cb + j] = (unsigned char)imin;
          }
        }
      }
    }
    if (p.pool) {
      if (bb == 0) {
        state[it * 4 + 0] = bmax; state[it * 4 + 1] = bmin;
        state[it * 4 + 2] = __builtin_bit_cast(float, rmax); state[it * 4 + 3] = __builtin_bit_cast(float, rmin);
      } else if (ty < hp && tx < wp) {
        const long q = (long)(b * hp + ty) * wp + tx;
        p.y[q * cpo + cb + j] = bmax;
        p.route[q * cpo + cb + j] = (unsigned char)rmax;
        if (ways == 3) {
          p.y[q * cpo + cs + cb + j] = bmin;
          p.route[q * cpo + cs + cb + j] = (unsigned char)rmin;
        }
      }
    }
  }
  // pad channels of z (written once, by channel block 0)
  if (nb == 0 && cpo > co) {
    const int npad = cpo - co;
    for (int it = tid; it < TB * 2 * npad; it += NT) {
      const int pc = it % npad, rest = it / npad;
      const int a = rest & 1, tl = rest >> 1;
      const int tile = t0 + tl;
      if (tile >= p.tiles) continue;
      const int b = (int)__umulhi((unsigned)tile, p.magic_per) + tile * p.one_per, r = tile - b * per;
      const int ty = (int)__umulhi((unsigned)r, p.magic_tw) + r * p.one_tw, tx = r - ty * p.tw;
      if (p.pool) {
        if (bb == 1 && a == 0 && ty < hp && tx < wp) p.y[((long)(b * hp + ty) * wp + tx) * cpo + co + pc] = 0.f;
      } else {
        const int oy = 2 * ty + a, ox = 2 * tx + bb;
        if (oy < p.h && ox < p.w) p.y[((long)(b * p.h + oy) * p.w + ox) * cpo + co + pc] = 0.f;
      }
    }
  }
}

template <int CTRL>
__device__ __forceinline__ float quad(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// Column step of the input transform B^T d B for the forward / data-gradient kernels, ONE instruction per element.
// Lane c of a quad holds the row-combined value t(c) of patch column c and produces output column j = c:
//   j=0: t0 - t2,  j=1: t1 + t2,  j=2: t2 - t1,  j=3: t1 - t3.
// Every one of them is (own value) +- (one neighbour, quad_perm [2,2,1,1]) — except that column 3 comes out NEGATED: own - nb =
// t3 - t1.  The sign is folded into U: wino_u_k negates the planes xi = 4i + 3, and (-V)(-U) = VU bit for bit.  So the step is
// t += r * quad_perm(t) with r = -1, +1, -1, -1, a single v_fmac_f32_dpp that reads and writes the same register (all lanes read
// before any writes) — 16 VALU per chunk where two v_mov_b32_dpp + one v_fma per element made 48.  The s_nop covers the two wait
// states a DPP read needs after a VALU write of the same register (inline asm is invisible to the compiler's hazard pass); the
// four instructions of a block touch four different registers.
__device__ __forceinline__ void col_step4(f32x4& t, float r) {
  float a = t[0], b = t[1], c = t[2], d = t[3];
  asm volatile(
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %0, %0, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %1, %1, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %2, %2, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f32_dpp %3, %3, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
      : "v"(r));
  t = f32x4{a, b, c, d};
}

// Block = 512 threads = 8 waves = 64 tiles x (16*NTB) output channels x all 16 xi.  The kernel is bound by the bytes in flight
// between L2 and LDS (a chunk is consumed in ~2 us, a load takes ~2.6 us under load, and LDS holds two stages), so the tile is
// as large as the 160 KB of LDS and the 256 registers per lane allow (the U chunk, 16*NB*8 floats, is shared by 64 tiles), and the
// pipeline is three deep on the x side: iteration c issues the patch loads of chunk c+2, transforms chunk c+1 (loaded during
// iteration c-1) into the free V stage while the MFMAs of chunk c run.
template <int NTB>
__device__ __forceinline__ void wino_body(const WinoP& p, float* smem) {
  constexpr int NB = 16 * NTB;
  constexpr int U_STAGE = 16 * NB * KC, STAGE = V_STAGE + U_STAGE;
  constexpr int RS = NB + 4;  // row stride of the epilogue exchange buffer (floats): 4*RS % 32 == 16 spreads the 4 row groups over the banks
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wh = wave & 1;
  const int fi = lane & 15, fq = lane >> 4;
  // XCD-aware bijective remap (workgroups go round-robin over the 8 XCDs, each with its own L2): consecutive logical blocks — the
  // channel blocks of one tile group, which read the same x patches and write the same output lines — share one XCD
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int nb = lid % p.nblocks, tb = lid / p.nblocks;
  const int t0 = tb * TB, n0 = nb * NB;
  const int per = p.th * p.tw;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

  // ---- input-transform coordinates: thread = (tile tt, channel group q, patch column c); c = lane & 3 (a DPP quad)
  const int tt = tid >> 3, q = (tid >> 2) & 1, c = tid & 3;
  unsigned xoff[4];
  {
    const int tile = t0 + tt;
    const int b = (int)__umulhi((unsigned)tile, p.magic_per) + tile * p.one_per, r = tile - b * per;
    const int ty = (int)__umulhi((unsigned)r, p.magic_tw) + r * p.one_tw, tx = r - ty * p.tw;
    const int ix = 2 * tx - 1 + c;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int iy = 2 * ty - 1 + rr;
      const bool ok = tile < p.tiles && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
      xoff[rr] = ok ? (unsigned)((((b * p.h + iy) * p.w + ix) * p.cin_p + q * 4) * 4) : EFM_OOB;
    }
  }
  // ---- U staging: the block's slice of chunk ch is 16*NB*KC consecutive floats of U (chunk-major layout, wino_u_k) in the order of
  // the LDS stage; a wave issues NTB LDS-DMA instructions per chunk, each 1 KiB of consecutive bytes
  const unsigned ubase = (unsigned)(((long)nb * p.chunks * 16 * NB * KC + wave * NTB * 256 + lane * 4) * 4);
  const unsigned uchunk = (unsigned)(16 * NB * KC * 4);
  u32x4 xreg[4];
  auto load_x = [&](int ch) {
    const bool cok = ch * KC + q * 4 < p.cin_p;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      xreg[rr] = __builtin_amdgcn_raw_buffer_load_b128(xr, (cok && xoff[rr] != EFM_OOB) ? xoff[rr] + (unsigned)(ch * KC * 4) : EFM_OOB, 0, 0);
  };
  auto dma_u = [&](int ch, int buf) {
    float* Us = smem + buf * STAGE + V_STAGE;
#pragma unroll
    for (int j = 0; j < NTB; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (__attribute__((address_space(3))) void*)(Us + (wave * NTB + j) * 256), 16,
                                               ubase + (unsigned)ch * uchunk + (unsigned)(j * 1024), 0, 0, 0);
  };
  // B^T d B for 4 channels; this lane ends with column j = c of every row i and stores V[4i + c][tt][4q..4q+3]
  auto transform = [&](int buf) {
    float* Vs = smem + buf * STAGE;
    f32x4 d0 = __builtin_bit_cast(f32x4, xreg[0]), d1 = __builtin_bit_cast(f32x4, xreg[1]);
    f32x4 d2 = __builtin_bit_cast(f32x4, xreg[2]), d3 = __builtin_bit_cast(f32x4, xreg[3]);
    f32x4 t[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
    const float sg = (c == 1) ? 1.f : -1.f;
    const int vrow = (tt ^ (c << 1)) * KC + q * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      col_step4(t[i], sg);  // column 3 is stored negated (see col_step4): U's planes 4i+3 carry the other minus sign
      *reinterpret_cast<f32x4*>(Vs + (4 * i + c) * (TB * KC) + vrow) = t[i];
    }
  };

  f32x4 acc[4][2][NTB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int b = 0; b < NTB; ++b) acc[a][m][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const float* Vs = smem + buf * STAGE;
    const float* Us = Vs + V_STAGE;
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
      const int xi = 4 * wi + jx;
      f32x2 a[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        a[mt] = *reinterpret_cast<const f32x2*>(Vs + xi * (TB * KC) + ((wh * 32 + mt * 16 + fi) ^ (jx << 1)) * KC + fq * 2);
#pragma unroll
      for (int nt = 0; nt < NTB; ++nt) {
        const f32x2 b = *reinterpret_cast<const f32x2*>(Us + ((xi * NB + nt * 16 + fi) * KC) + fq * 2);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          acc[jx][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][0], b[0], acc[jx][mt][nt], 0, 0, 0);
          acc[jx][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][1], b[1], acc[jx][mt][nt], 0, 0, 0);
        }
      }
    }
  };

  const bool dx_ = !(p.dbg & 1), du_ = !(p.dbg & 2), dc_ = !(p.dbg & 4);
  if (dx_) load_x(0);
  if (du_) dma_u(0, 0);
  if (dx_) transform(0);
  if (dx_ && p.chunks > 1) load_x(1);
  __syncthreads();
  for (int ch = 0; ch < p.chunks; ++ch) {
    const bool more = ch + 1 < p.chunks;
    if (more && du_) dma_u(ch + 1, (ch + 1) & 1);
    if (more && dx_) transform((ch + 1) & 1);            // x(ch+1) was loaded during the previous iteration
    if (ch + 2 < p.chunks && dx_) load_x(ch + 2);
    if (dc_) compute(ch & 1);
    __syncthreads();
  }

  // ---- epilogue.  In registers: R_i[b] = sum_j M[i][j] * At[b][j]  (At = [1 1 1 0; 0 1 -1 -1]); through LDS, one output column b
  // per pass: y[a][b] = sum_i At[a][i] R_i[b]
  float* Rs = smem;  // [i][tile 64][RS]
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NTB; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float m0 = acc[0][mt][nt][r], m1 = acc[1][mt][nt][r], m2 = acc[2][mt][nt][r], m3 = acc[3][mt][nt][r];
          const int trow = wh * 32 + mt * 16 + 4 * fq + r;
          Rs[(wi * TB + trow) * RS + nt * 16 + fi] = (bb == 0) ? (m0 + m1 + m2) : (m1 - m2 - m3);
        }
      }
    }
    __syncthreads();
    wino_final<NB, 512>(p, Rs, smem + 4 * TB * RS, bb, t0, nb, tid);
    if (bb == 0) __syncthreads();
  }
}

template <int NTB>
__global__ void __launch_bounds__(512, 1) wino_fwd_k(const WinoP p) {
  constexpr int STAGE = V_STAGE + 16 * 16 * NTB * KC;
  constexpr int EPI = 4 * TB * (16 * NTB + 4) + TB * (8 * NTB) * 4;  // R exchange + the pooled epilogue's per-item state
  __shared__ __attribute__((aligned(16))) float smem[(2 * STAGE > EPI) ? 2 * STAGE : EPI];
  wino_body<NTB>(p, smem);
}

// ---- 4-wave variant: two blocks per CU.  Block = 256 threads = 64 tiles x (16*NTB <= 48) channels x 16 xi, K chunk = 4 channels,
// wave i owns xi = 4i..4i+3 for all 64 tiles (16*NTB accumulator tiles); 2 stages of (V 16 KB + U 4*NTB KB) = 56 KB at NTB = 3,
// so two blocks share a CU and cover each other's load latency, barriers and epilogues.
constexpr int KC4 = 4;
constexpr int V4_STAGE = 16 * TB * KC4;  // floats: V[xi][row][4], row = tile ^ ((xi & 3) << 1)

template <int NTB>
__device__ __forceinline__ void wino4_body(const WinoP& p, float* smem) {
  constexpr int NB = 16 * NTB;
  constexpr int U_STAGE = 16 * NB * KC4, STAGE = V4_STAGE + U_STAGE;
  constexpr int RS = NB + 4;
  const int tid = threadIdx.x, lane = tid & 63, wi = tid >> 6;
  const int fi = lane & 15, fq = lane >> 4;
  // XCD-aware bijective remap (workgroups go round-robin over the 8 XCDs, each with its own L2): consecutive logical blocks — the
  // channel blocks of one tile group, which read the same x patches and write the same output lines — share one XCD
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
  const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int nb = lid % p.nblocks, tb = lid / p.nblocks;
  const int t0 = tb * TB, n0 = nb * NB;
  const int per = p.th * p.tw;

  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

  // input transform: thread = (tile tt, patch column c)
  const int tt = tid >> 2, c = tid & 3;
  unsigned xoff[4];
  {
    const int tile = t0 + tt;
    const int b = (int)__umulhi((unsigned)tile, p.magic_per) + tile * p.one_per, r = tile - b * per;
    const int ty = (int)__umulhi((unsigned)r, p.magic_tw) + r * p.one_tw, tx = r - ty * p.tw;
    const int ix = 2 * tx - 1 + c;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int iy = 2 * ty - 1 + rr;
      const bool ok = tile < p.tiles && (unsigned)iy < (unsigned)p.h && (unsigned)ix < (unsigned)p.w;
      xoff[rr] = ok ? (unsigned)((((b * p.h + iy) * p.w + ix) * p.cin_p) * 4) : EFM_OOB;
    }
  }
  // U staging: NTB LDS-DMA instructions per wave and chunk, each 1 KiB of consecutive bytes of the block's chunk slice (chunk-major U)
  const unsigned ubase = (unsigned)(((long)nb * p.chunks * 16 * NB * KC4 + wi * NTB * 256 + lane * 4) * 4);
  const unsigned uchunk = (unsigned)(16 * NB * KC4 * 4);

  u32x4 xreg[4];
  auto load_x = [&](int ch) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      xreg[rr] = __builtin_amdgcn_raw_buffer_load_b128(xr, (xoff[rr] != EFM_OOB) ? xoff[rr] + (unsigned)(ch * KC4 * 4) : EFM_OOB, 0, 0);
  };
  auto dma_u = [&](int ch, int buf) {
    float* Us = smem + buf * STAGE + V4_STAGE;
#pragma unroll
    for (int j = 0; j < NTB; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ur, (__attribute__((address_space(3))) void*)(Us + (wi * NTB + j) * 256), 16,
                                               ubase + (unsigned)ch * uchunk + (unsigned)(j * 1024), 0, 0, 0);
  };
  auto transform = [&](int buf) {
    float* Vs = smem + buf * STAGE;
    f32x4 d0 = __builtin_bit_cast(f32x4, xreg[0]), d1 = __builtin_bit_cast(f32x4, xreg[1]);
    f32x4 d2 = __builtin_bit_cast(f32x4, xreg[2]), d3 = __builtin_bit_cast(f32x4, xreg[3]);
    f32x4 t[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
    const float sg = (c == 1) ? 1.f : -1.f;
    const int vrow = (tt ^ (c << 1)) * KC4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      col_step4(t[i], sg);  // column 3 is stored negated (see col_step4): U's planes 4i+3 carry the other minus sign
      *reinterpret_cast<f32x4*>(Vs + (4 * i + c) * (TB * KC4) + vrow) = t[i];
    }
  };

  f32x4 acc[4][4][NTB];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int b = 0; b < NTB; ++b) acc[a][m][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const float* Vs = smem + buf * STAGE;
    const float* Us = Vs + V4_STAGE;
#pragma unroll
    for (int jx = 0; jx < 4; ++jx) {
      const int xi = 4 * wi + jx;
      float a[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) a[mt] = Vs[xi * (TB * KC4) + ((mt * 16 + fi) ^ (jx << 1)) * KC4 + fq];
#pragma unroll
      for (int nt = 0; nt < NTB; ++nt) {
        const float b = Us[(xi * NB + nt * 16 + fi) * KC4 + fq];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[jx][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b, acc[jx][mt][nt], 0, 0, 0);
      }
    }
  };

  load_x(0);
  dma_u(0, 0);
  transform(0);
  if (p.chunks > 1) load_x(1);
  __syncthreads();
  for (int ch = 0; ch < p.chunks; ++ch) {
    const bool more = ch + 1 < p.chunks;
    if (more) dma_u(ch + 1, (ch + 1) & 1);
    if (more) transform((ch + 1) & 1);
    if (ch + 2 < p.chunks) load_x(ch + 2);
    compute(ch & 1);
    __syncthreads();
  }

  float* Rs = smem;  // [i][tile 64][RS]
#pragma unroll
  for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NTB; ++nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float m0 = acc[0][mt][nt][r], m1 = acc[1][mt][nt][r], m2 = acc[2][mt][nt][r], m3 = acc[3][mt][nt][r];
          Rs[(wi * TB + mt * 16 + 4 * fq + r) * RS + nt * 16 + fi] = (bb == 0) ? (m0 + m1 + m2) : (m1 - m2 - m3);
        }
      }
    }
    __syncthreads();
    wino_final<NB, 256>(p, Rs, smem + 4 * TB * RS, bb, t0, nb, tid);
    if (bb == 0) __syncthreads();
  }
}

template <int NTB>
__global__ void __launch_bounds__(256, 2) wino4_k(const WinoP p) {
  constexpr int STAGE = V4_STAGE + 16 * 16 * NTB * KC4;
  constexpr int EPI = 4 * TB * (16 * NTB + 4) + TB * (8 * NTB) * 4;
  __shared__ __attribute__((aligned(16))) float smem[(2 * STAGE > EPI) ? 2 * STAGE : EPI];
  wino4_body<NTB>(p, smem);
}

// U[channel block][xi = 4i + j][row][k] = sum_{pq} G[i][p] g[p][q] G[j][q];  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].
// forward:       row n = output channel, k = input channel,  g[p][q] = w[n][(p*3 + q)*cin_p + k]
// data gradient: row n = input channel,  k = output channel, g[p][q] = w[k][((2-p)*3 + (2-q))*cin_p + n]   (tap-flipped transpose)
// both read the packed forward weights w[n_pad16][k_pad].  One thread per (n, k).
// fused-epilogue forward (ways > 0): row r of channel block blk is output channel slice*cs + blk*cn + r % cnb, slice = r / cnb
// (every slice of a channel in one block), rows past the last slice are zero.
__device__ __forceinline__ void wino_u_elem(long i, const float* __restrict__ w, float* __restrict__ u, int dgrad, int cout, int cin,
                                            int k_pad_src, int cin_p, int n_rows, int nbr, int kpad, int ways, int cn, int kc) {
  if (i >= (long)n_rows * kpad) return;
  const int nrow = (int)(i / kpad), k = (int)(i - (long)nrow * kpad);
  int n = nrow;
  if (ways > 0) {
    const int blk0 = nrow / nbr, r = nrow - blk0 * nbr;
    const int cs = cout / ways, cb = blk0 * cn, cnb = min(cn, cs - cb);
    const int sl = (cnb > 0) ? r / cnb : ways;
    n = (sl < ways) ? sl * cs + cb + (r - sl * cnb) : cout;  // cout = out of range -> zero row
  }
  const bool ok = dgrad ? (n < cin && k < cout) : (n < cout && k < cin);
  float g[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float v = 0.f;
      if (ok) v = dgrad ? w[(long)k * k_pad_src + ((2 - a) * 3 + (2 - b)) * cin_p + n] : w[(long)n * k_pad_src + (a * 3 + b) * cin_p + k];
      g[a][b] = v;
    }
  float t[4][3];  // G g
#pragma unroll
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
    t[3][b] = g[2][b];
  }
  // chunk-major: U[channel block][K chunk of kc][16 xi][nbr rows][kc] — the 16 planes of one chunk of one block are ONE contiguous
  // run (16*nbr*kc floats), in exactly the order of the kernels' LDS stage, so an LDS-DMA wave instruction copies 1 KiB of
  // consecutive bytes (8 cache lines) instead of 16- / 32-byte pieces of 32-64 different rows (lines) of a [row][kpad] matrix.
  const int blk = nrow / nbr, row = nrow - blk * nbr;
  const int chn = k / kc, kk = k - chn * kc;
  float* dst = u + (((long)blk * (kpad / kc) + chn) * 16 * nbr + row) * kc + kk;
  const long plane = (long)nbr * kc;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    dst[(4 * a + 0) * plane] = t[a][0];
    dst[(4 * a + 1) * plane] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
    dst[(4 * a + 2) * plane] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
    dst[(4 * a + 3) * plane] = -t[a][2];  // negated: the kernels store column 3 of V negated (col_step4)
  }
}

__global__ void __launch_bounds__(256) wino_u_k(const float* __restrict__ w, float* __restrict__ u, int dgrad, int cout, int cin, int k_pad_src,
                                                int cin_p, int n_rows, int nbr, int kpad, int ways, int cn, int kc) {
  wino_u_elem((long)blockIdx.x * 256 + threadIdx.x, w, u, dgrad, cout, cin, k_pad_src, cin_p, n_rows, nbr, kpad, ways, cn, kc);
}

// Every layer's U of a step in ONE launch (the weights do not change inside a step): a network's ~54 per-layer wino_u_k launches were
// 8-microsecond kernels strung between the convolutions, each leaving the chip almost empty.  The job table travels by value
// in the kernel arguments (no device allocation, no copy).
struct UJob {
  const float* w;
  float* u;
  int dgrad, cout, cin, k_pad_src, cin_p, n_rows, nbr, kpad, ways, cn, kc;
  unsigned first_block;  // jobs own consecutive block ranges
};
constexpr int UJOBS_MAX = 56;  // 56 * 64 B = 3.5 KB of kernel arguments
struct UJobs {
  int n;
  UJob j[UJOBS_MAX];
};

__global__ void __launch_bounds__(256) wino_u_multi_k(const UJobs jobs) {
  int k = 0;
  for (int q = 1; q < jobs.n; ++q)  // block-uniform scan over <= 56 entries
    if (blockIdx.x >= jobs.j[q].first_block) k = q;
  const UJob& jb = jobs.j[k];
  wino_u_elem((long)(blockIdx.x - jb.first_block) * 256 + threadIdx.x, jb.w, jb.u, jb.dgrad, jb.cout, jb.cin, jb.k_pad_src, jb.cin_p, jb.n_rows,
              jb.nbr, jb.kpad, jb.ways, jb.cn, jb.kc);
}

// (the weight gradient in Winograd form lives in efm_wino_wgrad.hip)

struct WinoPlan {
  int NTB, nblocks, n_rows, kpad, variant;
  int cn;  // fused epilogue: channels of every slice per block
};

// bits 9:8 of a descriptor's tune field pick the kernel variant: 0 = default (EFM_WINO_VARIANT, else the 4-wave kernel),
// 1 = 8-wave blocks (one per CU), 2 = 4-wave blocks (two per CU).  U's layout depends on it: make_u and the launch read the same field.
int wino_variant(int tune) {
  static const int dflt = [] { const char* e = getenv("EFM_WINO_VARIANT"); return e ? atoi(e) : 4; }();
  const int v = (tune >> 8) & 3;
  return v == 1 ? 8 : (v == 2 ? 4 : dflt);
}

// output-channel tiles per block: as few blocks as possible with at most EFM_WINO_NTB (default 5) tiles each
// (two stages of V 32 KB + U 8*NTB KB must fit the 160 KB of LDS, 32*NTB accumulator registers the 256 per lane)
WinoPlan plan_wino(int cin_p, int cout, int tune, int ways = 0) {
  WinoPlan pl;
  const int tiles = (cout + 15) / 16;
  pl.variant = wino_variant(tune);
  pl.cn = 0;
  if (ways > 0) {  // fused epilogue: a block holds `ways` slices of cn channels; as few blocks as the widest tile allows
    static const int max8 = [] { const char* e = getenv("EFM_WINO_NTB"); return std::min(5, std::max(3, e ? atoi(e) : 5)); }();
    const int maxnb = (pl.variant == 4) ? 48 : 16 * max8;
    const int cs = cout / ways;
    pl.nblocks = (cs + maxnb / ways - 1) / (maxnb / ways);
    pl.cn = (cs + pl.nblocks - 1) / pl.nblocks;
    pl.nblocks = (cs + pl.cn - 1) / pl.cn;
    pl.NTB = std::max(pl.variant == 4 ? 2 : 3, (ways * pl.cn + 15) / 16);
    pl.n_rows = pl.nblocks * pl.NTB * 16;
    pl.kpad = (pl.variant == 4) ? cin_p : (cin_p + KC - 1) / KC * KC;
    return pl;
  }
  if (pl.variant == 4) {  // 4-wave blocks: 2 or 3 channel tiles per block, whichever pads less (3 on a tie)
    const int b3 = (tiles + 2) / 3, b2 = (tiles + 1) / 2;
    pl.NTB = (3 * b3 <= 2 * b2) ? 3 : 2;
    pl.nblocks = (pl.NTB == 3) ? b3 : b2;
    pl.n_rows = pl.nblocks * pl.NTB * 16;
    pl.kpad = cin_p;
    return pl;
  }
  static const int max_ntb = [] { const char* e = getenv("EFM_WINO_NTB"); return std::min(5, std::max(3, e ? atoi(e) : 5)); }();
  pl.nblocks = (tiles + max_ntb - 1) / max_ntb;
  pl.NTB = std::max(3, (tiles + pl.nblocks - 1) / pl.nblocks);
  pl.n_rows = pl.nblocks * pl.NTB * 16;
  pl.kpad = (cin_p + KC - 1) / KC * KC;
  return pl;
}

int run_wino(const float* x, const float* u, const float* bias, const float* res, float* y, int batch, int h, int w, int cin_p, int cout,
             int cout_p, int tune, hipStream_t s, unsigned char* route = nullptr, int ways = 0, int order = 0, int pool = 0) {
  const WinoPlan pl = plan_wino(cin_p, cout, tune, ways);
  WinoP p;
  p.route = route; p.cout = cout; p.ways = ways; p.order = order; p.pool = pool; p.cn = pl.cn;
  p.cpo = ways ? efm_pad4(ways == 3 ? 2 * (cout / 3) : cout / 2) : 0;
  p.x = x; p.u = u; p.bias = bias; p.res = res; p.y = y;
  p.batch = batch; p.h = h; p.w = w; p.cin_p = cin_p; p.cout_p = cout_p;
  p.th = (h + 1) / 2; p.tw = (w + 1) / 2; p.tiles = batch * p.th * p.tw;
  EFM_REQUIRE((unsigned long long)p.tiles * (unsigned)(p.th * p.tw) < 0x100000000ULL, "winograd: batch * tiles-per-image^2 must stay below 2^32");
  p.one_per = (p.th * p.tw == 1); p.one_tw = (p.tw == 1);
  p.magic_per = p.one_per ? 0u : (unsigned)((0x100000000ULL + (unsigned)(p.th * p.tw) - 1) / (unsigned)(p.th * p.tw));
  p.magic_tw = p.one_tw ? 0u : (unsigned)((0x100000000ULL + (unsigned)p.tw - 1) / (unsigned)p.tw);
  p.kpad = pl.kpad; p.chunks = pl.kpad / (pl.variant == 4 ? KC4 : KC); p.nblocks = pl.nblocks;
  p.x_bytes = (unsigned)((size_t)batch * h * w * cin_p * 4);
  p.u_bytes = (unsigned)((size_t)16 * pl.n_rows * pl.kpad * 4);
#ifdef EFM_ABLATE  // measurement builds only (-DEFM_ABLATE): a production library never skips loads or MFMAs, whatever the environment says
  { const char* e = getenv("EFM_WINO_DBG"); p.dbg = e ? atoi(e) : 0; }
#else
  p.dbg = 0;
#endif
  dim3 grid((unsigned)(efm::cdiv(p.tiles, TB) * pl.nblocks));
  if (pl.variant == 4) {
    if (pl.NTB == 3) hipLaunchKernelGGL((wino4_k<3>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((wino4_k<2>), grid, dim3(256), 0, s, p);
    return efm::check_launch("wino4");
  }
  switch (pl.NTB) {
#define EFM_CASE(N)                                                        \
  case N:                                                                  \
    hipLaunchKernelGGL((wino_fwd_k<N>), grid, dim3(512), 0, s, p);         \
    break;
    EFM_CASE(3) EFM_CASE(4) EFM_CASE(5)  // 6 tiles would fill the LDS exactly (160 KB) but spill accumulators
#undef EFM_CASE
    default:
      efm::set_error("wino: unsupported NTB=%d", pl.NTB);
      return EFM_E_INVALID;
  }
  return efm::check_launch("wino_fwd");
}

}  // namespace

extern "C" {

}  // extern "C"

namespace efm {
// Kernel instance and EXECUTED matrix-core flops of a Winograd launch (16 GEMMs of [tiles padded to 64] x [n_rows] x [kpad]).
// pass: 4 = forward, 5 = forward with fused epilogue (`ways`), 6 = data gradient.
int wino_kernel_info(const efm_conv_desc* d, int pass, int ways, char* name, size_t len, double* flops) {
  if (!efm_wino_supported(d)) return EFM_E_INVALID;
  const bool dg = pass == 6;
  const WinoPlan pl = dg ? plan_wino(d->cout_p, d->cin, d->tune_dgrad) : plan_wino(d->cin_p, d->cout, d->tune_fwd, pass == 5 ? ways : 0);
  const long tiles = (long)d->batch * ((d->hin + 1) / 2) * ((d->win + 1) / 2);
  const long tiles_pad = (tiles + TB - 1) / TB * TB;
  if (name) snprintf(name, len, "%s<%d>", pl.variant == 4 ? "wino4_k" : "wino_fwd_k", pl.NTB);
  if (flops) *flops = 2.0 * 16.0 * (double)tiles_pad * (double)pl.n_rows * (double)pl.kpad;
  return EFM_OK;
}
}  // namespace efm

extern "C" {

int efm_wino_supported(const efm_conv_desc* d) {
  return d && d->kh == 3 && d->kw == 3 && d->pad_h == 1 && d->pad_w == 1 && d->hin >= 2 && d->win >= 2;
}

size_t efm_wino_u_elems(const efm_conv_desc* d, int dgrad) {
  const WinoPlan pl = dgrad ? plan_wino(d->cout_p, d->cin, d->tune_dgrad) : plan_wino(d->cin_p, d->cout, d->tune_fwd);
  return (size_t)16 * pl.n_rows * pl.kpad;
}

int efm_wino_make_u(const efm_conv_desc* d, const float* w_packed, float* u, int dgrad, void* stream) {
  EFM_REQUIRE(efm_wino_supported(d) && w_packed && u, "wino_make_u: unsupported descriptor or null argument");
  const WinoPlan pl = dgrad ? plan_wino(d->cout_p, d->cin, d->tune_dgrad) : plan_wino(d->cin_p, d->cout, d->tune_fwd);
  const long total = (long)pl.n_rows * pl.kpad;
  hipLaunchKernelGGL(wino_u_k, dim3((unsigned)efm::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_packed, u, dgrad ? 1 : 0, d->cout,
                     d->cin, d->k_pad, d->cin_p, pl.n_rows, pl.NTB * 16, pl.kpad, 0, 0, pl.variant == 4 ? KC4 : KC);
  return efm::check_launch("wino_make_u");
}

size_t efm_wino_mfm_u_elems(const efm_conv_desc* d, int ways) {
  if (!d || (ways != 2 && ways != 3) || d->cout % ways) return 0;
  const WinoPlan pl = plan_wino(d->cin_p, d->cout, d->tune_fwd, ways);
  return (size_t)16 * pl.n_rows * pl.kpad;
}

int efm_wino_mfm_make_u(const efm_conv_desc* d, const float* w_packed, float* u, int ways, void* stream) {
  EFM_REQUIRE(efm_wino_supported(d) && w_packed && u, "wino_mfm_make_u: unsupported descriptor or null argument");
  EFM_REQUIRE((ways == 2 || ways == 3) && d->cout % ways == 0, "wino_mfm_make_u: cout=%d not divisible by ways=%d", d->cout, ways);
  const WinoPlan pl = plan_wino(d->cin_p, d->cout, d->tune_fwd, ways);
  const long total = (long)pl.n_rows * pl.kpad;
  hipLaunchKernelGGL(wino_u_k, dim3((unsigned)efm::cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w_packed, u, 0, d->cout, d->cin,
                     d->k_pad, d->cin_p, pl.n_rows, pl.NTB * 16, pl.kpad, ways, pl.cn, pl.variant == 4 ? KC4 : KC);
  return efm::check_launch("wino_mfm_make_u");
}

int efm_wino_make_u_batch(int n, const efm_conv_desc* const* descs, const float* const* w_packed, float* const* u, const int* dgrad,
                          const int* ways, void* stream) {
  EFM_REQUIRE(n >= 0 && (n == 0 || (descs && w_packed && u && dgrad && ways)), "wino_make_u_batch: null argument");
  int done = 0;
  while (done < n) {
    UJobs jobs;
    unsigned blocks = 0;
    jobs.n = 0;
    for (; done < n && jobs.n < UJOBS_MAX; ++done) {
      const efm_conv_desc* d = descs[done];
      EFM_REQUIRE(efm_wino_supported(d) && w_packed[done] && u[done], "wino_make_u_batch: job %d: unsupported descriptor or null pointer", done);
      const int wy = ways[done];
      EFM_REQUIRE(wy == 0 || ((wy == 2 || wy == 3) && !dgrad[done] && d->cout % wy == 0), "wino_make_u_batch: job %d: bad ways %d", done, wy);
      const WinoPlan pl = dgrad[done] ? plan_wino(d->cout_p, d->cin, d->tune_dgrad) : plan_wino(d->cin_p, d->cout, d->tune_fwd, wy);
      UJob& jb = jobs.j[jobs.n++];
      jb.w = w_packed[done]; jb.u = u[done];
      jb.dgrad = dgrad[done] ? 1 : 0; jb.cout = d->cout; jb.cin = d->cin; jb.k_pad_src = d->k_pad; jb.cin_p = d->cin_p;
      jb.n_rows = pl.n_rows; jb.nbr = pl.NTB * 16; jb.kpad = pl.kpad; jb.ways = wy; jb.cn = pl.cn; jb.kc = pl.variant == 4 ? KC4 : KC;
      jb.first_block = blocks;
      blocks += (unsigned)efm::cdiv((long)pl.n_rows * pl.kpad, 256);
    }
    if (blocks) hipLaunchKernelGGL(wino_u_multi_k, dim3(blocks), dim3(256), 0, (hipStream_t)stream, jobs);
    const int rc = efm::check_launch("wino_make_u_batch");
    if (rc != EFM_OK) return rc;
  }
  return EFM_OK;
}

int efm_wino_mfm_fwd(const efm_conv_desc* d, const float* x, const float* u, const float* bias, float* z, unsigned char* route, int ways,
                     int order, int pool, void* stream) {
  EFM_REQUIRE(efm_wino_supported(d) && x && u && z && route, "wino_mfm_fwd: unsupported descriptor or null argument");
  EFM_REQUIRE_RANGE(d, 4, "wino_mfm_fwd");
  EFM_REQUIRE((ways == 2 || ways == 3) && d->cout % ways == 0, "wino_mfm_fwd: cout=%d not divisible by ways=%d", d->cout, ways);
  EFM_REQUIRE(order == EFM_MFM_ORDER_GROUP || order == EFM_MFM_ORDER_RES, "wino_mfm_fwd: bad order %d", order);
  EFM_REQUIRE(!pool || (d->hout >= 2 && d->wout >= 2), "wino_mfm_fwd: pooling needs a map of at least 2x2");
  return run_wino(x, u, bias, nullptr, z, d->batch, d->hin, d->win, d->cin_p, d->cout, d->cout_p, d->tune_fwd, (hipStream_t)stream, route,
                  ways, order, pool ? 1 : 0);
}

int efm_wino_fwd(const efm_conv_desc* d, const float* x, const float* u, const float* bias, const float* residual, float* y, void* stream) {
  EFM_REQUIRE(efm_wino_supported(d) && x && u && y, "wino_fwd: unsupported descriptor or null argument");
  EFM_REQUIRE_RANGE(d, 4, "wino_fwd");
  return run_wino(x, u, bias, residual, y, d->batch, d->hin, d->win, d->cin_p, d->cout, d->cout_p, d->tune_fwd, (hipStream_t)stream);
}

int efm_wino_bwd_data(const efm_conv_desc* d, const float* dy, const float* u_dgrad, const float* add, float* dx, void* stream) {
  EFM_REQUIRE(efm_wino_supported(d) && dy && u_dgrad && dx, "wino_bwd_data: unsupported descriptor or null argument");
  EFM_REQUIRE_RANGE(d, 4, "wino_bwd_data");
  return run_wino(dy, u_dgrad, nullptr, add, dx, d->batch, d->hout, d->wout, d->cout_p, d->cin, d->cin_p, d->tune_dgrad, (hipStream_t)stream);
}

}  // extern "C"
