// Embedding-head kernels: L2 normalisation, negative gather, triplet-margin loss, cosine pairs,
// batch-all-pairs cosine (Gram) matrix, semi-hard mining, and the flat-buffer optimisers.
// One 64-lane wavefront per embedding row; reductions are xor-shuffle trees (no LDS, no atomics),
// so every result is bitwise reproducible.
//
// Replaces: mx.nd.norm + divide (ref: train_efm.py:241; final_efm.py:240-243), the Python
// negative-pick copy loop (ref: train_efm.py:234-239; pre-trained_efm_v3.py:202-207),
// gluon.loss.TripletLoss (ref: train_efm.py:210,241; pre-trained_efm_v3.py:183,210), cosine_dist
// (ref: train_efm.py:26-34), Trainer.step with sgd / adam (ref: train_efm.py:213-214,245;
// pre-trained_efm_v3.py:185,212; mutli_gpu_v3.py:159).
#include "efm_common.h"

namespace {

using efm::wave_sum;

// ---- L2 normalisation ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) l2norm_row_fwd_k(const float* __restrict__ x, float* __restrict__ y,
                                                        float* __restrict__ norm, int rows, int d, int ldx, int ldy) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * ldx;
  float s = 0.f;
  for (int k = lane; k < d; k += 64) s = fmaf(xr[k], xr[k], s);
  s = wave_sum(s);
  const float nrm = sqrtf(s);
  if (lane == 0) norm[row] = nrm;
  float* yr = y + (long)row * ldy;
  for (int k = lane; k < d; k += 64) yr[k] = xr[k] / nrm;
}

// dx = (dy - y * <y, dy>) / ||x||
__global__ void __launch_bounds__(256) l2norm_row_bwd_k(const float* __restrict__ y, const float* __restrict__ norm,
                                                        const float* __restrict__ dy, float* __restrict__ dx, int rows,
                                                        int d, int ldy, int lddy, int lddx) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* yr = y + (long)row * ldy;
  const float* gr = dy + (long)row * lddy;
  float s = 0.f;
  for (int k = lane; k < d; k += 64) s = fmaf(yr[k], gr[k], s);
  s = wave_sum(s);
  const float inv = 1.f / norm[row];
  float* dr = dx + (long)row * lddx;
  for (int k = lane; k < d; k += 64) dr[k] = (gr[k] - yr[k] * s) * inv;
}

// Whole-matrix reductions for the Frobenius mode: single block of 1024 threads (matrices here
// are <= 16384 x 684), fixed summation order.
__device__ float block_sum_1024(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  float t = (lane < 16) ? sh[lane] : 0.f;
  t = wave_sum(t);
  __syncthreads();
  return t;
}

__global__ void __launch_bounds__(1024) l2norm_frob_fwd_k(const float* __restrict__ x, float* __restrict__ y,
                                                          float* __restrict__ norm, int rows, int d, int ldx, int ldy) {
  __shared__ float sh[16];
  const long total = (long)rows * d;
  float s = 0.f;
  for (long i = threadIdx.x; i < total; i += 1024) {
    const long r = i / d;
    const float v = x[r * ldx + (i - r * d)];
    s = fmaf(v, v, s);
  }
  const float nrm = sqrtf(block_sum_1024(s, sh));
  if (threadIdx.x == 0) norm[0] = nrm;
  for (long i = threadIdx.x; i < total; i += 1024) {
    const long r = i / d;
    const long c = i - r * d;
    y[r * ldy + c] = x[r * ldx + c] / nrm;
  }
}

__global__ void __launch_bounds__(1024) l2norm_frob_bwd_k(const float* __restrict__ y, const float* __restrict__ norm,
                                                          const float* __restrict__ dy, float* __restrict__ dx, int rows,
                                                          int d, int ldy, int lddy, int lddx) {
  __shared__ float sh[16];
  const long total = (long)rows * d;
  float s = 0.f;
  for (long i = threadIdx.x; i < total; i += 1024) {
    const long r = i / d;
    const long c = i - r * d;
    s = fmaf(y[r * ldy + c], dy[r * lddy + c], s);
  }
  s = block_sum_1024(s, sh);
  const float inv = 1.f / norm[0];
  for (long i = threadIdx.x; i < total; i += 1024) {
    const long r = i / d;
    const long c = i - r * d;
    dx[r * lddx + c] = (dy[r * lddy + c] - y[r * ldy + c] * s) * inv;
  }
}

// ---- row gather --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gather_rows_k(const float* __restrict__ x, const int32_t* __restrict__ idx,
                                                     float* __restrict__ y, int rows, int d, int ldx, int ldy) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)idx[row] * ldx;
  float* yr = y + (long)row * ldy;
  for (int k = lane; k < d; k += 64) yr[k] = xr[k];
}

// ---- triplet loss ------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) triplet_fwd_k(const float* __restrict__ a, const float* __restrict__ p,
                                                     const float* __restrict__ n, float* __restrict__ loss, int rows,
                                                     int d, int lda, int ldp, int ldn, float margin) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* ar = a + (long)row * lda;
  const float* pr = p + (long)row * ldp;
  const float* nr = n + (long)row * ldn;
  float s = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float dp = pr[k] - ar[k], dn = nr[k] - ar[k];
    s += dp * dp - dn * dn;
  }
  s = wave_sum(s);
  if (lane == 0) loss[row] = fmaxf(s + margin, 0.f);
}

// d loss_i / d a = 2 (n - p), / d p = 2 (p - a), / d n = -2 (n - a), gated by loss_i > 0.
__global__ void __launch_bounds__(256) triplet_bwd_k(const float* __restrict__ a, const float* __restrict__ p,
                                                     const float* __restrict__ n, const float* __restrict__ loss,
                                                     const float* __restrict__ gloss, float* __restrict__ da,
                                                     float* __restrict__ dp, float* __restrict__ dn, int rows, int d,
                                                     int lda, int ldp, int ldn, int ldg) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float g = (loss[row] > 0.f) ? 2.f * gloss[row] : 0.f;
  const float* ar = a + (long)row * lda;
  const float* pr = p + (long)row * ldp;
  const float* nr = n + (long)row * ldn;
  for (int k = lane; k < d; k += 64) {
    const float av = ar[k], pv = pr[k], nv = nr[k];
    if (da) da[(long)row * ldg + k] = g * (nv - pv);
    if (dp) dp[(long)row * ldg + k] = g * (pv - av);
    if (dn) dn[(long)row * ldg + k] = -g * (nv - av);
  }
}

// ---- triplet loss over index vectors (in-batch mining): anchor i = row anchor[i]... here every row is an anchor ----------
// loss[i] = relu(|e_i - e_pos[i]|^2 - |e_i - e_neg[i]|^2 + margin); rows with neg[i] < 0 (no other identity) give 0.
__global__ void __launch_bounds__(256) triplet_indexed_fwd_k(const float* __restrict__ e, const int32_t* __restrict__ pos,
                                                             const int32_t* __restrict__ neg, float* __restrict__ loss,
                                                             int rows, int d, int lde, float margin) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int ni = neg[row];
  if (ni < 0) {
    if (lane == 0) loss[row] = 0.f;
    return;
  }
  const float* ar = e + (long)row * lde;
  const float* pr = e + (long)pos[row] * lde;
  const float* nr = e + (long)ni * lde;
  float s = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float dp = pr[k] - ar[k], dn = nr[k] - ar[k];
    s += dp * dp - dn * dn;
  }
  s = wave_sum(s);
  if (lane == 0) loss[row] = fmaxf(s + margin, 0.f);
}

// de[i] = 2 g_i (e_neg[i] - e_pos[i])            (row i as anchor)
//       + 2 g_j (e_i - e_j),  j = inv_pos[i]      (row i as the positive of anchor j; pos is a permutation)
// negatives are detached (ref: train_efm.py:238-239).  No atomics: every output row is written by one wave.
__global__ void __launch_bounds__(256) triplet_indexed_bwd_k(const float* __restrict__ e, const int32_t* __restrict__ pos,
                                                             const int32_t* __restrict__ neg, const int32_t* __restrict__ inv_pos,
                                                             const float* __restrict__ loss, const float* __restrict__ gloss,
                                                             float* __restrict__ de, int rows, int d, int lde, int ldg) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int j = inv_pos[row];
  const float gi = (loss[row] > 0.f) ? 2.f * gloss[row] : 0.f;
  const float gj = (j >= 0 && loss[j] > 0.f) ? 2.f * gloss[j] : 0.f;
  const float* er = e + (long)row * lde;
  const float* pr = e + (long)pos[row] * lde;
  const float* nr = e + (long)(neg[row] < 0 ? row : neg[row]) * lde;
  const float* jr = e + (long)(j < 0 ? row : j) * lde;
  for (int k = lane; k < d; k += 64) de[(long)row * ldg + k] = gi * (nr[k] - pr[k]) + gj * (er[k] - jr[k]);
}

// ---- cosine similarities -----------------------------------------------------------------------
__global__ void __launch_bounds__(256) cosine_pairs_k(const float* __restrict__ a, const float* __restrict__ p,
                                                      const float* __restrict__ n, float* __restrict__ s_ap,
                                                      float* __restrict__ s_an, int rows, int d, int lda, int ldp,
                                                      int ldn) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* ar = a + (long)row * lda;
  const float* pr = p + (long)row * ldp;
  const float* nr = n + (long)row * ldn;
  float aa = 0.f, pp = 0.f, nn = 0.f, ap = 0.f, an = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float av = ar[k], pv = pr[k], nv = nr[k];
    aa = fmaf(av, av, aa); pp = fmaf(pv, pv, pp); nn = fmaf(nv, nv, nn);
    ap = fmaf(av, pv, ap); an = fmaf(av, nv, an);
  }
  aa = wave_sum(aa); pp = wave_sum(pp); nn = wave_sum(nn); ap = wave_sum(ap); an = wave_sum(an);
  if (lane == 0) {
    const float na = sqrtf(aa);
    s_ap[row] = ap / (na * sqrtf(pp));
    s_an[row] = an / (na * sqrtf(nn));
  }
}

__global__ void __launch_bounds__(256) pair_distance_k(const float* __restrict__ a, const float* __restrict__ b,
                                                       const float* __restrict__ mean, float* __restrict__ sqdist,
                                                       float* __restrict__ cosine, int rows, int d, int lda, int ldb) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* ar = a + (long)row * lda;
  const float* br = b + (long)row * ldb;
  float dd = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float mu = mean ? mean[k] : 0.f;
    const float av = ar[k] - mu, bv = br[k] - mu;
    const float df = av - bv;
    dd = fmaf(df, df, dd); aa = fmaf(av, av, aa); bb = fmaf(bv, bv, bb); ab = fmaf(av, bv, ab);
  }
  dd = wave_sum(dd); aa = wave_sum(aa); bb = wave_sum(bb); ab = wave_sum(ab);
  if (lane == 0) {
    sqdist[row] = dd;
    cosine[row] = ab / (sqrtf(aa) * sqrtf(bb));
  }
}

// scores[q][i] = <query_q, gallery_i>: the gallery scan of the deployment code (simd_dot over unit-norm 342-d features,
// ref: Feature.hpp:273-293,345-392), one wave per (gallery row, query), queries staged in LDS.
__global__ void __launch_bounds__(256) gallery_scores_k(const float* __restrict__ query, const float* __restrict__ gallery,
                                                        float* __restrict__ scores, int nq, int n, int d, int ldq, int ldg) {
  extern __shared__ __attribute__((aligned(16))) float qs[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int k = threadIdx.x; k < nq * d; k += 256) qs[k] = query[(long)(k / d) * ldq + (k % d)];
  __syncthreads();
  const int row = blockIdx.x * 4 + wv;
  if (row >= n) return;
  const float* gr = gallery + (long)row * ldg;
  for (int q = 0; q < nq; ++q) {
    float s = 0.f;
    for (int k = lane; k < d; k += 64) s = fmaf(qs[q * d + k], gr[k], s);
    s = wave_sum(s);
    if (lane == 0) scores[(long)q * n + row] = s;
  }
}

// g[i][j] = <e_i, e_j> / (|e_i||e_j|): the B x B cosine matrix the miner scans.  Block = 4 waves = a 64 x 64 tile of g on the fp32
// matrix cores (v_mfma_f32_16x16x4_f32: wave w owns rows 16 w .. 16 w + 15, four 16 x 16 tiles); the two 64-row panels of e stream
// through LDS in chunks of 32 columns (row stride 33 floats: conflict-free fragment reads), the squared norms of both panels are
// summed from the same chunks.  (Was: one block per row, a shuffle-reduced dot per pair — 0.17 ms at B = 512, D = 256, 2.5 % of the
// LightCNN-9 step for a 0.13 GFLOP product.)
__global__ void __launch_bounds__(256) gram_cosine_k(const float* __restrict__ e, float* __restrict__ g, int rows, int d, int lde) {
  constexpr int KC = 32, LD = KC + 1;
  __shared__ float Ei[64 * LD], Ej[64 * LD], ni[64], nj[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int fi = lane & 15, fq = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int srow = tid >> 2, spart = tid & 3;   // norm partials: thread = (row, quarter of the chunk)
  float si = 0.f, sj = 0.f;
  for (int k0 = 0; k0 < d; k0 += KC) {
    for (int t = tid; t < 64 * KC; t += 256) {
      const int r = t / KC, k = t - r * KC;
      const bool kv = k0 + k < d;
      Ei[r * LD + k] = (kv && i0 + r < rows) ? e[(long)(i0 + r) * lde + k0 + k] : 0.f;
      Ej[r * LD + k] = (kv && j0 + r < rows) ? e[(long)(j0 + r) * lde + k0 + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float a = Ei[srow * LD + spart * 8 + k], b = Ej[srow * LD + spart * 8 + k];
      si = fmaf(a, a, si);
      sj = fmaf(b, b, sj);
    }
#pragma unroll
    for (int s4 = 0; s4 < KC / 4; ++s4) {
      const float a = Ei[(wave * 16 + fi) * LD + 4 * s4 + fq];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Ej[(t * 16 + fi) * LD + 4 * s4 + fq], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
  si += __shfl_xor(si, 1, 64); si += __shfl_xor(si, 2, 64);
  sj += __shfl_xor(sj, 1, 64); sj += __shfl_xor(sj, 2, 64);
  if (spart == 0) { ni[srow] = sqrtf(si); nj[srow] = sqrtf(sj); }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int j = j0 + t * 16 + fi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int il = wave * 16 + 4 * fq + r, i = i0 + il;
      if (i < rows && j < rows) g[(long)i * rows + j] = acc[t][r] / (ni[il] * nj[t * 16 + fi]);
    }
  }
}

// Semi-hard negative per anchor: one wave per anchor; lanes scan candidates j, keep
// (best semi-hard: smallest d_an above d_ap) and (fallback: largest d_an); lowest index wins ties.
__global__ void __launch_bounds__(256) mine_semihard_k(const float* __restrict__ g, const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ anchor_idx,
                                                       const int32_t* __restrict__ pos_idx, int32_t* __restrict__ neg_idx,
                                                       int n_anchor, int rows) {
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= n_anchor) return;
  const int a = anchor_idx[t], p = pos_idx[t];
  const int la = labels[a];
  const float* ga = g + (long)a * rows;
  const float d_ap = 1.f - ga[p];
  float best_sh = INFINITY, best_fb = -INFINITY;
  int i_sh = 0x7fffffff, i_fb = 0x7fffffff;
  for (int j = lane; j < rows; j += 64) {
    if (labels[j] == la) continue;
    const float dj = 1.f - ga[j];
    if (dj > d_ap && dj < best_sh) { best_sh = dj; i_sh = j; }
    if (dj > best_fb) { best_fb = dj; i_fb = j; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float os = __shfl_xor(best_sh, o, 64), of = __shfl_xor(best_fb, o, 64);
    const int ois = __shfl_xor(i_sh, o, 64), oif = __shfl_xor(i_fb, o, 64);
    if (os < best_sh || (os == best_sh && ois < i_sh)) { best_sh = os; i_sh = ois; }
    if (of > best_fb || (of == best_fb && oif < i_fb)) { best_fb = of; i_fb = oif; }
  }
  if (lane == 0) neg_idx[t] = (i_sh != 0x7fffffff) ? i_sh : ((i_fb != 0x7fffffff) ? i_fb : -1);
}

// ---- optimisers --------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sgd_k(float* __restrict__ w, const float* __restrict__ g, long n4, float lr,
                                             float wd, float rescale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 wv = reinterpret_cast<f32x4*>(w)[i];
  const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
  for (int k = 0; k < 4; ++k) wv[k] = wv[k] - lr * (rescale * gv[k] + wd * wv[k]);
  reinterpret_cast<f32x4*>(w)[i] = wv;
}

__global__ void __launch_bounds__(256) adam_k(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, long n4, float lr_t, float beta1, float beta2,
                                              float eps, float wd, float rescale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 wv = reinterpret_cast<f32x4*>(w)[i];
  const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
  f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float gr = rescale * gv[k] + wd * wv[k];
    mv[k] = beta1 * mv[k] + (1.f - beta1) * gr;
    vv[k] = beta2 * vv[k] + (1.f - beta2) * gr * gr;
    wv[k] = wv[k] - lr_t * mv[k] / (sqrtf(vv[k]) + eps);
  }
  reinterpret_cast<f32x4*>(w)[i] = wv;
  reinterpret_cast<f32x4*>(m)[i] = mv;
  reinterpret_cast<f32x4*>(v)[i] = vv;
}

}  // namespace

extern "C" {

int efm_l2norm_fwd(const float* x, float* y, float* norm_out, int rows, int d, int ldx, int ldy, int mode, void* stream) {
  EFM_REQUIRE(x && y && norm_out && rows > 0 && d > 0 && ldx >= d && ldy >= d, "l2norm_fwd: bad argument");
  if (mode == EFM_L2_ROW)
    hipLaunchKernelGGL(l2norm_row_fwd_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, norm_out, rows, d, ldx, ldy);
  else if (mode == EFM_L2_FROBENIUS)
    hipLaunchKernelGGL(l2norm_frob_fwd_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, y, norm_out, rows, d, ldx, ldy);
  else
    EFM_REQUIRE(false, "l2norm_fwd: bad mode %d", mode);
  return efm::check_launch("l2norm_fwd");
}

int efm_l2norm_bwd(const float* y, const float* norm, const float* dy, float* dx, int rows, int d, int ldy, int lddy,
                   int lddx, int mode, void* stream) {
  EFM_REQUIRE(y && norm && dy && dx && rows > 0 && d > 0, "l2norm_bwd: bad argument");
  if (mode == EFM_L2_ROW)
    hipLaunchKernelGGL(l2norm_row_bwd_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, y, norm, dy, dx, rows, d, ldy, lddy, lddx);
  else if (mode == EFM_L2_FROBENIUS)
    hipLaunchKernelGGL(l2norm_frob_bwd_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, y, norm, dy, dx, rows, d, ldy, lddy, lddx);
  else
    EFM_REQUIRE(false, "l2norm_bwd: bad mode %d", mode);
  return efm::check_launch("l2norm_bwd");
}

int efm_gather_rows(const float* x, const int32_t* idx, float* y, int rows, int d, int ldx, int ldy, void* stream) {
  EFM_REQUIRE(x && idx && y && rows > 0 && d > 0, "gather_rows: bad argument");
  hipLaunchKernelGGL(gather_rows_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, idx, y, rows, d, ldx, ldy);
  return efm::check_launch("gather_rows");
}

int efm_triplet_fwd(const float* a, const float* p, const float* n, float* loss, int rows, int d, int lda, int ldp,
                    int ldn, float margin, void* stream) {
  EFM_REQUIRE(a && p && n && loss && rows > 0 && d > 0, "triplet_fwd: bad argument");
  hipLaunchKernelGGL(triplet_fwd_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, p, n, loss, rows, d, lda, ldp, ldn, margin);
  return efm::check_launch("triplet_fwd");
}

int efm_triplet_bwd(const float* a, const float* p, const float* n, const float* loss, const float* gloss, float* da,
                    float* dp, float* dn, int rows, int d, int lda, int ldp, int ldn, int ldg, void* stream) {
  EFM_REQUIRE(a && p && n && loss && gloss && rows > 0 && d > 0, "triplet_bwd: bad argument");
  hipLaunchKernelGGL(triplet_bwd_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, p, n, loss, gloss, da, dp, dn, rows, d, lda, ldp, ldn, ldg);
  return efm::check_launch("triplet_bwd");
}

int efm_triplet_indexed_fwd(const float* e, const int32_t* pos, const int32_t* neg, float* loss, int rows, int d, int lde,
                            float margin, void* stream) {
  EFM_REQUIRE(e && pos && neg && loss && rows > 0 && d > 0, "triplet_indexed_fwd: bad argument");
  hipLaunchKernelGGL(triplet_indexed_fwd_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, e, pos, neg, loss, rows, d, lde, margin);
  return efm::check_launch("triplet_indexed_fwd");
}

int efm_triplet_indexed_bwd(const float* e, const int32_t* pos, const int32_t* neg, const int32_t* inv_pos, const float* loss,
                            const float* gloss, float* de, int rows, int d, int lde, int ldg, void* stream) {
  EFM_REQUIRE(e && pos && neg && inv_pos && loss && gloss && de && rows > 0 && d > 0, "triplet_indexed_bwd: bad argument");
  hipLaunchKernelGGL(triplet_indexed_bwd_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, e, pos, neg, inv_pos, loss, gloss, de,
                     rows, d, lde, ldg);
  return efm::check_launch("triplet_indexed_bwd");
}

int efm_cosine_pairs(const float* a, const float* p, const float* n, float* s_ap, float* s_an, int rows, int d,
                     int lda, int ldp, int ldn, void* stream) {
  EFM_REQUIRE(a && p && n && s_ap && s_an && rows > 0 && d > 0, "cosine_pairs: bad argument");
  hipLaunchKernelGGL(cosine_pairs_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, p, n, s_ap, s_an, rows, d, lda, ldp, ldn);
  return efm::check_launch("cosine_pairs");
}

int efm_pair_distance(const float* a, const float* b, const float* mean, float* sqdist, float* cosine, int rows, int d,
                      int lda, int ldb, void* stream) {
  EFM_REQUIRE(a && b && sqdist && cosine && rows > 0 && d > 0, "pair_distance: bad argument");
  hipLaunchKernelGGL(pair_distance_k, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, a, b, mean, sqdist, cosine, rows, d, lda, ldb);
  return efm::check_launch("pair_distance");
}

int efm_gallery_scores(const float* query, const float* gallery, float* scores, int nq, int n, int d, int ldq, int ldg, void* stream) {
  EFM_REQUIRE(query && gallery && scores && nq > 0 && n > 0 && d > 0 && (long)nq * d * 4 <= 64 * 1024, "gallery_scores: bad argument");
  hipLaunchKernelGGL(gallery_scores_k, dim3((n + 3) / 4), dim3(256), (size_t)nq * d * sizeof(float), (hipStream_t)stream, query, gallery, scores,
                     nq, n, d, ldq, ldg);
  return efm::check_launch("gallery_scores");
}

int efm_gram_cosine(const float* e, float* g, int rows, int d, int lde, void* stream) {
  EFM_REQUIRE(e && g && rows > 0 && d > 0 && d <= 16384, "gram_cosine: bad argument");
  const unsigned tiles = (unsigned)((rows + 63) / 64);
  hipLaunchKernelGGL(gram_cosine_k, dim3(tiles, tiles), dim3(256), 0, (hipStream_t)stream, e, g, rows, d, lde);
  return efm::check_launch("gram_cosine");
}

int efm_mine_semihard(const float* g, const int32_t* labels, const int32_t* anchor_idx, const int32_t* pos_idx,
                      int32_t* neg_idx, int n_anchor, int rows, void* stream) {
  EFM_REQUIRE(g && labels && anchor_idx && pos_idx && neg_idx && n_anchor > 0 && rows > 0, "mine_semihard: bad argument");
  hipLaunchKernelGGL(mine_semihard_k, dim3((n_anchor + 3) / 4), dim3(256), 0, (hipStream_t)stream, g, labels, anchor_idx, pos_idx, neg_idx, n_anchor, rows);
  return efm::check_launch("mine_semihard");
}

int efm_sgd_update(float* w, const float* g, int64_t n, float lr, float wd, float rescale, void* stream) {
  EFM_REQUIRE(w && g && n > 0 && n % 4 == 0, "sgd_update: n must be a positive multiple of 4");
  hipLaunchKernelGGL(sgd_k, dim3((unsigned)efm::cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, w, g, (long)(n / 4), lr, wd, rescale);
  return efm::check_launch("sgd_update");
}

int efm_adam_update(float* w, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                    float eps, float wd, float rescale, int step, void* stream) {
  EFM_REQUIRE(w && g && m && v && n > 0 && n % 4 == 0 && step >= 1, "adam_update: bad argument");
  const double c1 = 1.0 - pow((double)beta1, step), c2 = 1.0 - pow((double)beta2, step);
  const float lr_t = (float)(lr * sqrt(c2) / c1);
  hipLaunchKernelGGL(adam_k, dim3((unsigned)efm::cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, (long)(n / 4), lr_t, beta1, beta2, eps, wd, rescale);
  return efm::check_launch("adam_update");
}

}  // extern "C"
