// A consumer of libefm_hip.so that knows nothing about Python or torch: plain C++ + the HIP runtime for memory.
// It runs one convolution forward, the fused conv+MFM3+pool forward and the triplet loss through the C ABI of include/efm_hip.h
// and checks them against straightforward CPU loops — the same shape of code a maintainer would write behind Feature.hpp
// (INTEGRATION.md §3).  Build:  hipcc -std=c++17 -I include tests/c_abi/efm_abi_example.cpp -L <pkg dir> -lefm_hip -o efm_abi_example
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "efm_hip.h"

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
  } while (0)
#define EFM(x)                                                                \
  do {                                                                        \
    int rc_ = (x);                                                            \
    if (rc_ != EFM_OK) { std::printf("efm error %d: %s at %s:%d\n", rc_, efm_last_error_string(), __FILE__, __LINE__); return 3; } \
  } while (0)

static unsigned long long sm_state = 0x1234;
static float urand() {  // splitmix64 -> U[-1, 1)
  unsigned long long z = (sm_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(z >> 40) * (2.0f / 16777216.0f) - 1.0f;
}

int main() {
  const int B = 2, H = 10, W = 12, CIN = 6, COUT = 18, K = 3, PAD = 1;
  efm_conv_desc d;
  EFM(efm_conv_desc_init(&d, B, H, W, CIN, COUT, K, K, PAD, PAD));
  // host tensors in the reference's layouts: x NCHW, w OIHW
  std::vector<float> x((size_t)B * CIN * H * W), w((size_t)COUT * CIN * K * K), bias(d.n_pad16, 0.f);
  for (auto& v : x) v = urand();
  for (auto& v : w) v = 0.2f * urand();
  for (int i = 0; i < COUT; ++i) bias[i] = urand();

  float *dx_nchw, *dx, *dw, *dwp, *db, *dy, *dz;
  unsigned char* droute;
  CK(hipMalloc(&dx_nchw, x.size() * 4));
  CK(hipMalloc(&dx, (size_t)B * H * W * d.cin_p * 4));
  CK(hipMalloc(&dw, w.size() * 4));
  CK(hipMalloc(&dwp, efm_conv_weight_elems(&d) * 4));
  CK(hipMalloc(&db, bias.size() * 4));
  CK(hipMalloc(&dy, (size_t)B * d.hout * d.wout * d.cout_p * 4));
  const int co = 2 * COUT / 3, cpo = (co + 3) & ~3, hp = d.hout / 2, wp = d.wout / 2;
  CK(hipMalloc(&dz, (size_t)B * hp * wp * cpo * 4));
  CK(hipMalloc(&droute, (size_t)B * hp * wp * cpo));
  CK(hipMemcpy(dx_nchw, x.data(), x.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  hipStream_t s;
  CK(hipStreamCreate(&s));

  EFM(efm_nchw_to_nhwc(dx_nchw, dx, B, CIN, H, W, s));
  EFM(efm_conv_pack_weights(&d, dw, dwp, s));
  EFM(efm_conv_fwd(&d, dx, dwp, db, nullptr, dy, s));
  EFM(efm_conv_mfm_fwd(&d, dx, dwp, db, dz, droute, 3, EFM_MFM_ORDER_GROUP, 1, s));
  CK(hipStreamSynchronize(s));

  std::vector<float> y((size_t)B * d.hout * d.wout * d.cout_p), z((size_t)B * hp * wp * cpo);
  CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(z.data(), dz, z.size() * 4, hipMemcpyDeviceToHost));

  // CPU reference: cross-correlation + bias (MXNet Convolution), then MFM3 and 2x2 max pooling
  std::vector<double> ref((size_t)B * COUT * H * W);
  double worst = 0, scale = 0;
  for (int b = 0; b < B; ++b)
    for (int o = 0; o < COUT; ++o)
      for (int i = 0; i < H; ++i)
        for (int j = 0; j < W; ++j) {
          double acc = bias[o];
          for (int c = 0; c < CIN; ++c)
            for (int p = 0; p < K; ++p)
              for (int q = 0; q < K; ++q) {
                const int ii = i + p - PAD, jj = j + q - PAD;
                if (ii >= 0 && ii < H && jj >= 0 && jj < W)
                  acc += (double)x[((size_t)(b * CIN + c) * H + ii) * W + jj] * w[((size_t)(o * CIN + c) * K + p) * K + q];
              }
          ref[((size_t)(b * COUT + o) * H + i) * W + j] = acc;
          const double got = y[((size_t)(b * H + i) * W + j) * d.cout_p + o];
          worst = std::fmax(worst, std::fabs(got - acc));
          scale = std::fmax(scale, std::fabs(acc));
        }
  std::printf("conv_fwd: max abs err %.3e of %.3e\n", worst, scale);
  if (worst > 2e-4 * scale) return 1;
  const int cs = COUT / 3;
  double worst2 = 0;
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < hp; ++i)
      for (int j = 0; j < wp; ++j)
        for (int c = 0; c < cs; ++c) {
          double mx = -1e30, mn = -1e30;
          for (int a = 0; a < 2; ++a)
            for (int bb = 0; bb < 2; ++bb) {
              double v[3];
              for (int sl = 0; sl < 3; ++sl) v[sl] = ref[((size_t)(b * COUT + sl * cs + c) * H + 2 * i + a) * W + 2 * j + bb];
              mx = std::fmax(mx, std::fmax(std::fmax(v[0], v[1]), v[2]));
              mn = std::fmax(mn, std::fmin(std::fmin(v[0], v[1]), v[2]));
            }
          const size_t q = ((size_t)(b * hp + i) * wp + j) * cpo;
          worst2 = std::fmax(worst2, std::fmax(std::fabs(z[q + c] - mx), std::fabs(z[q + cs + c] - mn)));
        }
  std::printf("conv_mfm_fwd (MFM3 + pool): max abs err %.3e\n", worst2);
  if (worst2 > 2e-4 * scale) return 1;

  // triplet loss on 3 rows of 8 numbers
  const int rows = 3, dim = 8;
  std::vector<float> a(rows * dim), p(rows * dim), n(rows * dim), loss(rows);
  for (auto& v : a) v = urand();
  for (auto& v : p) v = urand();
  for (auto& v : n) v = urand();
  float *da, *dp, *dn, *dl;
  CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&dp, a.size() * 4)); CK(hipMalloc(&dn, a.size() * 4)); CK(hipMalloc(&dl, rows * 4));
  CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dp, p.data(), a.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dn, n.data(), a.size() * 4, hipMemcpyHostToDevice));
  EFM(efm_triplet_fwd(da, dp, dn, dl, rows, dim, dim, dim, dim, 0.2f, s));
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(loss.data(), dl, rows * 4, hipMemcpyDeviceToHost));
  for (int r = 0; r < rows; ++r) {
    double acc = 0.2;
    for (int k = 0; k < dim; ++k) {
      const double ap = p[r * dim + k] - a[r * dim + k], an = n[r * dim + k] - a[r * dim + k];
      acc += ap * ap - an * an;
    }
    if (std::fabs(loss[r] - std::fmax(acc, 0.0)) > 1e-5) { std::printf("triplet row %d: %g vs %g\n", r, loss[r], acc); return 1; }
  }
  // error convention: a bad argument returns a negative code and a message, never throws or aborts
  if (efm_conv_fwd(&d, nullptr, dwp, db, nullptr, dy, s) >= 0 || efm_last_error_string()[0] == 0) return 1;
  std::printf("efm C ABI example: OK (library version %d)\n", efm_version());
  return 0;
}
