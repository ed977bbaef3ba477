// Native EFM-29 feature predictor with the call shape of MXNet's C predict API.
//
// The reference's deployment code drives the trained network through MXPredCreatePartialOut / MXPredSetInput /
// MXPredForward / MXPredGetOutputShape / MXPredGetOutput / MXPredFree (ref: feature_extraction/c_version/Feature.hpp:163-205)
// and reads the 342-d 'concat29_output' feature (Feature.hpp:24 fvSize = 342).  This file is the drop-in for that consumer
// side on an MI355X: the network structure is the Symbol EFM-29 (ref: efm_symbol.py:22-101), parameters come from an MXNet
// NDArray-list blob (the bytes of an `.params` file), the forward pass is the same fused HIP kernels the trainer uses.
// Unlike the operator entry points, a predictor OWNS its device buffers (as an MXNet predictor does).
#include <string.h>
#include <map>
#include <string>
#include <vector>

#include <stdlib.h>

#include "efm_common.h"
#include "../../include/c_predict_api.h"

namespace {

struct Blob {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

bool parse_params(const unsigned char* buf, size_t size, std::map<std::string, Blob>& out) {
  size_t off = 0;
  auto need = [&](size_t n) { return off + n <= size; };
  auto rd64 = [&](uint64_t& v) { if (!need(8)) return false; memcpy(&v, buf + off, 8); off += 8; return true; };
  auto rd32 = [&](uint32_t& v) { if (!need(4)) return false; memcpy(&v, buf + off, 4); off += 4; return true; };
  uint64_t magic, reserved, count;
  if (!rd64(magic) || !rd64(reserved) || !rd64(count) || magic != 0x112) return false;
  std::vector<Blob> arrays(count);
  for (uint64_t i = 0; i < count; ++i) {
    uint32_t m, ndim;
    if (!rd32(m)) return false;
    if (m == 0xF993FAC9u || m == 0xF993FACAu) {
      uint32_t stype;
      if (!rd32(stype) || stype != 0 || !rd32(ndim)) return false;
    } else if (m == 0xF993FAC8u) {
      if (!rd32(ndim)) return false;
    } else {
      return false;  // legacy (pre-1.0) records are not accepted here
    }
    Blob& b = arrays[i];
    size_t n = 1;
    for (uint32_t d = 0; d < ndim; ++d) {
      uint64_t v;
      if (!rd64(v)) return false;
      b.shape.push_back((int64_t)v);
      n *= (size_t)v;
    }
    if (ndim == 0) continue;
    uint32_t dev_type, dev_id, flag;
    if (!rd32(dev_type) || !rd32(dev_id) || !rd32(flag) || flag != 0) return false;  // float32 only
    if (!need(n * 4)) return false;
    b.data.resize(n);
    memcpy(b.data.data(), buf + off, n * 4);
    off += n * 4;
  }
  uint64_t ncount;
  if (!rd64(ncount) || ncount != count) return false;
  for (uint64_t i = 0; i < count; ++i) {
    uint64_t len;
    if (!rd64(len) || !need(len)) return false;
    std::string name((const char*)buf + off, (size_t)len);
    off += len;
    if (name.rfind("arg:", 0) == 0 || name.rfind("aux:", 0) == 0) name = name.substr(4);
    out[name] = std::move(arrays[i]);
  }
  return true;
}

struct Op {
  int kind;  // 0 conv plain (+residual), 1 conv fused mfm (pool flag), 2 standalone mfm
  efm_conv_desc d;
  float *w = nullptr, *bias = nullptr, *out = nullptr;
  unsigned char* route = nullptr;
  const float *in = nullptr, *res = nullptr;
  int ways = 3, order = 0, pool = 0, c = 0;
  int64_t rows = 0;
};

struct Predictor {
  int batch = 0, c = 0, h = 0, w = 0, feat = 342;
  float *x_nchw = nullptr, *x_nhwc = nullptr, *feat_dev = nullptr;
  // first convolution on a row-packed input (efm_rowpack_nchw), exactly as the training plan runs it: x_nhwc then holds
  // [b][h][w][pad4(kw*c)] and rowpack_kw / rowpack_pad say how it is filled (0 = plain NHWC conversion)
  int rowpack_kw = 0, rowpack_pad = 0;
  std::vector<Op> ops;
  std::vector<void*> owned;
  uint32_t out_shape[2] = {0, 0};
  hipStream_t stream = nullptr;
  // The forward is ~45 fixed launches on fixed buffers: captured once into a HIP graph and replayed (a single-image forward is
  // launch-bound: the deployment case of Feature.hpp).  graph_state: 0 = not tried, 1 = captured, -1 = capture unavailable.
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  int graph_state = 0;
  ~Predictor() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    for (void* p : owned) (void)hipFree(p);
    if (stream) (void)hipStreamDestroy(stream);
  }
  template <class T>
  T* alloc(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T)) != hipSuccess) return nullptr;
    owned.push_back(p);
    return (T*)p;
  }
};

struct Cur {
  const float* p;
  int c, h, w;
};

// conv (+ bias) with epilogue mode; returns false on a missing parameter / allocation failure
bool add_conv(Predictor& P, const std::map<std::string, Blob>& params, const std::string& name, Cur& cur, int cout, int k, int pad,
              int mode, int order, const float* residual) {
  auto wi = params.find(name + "_weight"), bi = params.find(name + "_bias");
  if (wi == params.end() || bi == params.end()) {
    // say which graph this predictor binds: a Gluon checkpoint (structural keys, shared convolutions) is a different network
    std::string first = params.empty() ? std::string("<none>") : params.begin()->first;
    efm::set_error("pred_create: parameter %s_weight / _bias missing. This predictor binds the Symbol EFM-29 of efm_symbol.py:81-101 "
                   "(parameters conv1, conv{L}{x}_res, conv{L}{x}_res_r, conv{L}_r, conv{L}, fc1 — the checkpoints mutli_gpu_v3.py / "
                   "Feature.hpp use); the blob holds %zu arrays, e.g. '%s' — a Gluon LightCNN_29 checkpoint of train_efm.py "
                   "(conv_net.N.conv_op_K.weight, shared convolutions, Dense(1026)) is a different graph and is not served here",
                   name.c_str(), params.size(), first.c_str());
    return false;
  }
  Op op;
  op.kind = mode ? 1 : 0;
  int kh = k, kw = k;
  if (k == 0) { kh = cur.h; kw = cur.w; }  // fully connected = 'valid' conv over the whole map
  // the FIRST convolution (its input is the network input) runs on the row-packed image like the training plan's (plan.py: rowpack):
  // kh x 1 on kw*c channels, weights re-indexed w'[n][j*c + ci][kh] = w[n][ci][kh][j] — same kernels, same bits as the trainer
  static const bool rowpack_on = [] { const char* e = getenv("EFM_ROWPACK"); return !(e && atoi(e) == 0); }();  // as plan.py
  const bool rowpack = rowpack_on && cur.p == P.x_nhwc && k > 1 && 2 * pad == k - 1 && k * cur.c <= 16 && P.rowpack_kw == 0;
  const size_t expect = (size_t)cout * cur.c * kh * kw;
  if (wi->second.data.size() != expect || (int)bi->second.data.size() != cout) {
    efm::set_error("pred_create: %s has %zu weights, expected %zu", name.c_str(), wi->second.data.size(), expect);
    return false;
  }
  std::vector<float> w_row;
  const float* w_host = wi->second.data.data();
  if (rowpack) {
    w_row.resize(expect);
    for (int n = 0; n < cout; ++n)
      for (int ci = 0; ci < cur.c; ++ci)
        for (int a = 0; a < kh; ++a)
          for (int j = 0; j < kw; ++j)
            w_row[((size_t)n * (kw * cur.c) + (j * cur.c + ci)) * kh + a] = w_host[(((size_t)n * cur.c + ci) * kh + a) * kw + j];
    w_host = w_row.data();
    P.rowpack_kw = kw;
    P.rowpack_pad = pad;
    if (efm_conv_desc_init(&op.d, P.batch, cur.h, cur.w, kw * cur.c, cout, kh, 1, pad, 0) != EFM_OK) return false;
  } else if (efm_conv_desc_init(&op.d, P.batch, cur.h, cur.w, cur.c, cout, kh, kw, pad, pad) != EFM_OK) {
    return false;
  }
  float* w_oihw = P.alloc<float>(expect);
  op.w = P.alloc<float>(efm_conv_weight_elems(&op.d));
  op.bias = P.alloc<float>(op.d.n_pad16);
  if (!w_oihw || !op.w || !op.bias) return false;
  if (hipMemcpy(w_oihw, w_host, expect * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(op.bias, 0, op.d.n_pad16 * 4) != hipSuccess ||
      hipMemcpy(op.bias, bi->second.data.data(), cout * 4, hipMemcpyHostToDevice) != hipSuccess) {
    efm::set_error("pred_create: uploading %s failed", name.c_str());
    return false;
  }
  if (efm_conv_pack_weights(&op.d, w_oihw, op.w, P.stream) != EFM_OK) return false;
  op.in = cur.p;
  op.res = residual;
  op.order = order;
  int oc = cout, oh = op.d.hout, ow = op.d.wout;
  if (mode) {
    op.pool = (mode == 2);
    oc = 2 * cout / 3;
    if (op.pool) { oh /= 2; ow /= 2; }
    op.route = P.alloc<unsigned char>((size_t)P.batch * oh * ow * efm_pad4(oc));
    if (!op.route) return false;
  }
  op.out = P.alloc<float>((size_t)P.batch * oh * ow * efm_pad4(oc));
  if (!op.out) return false;
  P.ops.push_back(op);
  cur = Cur{op.out, oc, oh, ow};
  return true;
}

bool add_mfm(Predictor& P, Cur& cur, int order) {
  Op op;
  op.kind = 2;
  op.in = cur.p;
  op.c = cur.c;
  op.order = order;
  op.rows = (int64_t)P.batch * cur.h * cur.w;
  const int oc = 2 * cur.c / 3;
  op.out = P.alloc<float>((size_t)op.rows * efm_pad4(oc));
  if (!op.out) return false;
  P.ops.push_back(op);
  cur = Cur{op.out, oc, cur.h, cur.w};
  return true;
}

bool build_efm29(Predictor& P, const std::map<std::string, Blob>& params) {
  // the five group() calls of efm_symbol.py:84-92: (num_r, num, kernel, pad, layer, res blocks)
  static const int G[5][5] = {{0, 99, 5, 2, 0}, {99, 198, 3, 1, 1}, {198, 387, 3, 1, 2}, {387, 261, 3, 1, 3}, {261, 261, 3, 1, 4}};
  Cur cur{P.x_nhwc, P.c, P.h, P.w};
  for (int g = 0; g < 5; ++g) {
    const std::string layer = std::to_string(g + 1);
    const int num_r = G[g][0], num = G[g][1], k = G[g][2], pad = G[g][3], tar = G[g][4];
    if (num_r > 0) {
      for (int x = 0; x < tar; ++x) {  // res_block (efm_symbol.py:22-44)
        const std::string ln = x == 0 ? layer : layer + std::to_string(x);
        const Cur data = cur;
        if (!add_mfm(P, cur, EFM_MFM_ORDER_RES)) return false;
        if (!add_conv(P, params, "conv" + ln + "_res", cur, num_r, 3, 1, 1, EFM_MFM_ORDER_RES, nullptr)) return false;
        if (!add_conv(P, params, "conv" + ln + "_res_r", cur, num_r * 2 / 3, 3, 1, 0, 0, data.p)) return false;
      }
      if (!add_conv(P, params, "conv" + layer + "_r", cur, num_r, 1, 0, 1, EFM_MFM_ORDER_RES, nullptr)) return false;
    }
    if (!add_conv(P, params, "conv" + layer, cur, num, k, pad, 2, EFM_MFM_ORDER_GROUP, nullptr)) return false;
  }
  if (!add_conv(P, params, "fc1", cur, 513, 0, 0, 0, 0, nullptr)) return false;
  if (!add_mfm(P, cur, EFM_MFM_ORDER_RES)) return false;  // 'concat29_output': the 342-d feature
  P.feat = cur.c;
  P.feat_dev = const_cast<float*>(cur.p);
  P.out_shape[0] = (uint32_t)P.batch;
  P.out_shape[1] = (uint32_t)P.feat;
  return true;
}

}  // namespace

extern "C" {

int efm_pred_create(const char* symbol_json, const void* param_bytes, int param_size, int dev_id, uint32_t num_input_nodes,
                    const char** input_keys, const uint32_t* input_shape_indptr, const uint32_t* input_shape_data,
                    void** out) {
  (void)symbol_json;
  EFM_REQUIRE(param_bytes && param_size > 0 && out, "pred_create: null argument");
  EFM_REQUIRE(num_input_nodes == 1 && input_keys && std::string(input_keys[0]) == "data", "pred_create: the one input is 'data'");
  EFM_REQUIRE(input_shape_indptr && input_shape_data && input_shape_indptr[1] - input_shape_indptr[0] == 4,
              "pred_create: input shape must be (N, C, H, W)");
  if (hipSetDevice(dev_id) != hipSuccess) {
    efm::set_error("pred_create: hipSetDevice(%d) failed", dev_id);
    return EFM_E_LAUNCH;
  }
  std::map<std::string, Blob> params;
  if (!parse_params((const unsigned char*)param_bytes, (size_t)param_size, params)) {
    efm::set_error("pred_create: not a float32 MXNet NDArray-list blob");
    return EFM_E_INVALID;
  }
  Predictor* P = new Predictor();
  const uint32_t* s = input_shape_data + input_shape_indptr[0];
  P->batch = (int)s[0]; P->c = (int)s[1]; P->h = (int)s[2]; P->w = (int)s[3];
  if (hipStreamCreate(&P->stream) != hipSuccess) {
    delete P;
    efm::set_error("pred_create: hipStreamCreate failed");
    return EFM_E_LAUNCH;
  }
  P->x_nchw = P->alloc<float>((size_t)P->batch * P->c * P->h * P->w);
  P->x_nhwc = P->alloc<float>((size_t)P->batch * P->h * P->w * (efm_pad4(P->c) > 16 ? efm_pad4(P->c) : 16));  // plain NHWC (pad4(c)) or the row-packed image (<= 16 channels)
  if (!P->x_nchw || !P->x_nhwc || !build_efm29(*P, params) || hipStreamSynchronize(P->stream) != hipSuccess) {
    delete P;
    return EFM_E_INVALID;
  }
  *out = P;
  return EFM_OK;
}

int efm_pred_set_input(void* handle, const char* key, const float* data, uint32_t size) {
  Predictor* P = (Predictor*)handle;
  EFM_REQUIRE(P && key && data && std::string(key) == "data", "pred_set_input: bad argument");
  EFM_REQUIRE(size == (uint32_t)((size_t)P->batch * P->c * P->h * P->w), "pred_set_input: size %u does not match the bound shape", size);
  if (hipMemcpyAsync(P->x_nchw, data, (size_t)size * 4, hipMemcpyHostToDevice, P->stream) != hipSuccess) return EFM_E_LAUNCH;
  return EFM_OK;
}

static int pred_enqueue(Predictor* P) {
  int rc = P->rowpack_kw ? efm_rowpack_nchw(P->x_nchw, P->x_nhwc, P->batch, P->c, P->h, P->w, P->rowpack_kw, P->rowpack_pad, 0, P->stream)
                         : efm_nchw_to_nhwc(P->x_nchw, P->x_nhwc, P->batch, P->c, P->h, P->w, P->stream);
  for (size_t i = 0; rc == EFM_OK && i < P->ops.size(); ++i) {
    const Op& op = P->ops[i];
    if (op.kind == 0)
      rc = efm_conv_fwd(&op.d, op.in, op.w, op.bias, op.res, op.out, P->stream);
    else if (op.kind == 1)
      rc = efm_conv_mfm_fwd(&op.d, op.in, op.w, op.bias, op.out, op.route, op.ways, op.order, op.pool, P->stream);
    else
      rc = efm_mfm_fwd(op.in, op.out, op.rows, op.c, 3, P->stream);
  }
  return rc;
}

int efm_pred_forward(void* handle) {
  Predictor* P = (Predictor*)handle;
  EFM_REQUIRE(P, "pred_forward: null handle");
  if (P->graph_state == 0) {
    const char* e = getenv("EFM_PRED_GRAPH");
    P->graph_state = -1;
    if (!e || atoi(e) != 0) {
      if (hipStreamBeginCapture(P->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        const int rc = pred_enqueue(P);
        hipGraph_t g = nullptr;
        const hipError_t ec = hipStreamEndCapture(P->stream, &g);
        if (rc == EFM_OK && ec == hipSuccess && g && hipGraphInstantiate(&P->graph_exec, g, nullptr, nullptr, 0) == hipSuccess) {
          P->graph = g;
          P->graph_state = 1;
        } else {
          if (g) (void)hipGraphDestroy(g);
          (void)hipGetLastError();  // capture unavailable: fall back to plain launches
        }
      } else {
        (void)hipGetLastError();
      }
    }
  }
  if (P->graph_state == 1) {
    if (hipGraphLaunch(P->graph_exec, P->stream) == hipSuccess) return EFM_OK;
    efm::set_error("pred_forward: hipGraphLaunch failed");
    return EFM_E_LAUNCH;
  }
  return pred_enqueue(P);
}

int efm_pred_get_output_shape(void* handle, uint32_t index, uint32_t** shape_data, uint32_t* shape_ndim) {
  Predictor* P = (Predictor*)handle;
  EFM_REQUIRE(P && index == 0 && shape_data && shape_ndim, "pred_get_output_shape: bad argument (one output: the feature)");
  *shape_data = P->out_shape;
  *shape_ndim = 2;
  return EFM_OK;
}

int efm_pred_get_output(void* handle, uint32_t index, float* data, uint32_t size) {
  Predictor* P = (Predictor*)handle;
  EFM_REQUIRE(P && index == 0 && data && size == (uint32_t)(P->batch * P->feat), "pred_get_output: bad argument");
  const int cp = efm_pad4(P->feat);
  if (hipMemcpy2DAsync(data, (size_t)P->feat * 4, P->feat_dev, (size_t)cp * 4, (size_t)P->feat * 4, P->batch, hipMemcpyDeviceToHost,
                       P->stream) != hipSuccess)
    return EFM_E_LAUNCH;
  return hipStreamSynchronize(P->stream) == hipSuccess ? EFM_OK : EFM_E_LAUNCH;  // blocking, like MXPredGetOutput
}

int efm_pred_free(void* handle) {
  delete (Predictor*)handle;
  return EFM_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// MXNet c_predict_api names and signatures (include/c_predict_api.h) over the predictor above: the symbols Feature.hpp links to.
// ------------------------------------------------------------------------------------------------------------------
const char* MXGetLastError(void) { return efm_last_error_string(); }

int MXPredCreatePartialOut(const char* symbol_json_str, const void* param_bytes, int param_size, int dev_type, int dev_id,
                           mx_uint num_input_nodes, const char** input_keys, const mx_uint* input_shape_indptr,
                           const mx_uint* input_shape_data, mx_uint num_output_nodes, const char** output_keys, PredictorHandle* out) {
  if (dev_type != 2) {  // 1 = cpu, 2 = gpu (Feature.hpp:165): this library has no CPU path
    efm::set_error("MXPredCreate: dev_type %d refused (2 = gpu is the only device type)", dev_type);
    return -1;
  }
  if (num_output_nodes > 1 || (num_output_nodes == 1 && (!output_keys || !output_keys[0]))) {
    efm::set_error("MXPredCreatePartialOut: one output node (the 342-d feature) is available, %u requested", num_output_nodes);
    return -1;
  }
  if (num_output_nodes == 1) {
    const std::string k(output_keys[0]);
    if (k != "concat29" && k != "concat29_output") {  // MXNet appends "_output" to the key itself
      efm::set_error("MXPredCreatePartialOut: unknown output node '%s' (the feature node is 'concat29')", k.c_str());
      return -1;
    }
  }
  return efm_pred_create(symbol_json_str, param_bytes, param_size, dev_id, num_input_nodes, input_keys, input_shape_indptr, input_shape_data,
                         out) == EFM_OK ? 0 : -1;
}

int MXPredCreate(const char* symbol_json_str, const void* param_bytes, int param_size, int dev_type, int dev_id, mx_uint num_input_nodes,
                 const char** input_keys, const mx_uint* input_shape_indptr, const mx_uint* input_shape_data, PredictorHandle* out) {
  return MXPredCreatePartialOut(symbol_json_str, param_bytes, param_size, dev_type, dev_id, num_input_nodes, input_keys, input_shape_indptr,
                                input_shape_data, 0, nullptr, out);
}

int MXPredGetOutputShape(PredictorHandle handle, mx_uint index, mx_uint** shape_data, mx_uint* shape_ndim) {
  return efm_pred_get_output_shape(handle, index, shape_data, shape_ndim) == EFM_OK ? 0 : -1;
}

int MXPredSetInput(PredictorHandle handle, const char* key, const mx_float* data, mx_uint size) {
  return efm_pred_set_input(handle, key, data, size) == EFM_OK ? 0 : -1;
}

int MXPredForward(PredictorHandle handle) { return efm_pred_forward(handle) == EFM_OK ? 0 : -1; }

int MXPredGetOutput(PredictorHandle handle, mx_uint index, mx_float* data, mx_uint size) {
  return efm_pred_get_output(handle, index, data, size) == EFM_OK ? 0 : -1;
}

int MXPredFree(PredictorHandle handle) { return efm_pred_free(handle) == EFM_OK ? 0 : -1; }

}  // extern "C"
