// The consumer side of the reference's deployment code, trimmed to its MXNet calls: the two functions below keep the call
// sequence, argument values and buffer handling of feature_extraction/c_version/Feature.hpp:163-187 (Feature_Net) and :189-205
// (Feature_Extract_exe), minus OpenCV / BufferFile (file bytes come from fread, the "image" is a float array).
// It includes ONLY include/c_predict_api.h and is built with plain g++ — no HIP, no torch, no efm_* names: what links here is what
// Feature.hpp would link to.   g++ -std=c++11 -I include tests/c_abi/feature_consumer.cpp -L <pkg> -lefm_hip -o feature_consumer
//   usage: feature_consumer <EFM_RES.params> <image size S> <in: S*S floats> <out: 342 floats> [layer]
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "c_predict_api.h"

static PredictorHandle Feature_Net(const std::vector<char>& json, const std::vector<char>& params, int IMG_SIZE, char* layer) {
  PredictorHandle pred_hnd = 0;
  int dev_type = 2;  // 1: cpu, 2: gpu
  int dev_id = 0;
  const char* input_key[1] = {"data"};
  const char** input_keys = input_key;
  const mx_uint input_shape_indptr[2] = {0, 4};
  const mx_uint input_shape_data[4] = {1, static_cast<mx_uint>(1), static_cast<mx_uint>(IMG_SIZE), static_cast<mx_uint>(IMG_SIZE)};
  MXPredCreatePartialOut((const char*)json.data(), (const char*)params.data(), static_cast<int>(params.size()), dev_type, dev_id, 1,
                         input_keys, input_shape_indptr, input_shape_data, 1, (const char**)&layer, &pred_hnd);
  return pred_hnd;
}

static void Feature_Extract_exe(const std::vector<mx_float>& image_datas, float* Feature_Vector, PredictorHandle pred_hnd) {
  MXPredSetInput(pred_hnd, "data", image_datas.data(), static_cast<mx_uint>(image_datas.size()));
  MXPredForward(pred_hnd);
  mx_uint output_index = 0;
  mx_uint* shape = 0;
  mx_uint shape_length;
  MXPredGetOutputShape(pred_hnd, output_index, &shape, &shape_length);
  size_t size = 1;
  for (mx_uint i = 0; i < shape_length; ++i) size *= shape[i];
  std::vector<float> data(size);
  MXPredGetOutput(pred_hnd, output_index, &(data[0]), static_cast<mx_uint>(size));
  for (size_t i = 0; i < size; ++i) Feature_Vector[i] = data[i];
}

static std::vector<char> slurp(const char* path) {
  std::vector<char> buf;
  FILE* f = std::fopen(path, "rb");
  if (!f) return buf;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  buf.resize((size_t)n);
  if (std::fread(buf.data(), 1, (size_t)n, f) != (size_t)n) buf.clear();
  std::fclose(f);
  return buf;
}

int main(int argc, char** argv) {
  if (argc < 5) {
    std::printf("usage: %s params S in.f32 out.f32 [layer]\n", argv[0]);
    return 2;
  }
  const int S = std::atoi(argv[2]);
  std::vector<char> params = slurp(argv[1]), raw = slurp(argv[3]);
  std::vector<char> json(3, 0);
  json[0] = '{'; json[1] = '}';
  char default_layer[] = "concat29";
  char* layer = argc > 5 ? argv[5] : default_layer;
  if (params.empty() || raw.size() != (size_t)S * S * sizeof(float)) return 2;
  // a predictor for the CPU device type must be refused, not silently served
  {
    PredictorHandle h = 0;
    const char* key[1] = {"data"};
    const mx_uint indptr[2] = {0, 4}, shp[4] = {1, 1, (mx_uint)S, (mx_uint)S};
    if (MXPredCreate(json.data(), params.data(), (int)params.size(), 1, 0, 1, key, indptr, shp, &h) != -1 || h != 0) return 4;
    std::printf("dev_type 1 refused: %s\n", MXGetLastError());
  }
  PredictorHandle pred = Feature_Net(json, params, S, layer);
  if (!pred) {
    std::printf("MXPredCreatePartialOut failed: %s\n", MXGetLastError());
    return 3;
  }
  std::vector<mx_float> image((size_t)S * S);
  for (size_t i = 0; i < image.size(); ++i) image[i] = reinterpret_cast<const float*>(raw.data())[i];
  const int fvSize = 342;  // Feature.hpp:24
  std::vector<float> fv(fvSize), fv2(fvSize);
  Feature_Extract_exe(image, fv.data(), pred);
  Feature_Extract_exe(image, fv2.data(), pred);  // second call: the captured graph replays
  for (int i = 0; i < fvSize; ++i)
    if (fv[i] != fv2[i]) return 5;
  FILE* f = std::fopen(argv[4], "wb");
  std::fwrite(fv.data(), sizeof(float), fv.size(), f);
  std::fclose(f);
  MXPredFree(pred);
  std::printf("feature_consumer: OK\n");
  return 0;
}
