/* MXNet's c_predict_api entry points, exported by libefm_hip.so with MXNet's exact C signatures, so that the reference's
 * deployment code (feature_extraction/c_version/Feature.hpp) relinks against this library WITHOUT source edits:
 *
 *   Feature.hpp:163-187  Feature_Net()          -> MXPredCreatePartialOut (dev_type 2 = gpu, one input "data" of shape
 *                                                   (1, 1, S, S), one output key = Configs["Feature_Layer"])
 *   Feature.hpp:189-205  Feature_Extract_exe()  -> MXPredSetInput / MXPredForward / MXPredGetOutputShape / MXPredGetOutput
 *   (MXPredFree / MXGetLastError complete the handle's life cycle)
 *
 * Signatures follow MXNet 1.x include/mxnet/c_predict_api.h (the reference pins no version; the header is not vendored in
 * /root/reference, so these are restated from the published API).  Return 0 on success, -1 on failure with the message in
 * MXGetLastError(), as MXNet does.  Thin shim over the efm_pred_* entry points of efm_hip.h (csrc/efm_predict.hip):
 * network = Symbol EFM-29 (efm_symbol.py:22-101); the symbol JSON is accepted and ignored; dev_type 1 (cpu) is REFUSED —
 * there is no CPU path; the only output node is the 342-d post-fc1 EFM feature, named "concat29" / "concat29_output"
 * (final_efm.py:208, Feature.hpp:24 fvSize = 342). */
#ifndef EFM_C_PREDICT_API_H_
#define EFM_C_PREDICT_API_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned int mx_uint;
typedef float mx_float;
typedef void* PredictorHandle;

const char* MXGetLastError(void);
int MXPredCreate(const char* symbol_json_str, const void* param_bytes, int param_size, int dev_type, int dev_id,
                 mx_uint num_input_nodes, const char** input_keys, const mx_uint* input_shape_indptr,
                 const mx_uint* input_shape_data, PredictorHandle* out);
int MXPredCreatePartialOut(const char* symbol_json_str, const void* param_bytes, int param_size, int dev_type, int dev_id,
                           mx_uint num_input_nodes, const char** input_keys, const mx_uint* input_shape_indptr,
                           const mx_uint* input_shape_data, mx_uint num_output_nodes, const char** output_keys,
                           PredictorHandle* out);
int MXPredGetOutputShape(PredictorHandle handle, mx_uint index, mx_uint** shape_data, mx_uint* shape_ndim);
int MXPredSetInput(PredictorHandle handle, const char* key, const mx_float* data, mx_uint size);
int MXPredForward(PredictorHandle handle);
int MXPredGetOutput(PredictorHandle handle, mx_uint index, mx_float* data, mx_uint size);
int MXPredFree(PredictorHandle handle);

#ifdef __cplusplus
}
#endif
#endif /* EFM_C_PREDICT_API_H_ */
