/*
 * efm_hip.h — C ABI of the MI355X (gfx950) hot path of the EFM triplet-loss trainer.
 *
 * The reference (joannhsiao/Improving_Face_recognition_Performance_using_Triplet_Loss)
 * has NO plugin / FFI / custom-op layer: its hot path is a sequence of stock MXNet
 * operator calls.  Every entry point below therefore replaces an MXNet operator *call
 * site* of the reference; the call site is cited as "ref: file:line".
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless it says "host";
 *   - the caller owns every buffer (activations, packed weights, workspaces);
 *     the library never allocates or frees device memory and keeps no mutable state;
 *   - every function enqueues asynchronously on the hipStream_t passed as `stream`
 *     (a void* here so that the header needs no HIP include) and returns at once;
 *   - return value: 0 = EFM_OK, negative = EFM_E_*; efm_last_error_string() gives the
 *     thread-local text of the last failure.  No C++ exception crosses this boundary.
 *
 * Tensor layouts (all fp32)
 *   - activations: NHWC with the channel stride padded to a multiple of 4,
 *       x[b][h][w][cp], cp = efm_pad4(c); channels c..cp-1 are ZERO (an invariant every
 *       kernel keeps: pads are written as zeros, never read as data).
 *   - packed conv weights ("OHWI, padded"): w[n][k], n < n_pad16 = pad16(cout),
 *       k = (kh*KW + kw)*cin_p + ci < k_pad = pad16(KH*KW*cin_p); everything outside
 *       (cout, KH, KW, cin) is zero.  MXNet's own layout is (cout, cin, KH, KW)
 *       (ref: efm_symbol.py:32 `mx.symbol.Convolution`); efm_conv_pack_weights /
 *       efm_conv_unpack_weights convert between the two.
 */
#ifndef EFM_HIP_H_
#define EFM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EFM_OK 0
#define EFM_E_INVALID (-1)     /* bad argument / shape the kernels do not support */
#define EFM_E_LAUNCH (-2)      /* hipGetLastError() after a launch was not hipSuccess */
#define EFM_E_WORKSPACE (-3)   /* workspace pointer null or too small */

#define EFM_ABI_VERSION 1

/* MFM tie rule: which operand wins when two slices are exactly equal.
 * ORDER_GROUP:  maximum(maximum(s0,s1), s2)  (ref: efm_symbol.py:70-73, lightcnn.py:23-26)
 * ORDER_RES:    maximum(s2, maximum(s0,s1))  (ref: efm_symbol.py:26-29)
 * MXNet's backward of maximum/minimum(lhs,rhs) sends the gradient to lhs on a tie. */
#define EFM_MFM_ORDER_GROUP 0
#define EFM_MFM_ORDER_RES 1

/* L2-normalisation modes. ROW: y[i] = x[i]/||x[i]||   (ref: final_efm.py:240-243)
 *                         FROBENIUS: y = x/||x||_F     (ref: train_efm.py:241) */
#define EFM_L2_ROW 0
#define EFM_L2_FROBENIUS 1

int efm_version(void);
const char* efm_last_error_string(void);

static inline int efm_pad4(int c) { return (c + 3) & ~3; }
static inline int efm_pad16(int c) { return (c + 15) & ~15; }

/* ------------------------------------------------------------------------------------
 * Convolution (stride 1, cross-correlation, bias) — ref: mx.symbol.Convolution at
 * efm_symbol.py:32,41,54,62,65,67 and nn.Conv2D at lightcnn.py:14-15,47-48.
 * Also serves FullyConnected (ref: efm_symbol.py:94 fc1, pre-trained_efm_v3.py:181
 * Dense(128)) as a KHxKW "valid" convolution whose output map is 1x1.
 * ------------------------------------------------------------------------------------ */
typedef struct efm_conv_desc {
  int32_t batch;
  int32_t hin, win, cin, cin_p;      /* input map; cin_p = efm_pad4(cin) = channel stride of x */
  int32_t hout, wout, cout, cout_p;  /* output map; cout_p = efm_pad4(cout) = channel stride of y */
  int32_t kh, kw, pad_h, pad_w;
  int32_t n_pad16;                   /* rows of the packed weight = efm_pad16(cout) */
  int32_t k_pad;                     /* row length of the packed weight = efm_pad16(kh*kw*cin_p) */
  int32_t dn_pad16;                  /* rows of the packed dgrad weight = efm_pad16(cin) */
  int32_t dk_pad;                    /* its row length = efm_pad16(kh*kw*cout_p) */
  /* Optional tiling choice of efm_conv_fwd / efm_conv_bwd_data: 0 = built-in heuristic, else MT | (nsplit << 4) with
   * MT in {1,2} (64- or 128-pixel tiles) and nsplit = number of channel blocks.  Any choice gives bit-identical results
   * (the K order of every output element is fixed); the host may time the candidates once and store the winner here. */
  int32_t tune_fwd, tune_dgrad;
  /* Same for efm_conv_bwd_weight (and its workspace size, which depends on it): 0 = heuristic, else KPW | (blocks64 << 4) with
   * KPW in {1,2} (64 or 128 weight columns per block) and blocks64 = target number of thread blocks / 64 (0 = default 2560).
   * Every choice is deterministic; different choices differ by fp32 summation order.
   * Bit 12 (0x1000), 3x3 / pad 1 / stride 1 geometries only: the Winograd form (efm_wino_bwd_weight's kernel) behind the same entry
   * points — 2.25x fewer multiplies, fp32 rounding differs from the direct kernel by ~1e-6; bits 3:0 = 1 + block shape (output x input
   * channel tiles of 16, 0 = the shape that pads the layer least), bits 9:4 = blocks to aim for / 64 (0 = one per CU). */
  int32_t tune_wgrad;
} efm_conv_desc;

/* Fill every derived field (hout = hin + 2*pad_h - kh + 1, paddings, packed sizes). */
int efm_conv_desc_init(efm_conv_desc* d, int batch, int hin, int win, int cin, int cout,
                       int kh, int kw, int pad_h, int pad_w);

size_t efm_conv_weight_elems(const efm_conv_desc* d);        /* n_pad16 * k_pad  */
size_t efm_conv_dgrad_weight_elems(const efm_conv_desc* d);  /* dn_pad16 * dk_pad */
size_t efm_conv_wgrad_workspace_bytes(const efm_conv_desc* d);

/* (cout,cin,KH,KW) <-> packed. */
int efm_conv_pack_weights(const efm_conv_desc* d, const float* w_oihw, float* w_packed, void* stream);
int efm_conv_unpack_weights(const efm_conv_desc* d, const float* w_packed, float* w_oihw, void* stream);
/* packed forward weight -> packed, tap-flipped, transposed weight used by efm_conv_bwd_data. */
int efm_conv_make_dgrad_weights(const efm_conv_desc* d, const float* w_packed, float* wd_packed, void* stream);

/* y = conv(x, w) + bias (+ residual).  bias[n_pad16] or NULL; residual has y's shape or NULL
 * (ref: efm_symbol.py:42 `data + conv_r1`). */
int efm_conv_fwd(const efm_conv_desc* d, const float* x, const float* w_packed, const float* bias,
                 const float* residual, float* y, void* stream);
/* Fused forward: z = [maxpool2x2(] MFM(conv(x, w) + bias) [)] in ONE launch — the conv result never reaches HBM.
 * Replaces the Convolution -> SliceChannel/maximum/minimum/Concat [-> Pooling] chains of group()/res_block()
 * (ref: efm_symbol.py:32-39, 54-60, 65-78).  z: [batch][h'][w'][pad4(c')] with c' = 2*cout/3 (ways 3) or cout/2 (ways 2)
 * and (h', w') = (hout/2, wout/2) when pool else (hout, wout).  route: one byte per element of z (same shape) recording the
 * slice (and, with pooling, the window pixel: 4*pixel + slice) the value came from, with MXNet's tie rules.
 * Wide layers are cut into channel blocks that each own a channel range of EVERY slice (weight rows permuted on the fly;
 * number of blocks = d->tune_fwd >> 4, 0 = heuristic), so any cout is supported: efm_conv_mfm_supported(d) != 0. */
int efm_conv_mfm_supported(const efm_conv_desc* d);
int efm_conv_mfm_fwd(const efm_conv_desc* d, const float* x, const float* w_packed, const float* bias, float* z,
                     unsigned char* route, int ways, int order, int pool, void* stream);
/* Backward of that epilogue: dy[batch][h][w][pad4(c)] (the full conv-output gradient, every element written) from dz and
 * route; (h, w, c) are the conv OUTPUT dims. */
int efm_mfm_pool_bwd(const unsigned char* route, const float* dz, float* dy, int batch, int h, int w, int c, int ways,
                     int pool, void* stream);
/* dx = conv_transpose(dy, w) (+ add).  `add` has dx's shape or NULL (skip-path gradient). */
int efm_conv_bwd_data(const efm_conv_desc* d, const float* dy, const float* wd_packed,
                      const float* add, float* dx, void* stream);
/* dw_packed[n][k] (+)= sum_m dy[m][n] * im2col(x)[m][k]; dbias[n] (+)= sum_m dy[m][n] (dbias may be NULL).
 * accumulate != 0 adds to the existing contents (weight sharing: the Gluon res_block re-applies the same two
 * convolutions, ref: lightcnn.py:47-48,52-69).  Deterministic: split over m into workspace slabs, then a
 * fixed-order reduction. */
int efm_conv_bwd_weight(const efm_conv_desc* d, const float* x, const float* dy, float* dw_packed,
                        float* dbias, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* The two launches of efm_conv_bwd_weight separately: `_slabs` = the matrix-core kernel (partial gradients per pixel split + bias
 * partials into the workspace), `_finish` = their fixed-order reduction into dw_packed / dbias (dbias NULL = no bias gradient; then
 * `_slabs` must have been called with want_bias = 0 or its bias partials are simply ignored).  A caller may enqueue `_finish` on
 * another stream (ordered after `_slabs` by an event) to run it under the next layer's matrix-core kernel; the workspace must stay
 * untouched in between.  efm_conv_bwd_weight == _slabs; _finish on one stream. */
int efm_conv_bwd_weight_slabs(const efm_conv_desc* d, const float* x, const float* dy, int want_bias, void* workspace,
                              size_t workspace_bytes, void* stream);
int efm_conv_bwd_weight_finish(const efm_conv_desc* d, float* dw_packed, float* dbias, int accumulate, const void* workspace,
                               size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * bf16 tensor-core path (BASELINE configs[2]: "bf16 — MFMA conv path").  Same operators and the same reference call sites as
 * the fp32 entry points above (mx.symbol.Convolution efm_symbol.py:32,41,54,62,65,67, FullyConnected :94, nn.Conv2D
 * lightcnn.py:14-15,47-48, the MFM idiom efm_symbol.py:25-30,63-64, Pooling :78) — the reference has no reduced-precision
 * mode of its own, this is MXNet AMP's role (`amp.init()` + bf16 cast of a Symbol).  Different storage:
 *   activations  bf16 NHWC, channel stride pad8(c), pads zero;        (uint16_t* here = raw bf16 bits)
 *   weights      bf16 packed wb[n][k], k = tap*pad8(cin) + ci, pad16(cout) rows of pad32(taps*pad8(cin)) elements,
 *                cast every step from the fp32 master weights (which stay in the fp32 packed layout, as do all gradients);
 *   arithmetic   v_mfma_f32_16x16x32_bf16, fp32 accumulate; the fused epilogue (bias, MFM, pool) runs on the fp32 accumulators.
 * The weight gradient contracts over pixels: its operands are read from the [pixel][channel] LDS image with
 * ds_read_b64_tr_b16 (hardware transpose).
 * ------------------------------------------------------------------------------------ */
size_t efm_convb_weight_elems(const efm_conv_desc* d);
size_t efm_convb_dgrad_weight_elems(const efm_conv_desc* d);
size_t efm_convb_wgrad_workspace_bytes(const efm_conv_desc* d);
int efm_nchw_to_nhwc_bf16(const float* x_nchw, uint16_t* y_nhwc_bf16, int batch, int c, int h, int w, void* stream);
/* fp32 packed master weight -> bf16 forward weight and (wdb != NULL) bf16 data-gradient weight */
int efm_convb_cast_weights(const efm_conv_desc* d, const float* w_packed, uint16_t* wb, uint16_t* wdb, void* stream);
int efm_convb_fwd(const efm_conv_desc* d, const uint16_t* x, const uint16_t* wb, const float* bias, const uint16_t* residual,
                  uint16_t* y, void* stream);
/* z is bf16 (channel stride pad8) or, with out_f32 != 0, float (stride pad4): the layer that feeds the fp32 head */
int efm_convb_mfm_fwd(const efm_conv_desc* d, const uint16_t* x, const uint16_t* wb, const float* bias, void* z, unsigned char* route,
                      int ways, int order, int pool, int out_f32, void* stream);
int efm_convb_bwd_data(const efm_conv_desc* d, const uint16_t* dy, const uint16_t* wdb, const uint16_t* add, uint16_t* dx, void* stream);
int efm_convb_mfm_pool_bwd(const unsigned char* route, const void* dz, int dz_f32, uint16_t* dy, int batch, int h, int w, int c, int ways,
                           int pool, void* stream);
/* dw_packed / dbias are FLOAT, in the fp32 packed layout of efm_conv_bwd_weight */
int efm_convb_bwd_weight(const efm_conv_desc* d, const uint16_t* x, const uint16_t* dy, float* dw_packed, float* dbias, int accumulate,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Weight gradient of a convolution whose fused epilogue was bias -> MFM2 -> 2x2 max pooling (efm_convb_mfm_fwd, ways 2, pool 1), taken
 * straight from the epilogue's output gradient dz [b][hout/2][wout/2][pad8(cout/2)] (bf16) and its route bytes: the conv-output
 * gradient (ref: the backward of SliceChannel / maximum / Pooling, efm_symbol.py:62-64,76-78) is formed in LDS inside the kernel and
 * never written to HBM.  For layers whose input needs no gradient (the first convolution) this replaces efm_convb_mfm_pool_bwd +
 * efm_convb_bwd_weight; efm_convb_mfm_bwd_weight_supported says whether a layer qualifies.  Workspace: efm_convb_wgrad_workspace_bytes. */
int efm_convb_mfm_bwd_weight_supported(const efm_conv_desc* d, int ways, int pool);
int efm_convb_mfm_bwd_weight(const efm_conv_desc* d, const uint16_t* x, const unsigned char* route, const uint16_t* dz, int ways, int pool,
                             float* dw_packed, float* dbias, int accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------
 * Layout conversion at the boundary (ImageRecordIter emits NCHW — ref: train_efm.py:179).
 * ------------------------------------------------------------------------------------ */
int efm_nchw_to_nhwc(const float* x_nchw, float* y_nhwc, int batch, int c, int h, int w, void* stream);
/* Row-packed input of the FIRST convolution (the 5x5 on 3 / 1 channels, ref: efm_symbol.py:84 group(data, 0, 99, (5,5), ...),
 * lightcnn.py:82 efm(0, 99, (5,5), ...)): y[b][h][w][j*c + ch] = x[b][ch][h][w + j - pad_w] (0 outside the row), j < kw, channel
 * stride pad4(kw*c) floats (bf16 = 0) or pad8(kw*c) bf16 (bf16 = 1).  A kh x kw convolution on c channels equals the kh x 1
 * convolution (pad (pad_h, 0)) of y with the weights re-indexed w'[n][j*c + ch][kh] = w[n][ch][kh][j]: MXNet's Convolution forms
 * the same im2col columns internally; here K = kh*kw*c packs densely instead of padding every tap's 3 channels to 4 / 8. */
int efm_rowpack_nchw(const float* x_nchw, void* y_rowpacked, int batch, int c, int h, int w, int kw, int pad_w, int bf16, void* stream);
/* mx.io.ImageRecordIter's augmentation on the device (ref: train_efm.py:179-181: scale=1./255, rand_crop, rand_mirror): decoded uint8
 * images src[b][ih][iw][c] (HWC, as an image decoder leaves them) -> dst[b][c][h][w] fp32 = scale * crop(mirror?(src)).
 * crop[b] = (y0, x0, mirror 0/1), drawn by the caller (the host iterator); the window must fit: y0 + h <= ih, x0 + w <= iw. */
int efm_crop_mirror_u8(const uint8_t* src_hwc, const int32_t* crop, float* dst_nchw, int batch, int ih, int iw, int c, int h, int w,
                       float scale, void* stream);
int efm_nhwc_to_nchw(const float* x_nhwc, float* y_nchw, int batch, int c, int h, int w, void* stream);

/* ------------------------------------------------------------------------------------
 * MFM — SliceChannel + maximum/minimum + Concat.
 * ways = 3: y[:, 0:c/3] = max(x0,x1,x2), y[:, c/3:2c/3] = min(x0,x1,x2)
 *           (ref: efm_symbol.py:25-30,34-39,55-60,69-74,96-101; lightcnn.py:22-27).
 * ways = 2: y = max(x[:, :c/2], x[:, c/2:])   (ref: efm_symbol.py:63-64,76-77).
 * x: [rows][efm_pad4(c)], y: [rows][efm_pad4(c_out)], c_out = 2c/3 or c/2.
 * bwd: dx = dMFM(x, dy) (+ add), gradient to the arg-max / arg-min slice.
 * ------------------------------------------------------------------------------------ */
int efm_mfm_fwd(const float* x, float* y, int64_t rows, int c, int ways, void* stream);
int efm_mfm_bwd(const float* x, const float* dy, const float* add, float* dx, int64_t rows, int c,
                int ways, int order, void* stream);

/* The same MFM on bf16 activations (channel stride pad8): the residual-block inputs of EFM-29 under the bf16 plan. */
int efm_mfmb_fwd(const uint16_t* x, uint16_t* y, int64_t rows, int c, int ways, void* stream);
int efm_mfmb_bwd(const uint16_t* x, const uint16_t* dy, const uint16_t* add, uint16_t* dx, int64_t rows, int c,
                 int ways, int order, void* stream);

/* Max pooling 2x2 stride 2, 'valid' (floor) — ref: efm_symbol.py:78, lightcnn.py:83. */
int efm_maxpool2_fwd(const float* x, float* y, int batch, int h, int w, int c, void* stream);
int efm_maxpool2_bwd(const float* x, const float* dy, float* dx, int batch, int h, int w, int c, void* stream);

/* ------------------------------------------------------------------------------------
 * Winograd F(2x2, 3x3) form of the 3x3 / pad 1 / stride 1 convolutions (same call sites as efm_conv_fwd / efm_conv_bwd_data:
 * efm_symbol.py:32,41,54,65,67) — 2.25x fewer multiplies, fp32, input / output transforms fused into the kernel.
 * `u` = transformed weights U = G g G^T, made from the packed fp32 weights of efm_conv_pack_weights whenever they change:
 * dgrad = 0 for efm_wino_fwd, dgrad = 1 (tap-flipped transpose) for efm_wino_bwd_data; efm_wino_u_elems(d, dgrad) floats.
 * Results equal the direct kernels' to fp32 rounding (different summation order), not bitwise.
 * Bits 9:8 of tune_fwd (forward) / tune_dgrad (data gradient) select the kernel variant and with it U's layout
 * (0 default, 1 = 8-wave blocks, 2 = 4-wave blocks): keep the field unchanged between efm_wino_make_u and the launch.
 */
int efm_wino_supported(const efm_conv_desc* d);
size_t efm_wino_u_elems(const efm_conv_desc* d, int dgrad);
int efm_wino_make_u(const efm_conv_desc* d, const float* w_packed, float* u, int dgrad, void* stream);
int efm_wino_fwd(const efm_conv_desc* d, const float* x, const float* u, const float* bias, const float* residual,
                 float* y, void* stream);
int efm_wino_bwd_data(const efm_conv_desc* d, const float* dy, const float* u_dgrad, const float* add, float* dx,
                      void* stream);
/* Winograd forward with the fused bias -> MFM (-> 2x2 max pooling) epilogue: same z / route outputs and tie rules as
 * efm_conv_mfm_fwd (so efm_mfm_pool_bwd is its backward), U made by efm_wino_mfm_make_u (rows grouped so that every slice of a
 * channel meets in one block). */
/* Weight gradient in Winograd form (same outputs, workspace protocol and determinism as efm_conv_bwd_weight; also reachable through
 * efm_conv_bwd_weight{,_slabs,_finish} with bit 12 of tune_wgrad, which is how the training plan uses it). */
size_t efm_wino_wgrad_workspace_bytes(const efm_conv_desc* d);
int efm_wino_bwd_weight(const efm_conv_desc* d, const float* x, const float* dy, float* dw_packed, float* dbias,
                        int accumulate, void* workspace, size_t workspace_bytes, void* stream);
size_t efm_wino_mfm_u_elems(const efm_conv_desc* d, int ways);
int efm_wino_mfm_make_u(const efm_conv_desc* d, const float* w_packed, float* u, int ways, void* stream);
/* n transformed-weight tensors in ONE launch (the weights of a step are fixed): job q = efm_wino_make_u(descs[q], w_packed[q], u[q],
 * dgrad[q]) when ways[q] == 0, efm_wino_mfm_make_u(descs[q], w_packed[q], u[q], ways[q]) otherwise (then dgrad[q] must be 0).
 * The arrays are HOST arrays of device pointers / flags; results are identical to the per-layer calls. */
int efm_wino_make_u_batch(int n, const efm_conv_desc* const* descs, const float* const* w_packed, float* const* u, const int* dgrad,
                          const int* ways, void* stream);
int efm_wino_mfm_fwd(const efm_conv_desc* d, const float* x, const float* u, const float* bias, float* z,
                     unsigned char* route, int ways, int order, int pool, void* stream);

/* ------------------------------------------------------------------------------------
 * Embedding head / loss (dense row-major matrices, leading dimension = ld* floats).
 * ------------------------------------------------------------------------------------ */
/* ref: train_efm.py:241 (FROBENIUS), final_efm.py:240-243 (ROW).  norm_out: [rows] (ROW) or [1]. */
int efm_l2norm_fwd(const float* x, float* y, float* norm_out, int rows, int d, int ldx, int ldy, int mode, void* stream);
int efm_l2norm_bwd(const float* y, const float* norm, const float* dy, float* dx, int rows, int d,
                   int ldy, int lddy, int lddx, int mode, void* stream);
/* y[i] = x[idx[i]]  — the reference's negative pick copies rows (ref: train_efm.py:234-239). */
int efm_gather_rows(const float* x, const int32_t* idx, float* y, int rows, int d, int ldx, int ldy, void* stream);
/* loss[i] = max(0, sum_d (p-a)^2 - sum_d (n-a)^2 + margin) — gluon.loss.TripletLoss
 * (ref: train_efm.py:210,241; pre-trained_efm_v3.py:183,210). */
int efm_triplet_fwd(const float* a, const float* p, const float* n, float* loss, int rows, int d,
                    int lda, int ldp, int ldn, float margin, void* stream);
/* da/dp/dn may each be NULL (dn is NULL in the reference: negatives are detached). */
int efm_triplet_bwd(const float* a, const float* p, const float* n, const float* loss, const float* gloss,
                    float* da, float* dp, float* dn, int rows, int d, int lda, int ldp, int ldn, int ldg,
                    void* stream);
/* The same loss with every row of e an anchor and positives / negatives given as row indices (in-batch mining, north star):
 * loss[i] = relu(|e_i - e_pos[i]|^2 - |e_i - e_neg[i]|^2 + margin), 0 where neg[i] < 0.  pos must be a permutation of the rows
 * (inv_pos its inverse, -1 where a row is nobody's positive); negatives are detached as in the reference, so
 * de[i] = 2 g_i (e_neg[i] - e_pos[i]) + 2 g_j (e_i - e_j), j = inv_pos[i] — one writer per row, no atomics. */
int efm_triplet_indexed_fwd(const float* e, const int32_t* pos, const int32_t* neg, float* loss, int rows, int d, int lde,
                            float margin, void* stream);
int efm_triplet_indexed_bwd(const float* e, const int32_t* pos, const int32_t* neg, const int32_t* inv_pos, const float* loss,
                            const float* gloss, float* de, int rows, int d, int lde, int ldg, void* stream);
/* s_ap[i] = cos(a_i,p_i), s_an[i] = cos(a_i,n_i) — cosine_dist (ref: train_efm.py:26-34). */
int efm_cosine_pairs(const float* a, const float* p, const float* n, float* s_ap, float* s_an, int rows,
                     int d, int lda, int ldp, int ldn, void* stream);
/* Verification-pair distances for the LFW protocol (ref: feature_extraction/facenet_version/facenet.py:412-426 `distance`):
 * sqdist[i] = sum_d (a_i - b_i - (mean_a - mean_b))^2 ... with mean == NULL: sum_d (a_i - b_i)^2;
 * cosine[i] = <a_i - mean, b_i - mean> / (|a_i - mean| |b_i - mean|).  `mean` is a [d] vector or NULL. */
int efm_pair_distance(const float* a, const float* b, const float* mean, float* sqdist, float* cosine, int rows, int d,
                      int lda, int ldb, void* stream);
/* Gallery scan of the deployment side: scores[q][i] = <query_q, gallery_i> (cosine for unit-norm features) — the batched
 * form of simd_dot + the per-row loop of Compare_Face_From_DB (ref: Feature.hpp:273-293,345-392).  nq*d*4 <= 64 KiB. */
int efm_gallery_scores(const float* query, const float* gallery, float* scores, int nq, int n, int d, int ldq, int ldg, void* stream);
/* g[i][j] = cos(e_i, e_j): the batch-all-pairs cosine matrix (north_star mining path; no reference). */
int efm_gram_cosine(const float* e, float* g, int rows, int d, int lde, void* stream);
/* Semi-hard negative per (anchor i, positive pos[i]) from the cosine matrix g[rows][rows]:
 * d = 1 - g; pick argmin_{label!=, d_an > d_ap} d_an, else argmax_{label!=} d_an (TF-addons rule).
 * neg_idx[i] = -1 when the batch holds a single identity. */
int efm_mine_semihard(const float* g, const int32_t* labels, const int32_t* anchor_idx,
                      const int32_t* pos_idx, int32_t* neg_idx, int n_anchor, int rows, void* stream);

/* ------------------------------------------------------------------------------------
 * Diagnostic (no reference call site: bench.py's roofline accounting).  Which kernel instance a convolution launch resolves
 * to under the descriptor's tuning fields, and the matrix-core flops that launch EXECUTES (padded tiles; Winograd: the 16
 * transformed-domain GEMMs), as opposed to the algorithmic 2*M*cout*cin*kh*kw.  pass: 0 forward, 1 forward + fused MFM(ways)
 * (+ pool) epilogue, 2 data gradient, 3 weight gradient, 4 / 5 / 6 = Winograd forward / fused forward / data gradient.
 * ------------------------------------------------------------------------------------ */
int efm_conv_kernel_info(const efm_conv_desc* d, int pass, int ways, int pool, char* name, size_t name_len, double* mfma_flops);

/* ------------------------------------------------------------------------------------
 * Optimiser on the flat packed parameter buffer.
 * SGD: w -= lr*(rescale*g + wd*w)                    (ref: pre-trained_efm_v3.py:185,212)
 * Adam (MXNet form): g' = rescale*g + wd*w; m,v EMA; w -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)
 *                                                    (ref: train_efm.py:213, mutli_gpu_v3.py:159)
 * ------------------------------------------------------------------------------------ */
int efm_sgd_update(float* w, const float* g, int64_t n, float lr, float wd, float rescale, void* stream);
int efm_adam_update(float* w, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                    float beta2, float eps, float wd, float rescale, int step, void* stream);

/* ------------------------------------------------------------------------------------
 * Predictor with the call shape of MXNet's c_predict_api — the consumer side the reference's deployment code uses
 * (ref: feature_extraction/c_version/Feature.hpp:163-205: MXPredCreatePartialOut / MXPredSetInput / MXPredForward /
 * MXPredGetOutputShape / MXPredGetOutput / MXPredFree).  Network = Symbol EFM-29 (efm_symbol.py:22-101), output 0 = the
 * 342-d 'concat29_output' feature (Feature.hpp:24).  param_bytes = contents of an MXNet `.params` file (float32).
 * symbol_json is accepted and ignored (the structure is fixed).  Host pointers in, host pointers out, blocking
 * get_output; a predictor owns its device buffers and its stream; one predictor per thread.
 * ------------------------------------------------------------------------------------ */
int efm_pred_create(const char* symbol_json, const void* param_bytes, int param_size, int dev_id, uint32_t num_input_nodes,
                    const char** input_keys, const uint32_t* input_shape_indptr, const uint32_t* input_shape_data,
                    void** out);
int efm_pred_set_input(void* handle, const char* key, const float* data, uint32_t size);
int efm_pred_forward(void* handle);
int efm_pred_get_output_shape(void* handle, uint32_t index, uint32_t** shape_data, uint32_t* shape_ndim);
int efm_pred_get_output(void* handle, uint32_t index, float* data, uint32_t size);
int efm_pred_free(void* handle);

#ifdef __cplusplus
}
#endif
#endif /* EFM_HIP_H_ */
