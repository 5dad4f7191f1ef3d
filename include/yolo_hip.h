/* yolo_hip.h — C-ABI of libyolo_hip.so, the MI355X (gfx950) replacement for the
 * conv-backbone + region-layer hot path of zhen8838/K210_Yolo_framework.
 *
 * Two groups of entry points:
 *
 *  (1) region_layer_*  — byte-for-byte drop-in for the reference's only native
 *      operator, yolo3_frame_test_public/region_layer.h:7-48.  Same four
 *      symbols, same struct layouts, same call order and error codes, so a
 *      main.c-style caller (main.c:278-324) links against this library
 *      unchanged.  The arithmetic runs on the GPU; rl->output / boxes / probs
 *      are mirrored back to the host buffers the reference exposes.
 *
 *  (2) yk_*            — the batched engine.  The reference has no batched
 *      native API, so these mirror the call shape its edge path already uses
 *      for the model run (kendryte SDK, un-vendored):
 *        kpu_load_kmodel(&task, blob)                 main.c:274  -> yk_plan_create
 *        kpu_run_kmodel(&task, img, dma, done_cb, 0)  main.c:303  -> yk_run_u8 / yk_run_f32 (async on a stream)
 *        kpu_get_output(&task, i, &ptr, &bytes)       main.c:310  -> yk_get_output
 *      and, for the Python path, the decode + per-class NMS block of
 *      keras_inference.py:94-135 (tf_xywh_to_all tools/utils.py:524-547,
 *      correct_box keras_inference.py:32-72)            -> yk_decode_py
 *      plus a batched form of region_layer_run           -> yk_region_batched
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success and a negative code on failure (region_layer_init keeps the
 * reference's -1..-4); pointers named d_* are DEVICE pointers, h_* host.
 * `stream` is a hipStream_t passed as void* (NULL = default stream).
 * No entry point falls back to a CPU implementation: without a usable HIP
 * device they fail with YK_ERR_NO_DEVICE.
 */
#ifndef YOLO_HIP_H_
#define YOLO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* (1) drop-in region layer — layouts from region_layer.h:7-42                */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint32_t obj_number;
    struct {
        uint32_t x1;
        uint32_t y1;
        uint32_t x2;
        uint32_t y2;
        uint32_t class_id;
        float prob;
    } obj[10];
} obj_info_t;

typedef struct {
    float threshold;          /* caller sets before init (main.c:280) */
    float nms_value;          /* caller sets before init (main.c:281) */
    uint32_t coords;
    uint32_t anchor_number;   /* caller sets before init (main.c:278) */
    float *anchor;            /* borrowed, 2*anchor_number floats (w,h) */
    uint32_t image_width;
    uint32_t image_height;
    uint32_t classes;
    uint32_t net_width;
    uint32_t net_height;
    uint32_t layer_width;
    uint32_t layer_height;
    uint32_t boxes_number;
    uint32_t output_number;
    void *boxes;              /* boxes_number x {x,y,w,h} float, callee-owned */
    float *input;             /* borrowed CHW fp32 [A*(5+C), H, W], set before run (main.c:313) */
    float *output;            /* callee-owned, output_number floats */
    float *probs_buf;         /* callee-owned, boxes_number*(classes+1) floats */
    float **probs;            /* callee-owned row pointers into probs_buf */
} region_layer_t;

typedef void (*callback_draw_box)(uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2,
                                  uint32_t class_id, float prob);

/* region_layer.h:44-45 / region_layer.c:19-66.  0 ok; -1..-4 = which host buffer
 * failed to allocate (as the reference); -5 = device state could not be created. */
int region_layer_init(region_layer_t *rl, int width, int height, int channels, int origin_width,
                      int origin_height);
/* region_layer.h:46 / region_layer.c:68-73 */
void region_layer_deinit(region_layer_t *rl);
/* region_layer.h:47 / region_layer.c:378-383.  obj_info may be NULL (main.c:314);
 * like the reference (call commented out at region_layer.c:382) it is not written. */
void region_layer_run(region_layer_t *rl, obj_info_t *obj_info);
/* region_layer.h:48 / region_layer.c:385-404.  Negative float -> uint32 is UB in the
 * reference; here it is defined as (uint32_t)(int64_t)x, the x86-64 behaviour. */
void region_layer_draw_boxes(region_layer_t *rl, callback_draw_box callback);

/* ------------------------------------------------------------------------- */
/* (2) batched engine                                                         */
/* ------------------------------------------------------------------------- */
#define YK_OK 0
#define YK_ERR_ARG (-10)
#define YK_ERR_HIP (-11)
#define YK_ERR_UNSUPPORTED (-12)
#define YK_ERR_NO_DEVICE (-13)
#define YK_ERR_NOMEM (-14)

/* op codes / activation codes / field indices of one op row (int32 x YK_OP_FIELDS);
 * produced by k210_yolo_framework_amd/netspec.py:compile_plan */
enum { YK_OP_CONV = 1, YK_OP_DWCONV = 2, YK_OP_MAXPOOL = 3, YK_OP_UPSAMPLE = 4, YK_OP_CONCAT = 5, YK_OP_ADD = 6 };
enum { YK_ACT_NONE = 0, YK_ACT_RELU = 1, YK_ACT_RELU6 = 2, YK_ACT_LEAKY = 3 };
enum {
    YK_F_TYPE = 0, YK_F_IN0, YK_F_IN1, YK_F_OUT, YK_F_CIN, YK_F_COUT, YK_F_K, YK_F_STRIDE, YK_F_PAD_T,
    YK_F_PAD_L, YK_F_ACT, YK_F_ALPHA /* float bits */, YK_F_W_OFF, YK_F_SCALE_OFF, YK_F_BIAS_OFF, YK_F_FLAGS,
    YK_F_IN_H, YK_F_IN_W, YK_F_OUT_H, YK_F_OUT_W
};
#define YK_OP_FIELDS 24
#define YK_FLAG_NET_OUTPUT 1
#define YK_MAX_LAYERS 4
#define YK_MAX_ANCHORS 8

typedef struct yk_plan yk_plan_t; /* opaque */

/* Last error text of the calling thread ("" if none). */
const char *yk_last_error(void);
/* Number of visible HIP devices (0 when there is none / no driver). */
int yk_device_count(void);

/* kpu_load_kmodel analogue.  ops: [n_ops][YK_OP_FIELDS] int32; tensors: [n_tensors][4]
 * int32 (h, w, c, is_input); blob: fp32 weights/scale/bias addressed by the op rows.
 * The plan owns every device buffer (weights, activation arena sized for max_batch).
 * Arithmetic: YK_PRECISION_F16X2 below - the mode whose outputs stay within 1e-3 of the fp32 Keras path with identical
 * class / box indices (what `yolo_model.predict`, keras_inference.py:88, returns); fp32 network outputs. */
int yk_plan_create(yk_plan_t **out, const int32_t *ops, int n_ops, const int32_t *tensors, int n_tensors,
                   const float *blob, size_t blob_len, const int32_t *outputs, int n_outputs, int max_batch,
                   int device);
/* The same with a choice of arithmetic (there is no reference counterpart: Keras computes in fp32 on the CPU):
 *   YK_PRECISION_F16X2  (= yk_plan_create) activations in HBM as fp16 PAIRS, x * 2^-e = hi + lo (22 significant bits, the bytes of
 *                       fp32, per-image storage exponent e), weights split the same way once; three fp16 MFMAs per product
 *                       (w_lo*x_hi + w_hi*x_lo + w_hi*x_hi), fp32 accumulate: fp32-class results (the "within 1e-3, indices
 *                       exact" clause of the north star; measured 2e-6 of max|logit|).  Per-image operand scaling: results
 *                       never depend on the batch mates.
 *   YK_PRECISION_F16    fp16 activations in HBM, one fp16 MFMA per product, fp32 accumulate: about twice as fast, and outside
 *                       the tolerance (scores drift ~2e-4 mean / 2e-3..5e-3 max from the fp32 Keras path). */
#define YK_PRECISION_F16 0
#define YK_PRECISION_F16X2 1
/* OR-ed into `precision` (f16x2 only; ignored by f16): how the late backbone and the detection heads are launched.
 *   YK_SCHEDULE_THROUGHPUT (default)  one launch per layer: kernels of several batches in flight on several streams overlap best
 *                                      (bench.py `value`: four batches in flight);
 *   YK_SCHEDULE_LATENCY               two launches of per-image workgroup clusters (csrc/yk_xpersist.h, csrc/yk_xheads.h): every CU
 *                                      is held for their duration, the time of ONE batch is shortest (669 -> 542 us of kernels at 32 images).
 * Same arithmetic; results differ at the 2^-22 rounding level (summation order).  YK_PERSIST / YK_HEADS = 0|1 in the environment override. */
#define YK_SCHEDULE_THROUGHPUT 0x000
#define YK_SCHEDULE_LATENCY 0x100
#define YK_SCHEDULE_MASK 0xf00
int yk_plan_create_ex(yk_plan_t **out, const int32_t *ops, int n_ops, const int32_t *tensors, int n_tensors,
                      const float *blob, size_t blob_len, const int32_t *outputs, int n_outputs, int max_batch,
                      int device, int precision);
void yk_plan_destroy(yk_plan_t *p);

/* kpu_run_kmodel analogue, asynchronous on `stream`.
 * yk_run_u8: d_frames = device uint8 [batch][H][W][3]; fuses Helper._process_img's
 *            `img / np.max(img)` (tools/utils.py:405) as a per-image max reduction.
 * yk_run_f32: d_input = device float [batch][H][W][3], already normalised (what
 *            yolo_model.predict receives, keras_inference.py:88). */
int yk_run_u8(yk_plan_t *p, const uint8_t *d_frames, int batch, void *stream);
int yk_run_f32(yk_plan_t *p, const float *d_input, int batch, void *stream);

/* kpu_get_output analogue: borrowed device pointer to network output idx,
 * NHWC fp32 [max_batch][h][w][A*(5+C)] (rows beyond the last run's batch are stale). */
int yk_get_output(yk_plan_t *p, int idx, float **d_ptr, size_t *bytes, int *h, int *w, int *c);

/* Synchronises the device and returns an error if an earlier asynchronous run of this plan failed ON the device (the f16x2 mode's
 * persistent stage bounds every inter-workgroup wait and gives up rather than hang: then this call says so).  YK_OK otherwise. */
int yk_plan_check(yk_plan_t *p);
/* The same sticky word without waiting for the device (a read of mapped host memory): *error_out != 0 means a run that has already
 * FINISHED failed on the device; the caller has synchronised with the runs it asks about (event / stream).  clear != 0 resets it. */
int yk_plan_peek_error(yk_plan_t *plan, int clear, unsigned *error_out);
int yk_plan_debug_set_error(yk_plan_t *plan, unsigned value); /* test hook: what a failing cluster launch stores */

/* Debug/parity access to any intermediate activation (fp16, channel pitch padded
 * to a multiple of 8): copies tensor `tid` of the last run to host as fp32 NHWC. */
int yk_debug_read_tensor(yk_plan_t *p, int tid, int batch, float *h_dst, size_t dst_elems);
/* Number of kernel launches one yk_run_* issues (after fusion). */
int yk_plan_launch_count(const yk_plan_t *p);
/* Name/shape of the i-th launch for bench/roofline bookkeeping. */
int yk_plan_launch_info(const yk_plan_t *p, int i, char *name, size_t name_len, double *flops_per_image,
                        double *bytes_per_image);

/* Per-launch timing: replays the plan `iters` times with HIP events recorded on `stream` around every
 * launch; ms_out[i] (i < yk_plan_launch_count) = median duration of launch i over the replays, in milliseconds. */
int yk_plan_profile(yk_plan_t *p, const uint8_t *d_frames, int batch, int iters, void *stream, float *ms_out);

/* ---- decode ---------------------------------------------------------------- */
typedef struct {
    int32_t n_layers;                       /* output scales (2 or 3) */
    int32_t anchor_num;                     /* A */
    int32_t class_num;                      /* C */
    int32_t in_h, in_w;                     /* network input size (224, 320) */
    int32_t out_h[YK_MAX_LAYERS];           /* Helper.out_hw */
    int32_t out_w[YK_MAX_LAYERS];
    float anchors[YK_MAX_LAYERS][YK_MAX_ANCHORS][2]; /* Helper.anchors (w,h), image-relative */
} yk_decode_cfg_t;

/* Python-mode decode + per-class NMS for a batch (keras_inference.py:94-135).
 * d_pred[l]: device fp32 [batch][out_h][out_w][A][5+C] (== the engine's outputs).
 * d_image_hw: device float [batch][2] original image (h,w) per image, or NULL = in_hw.
 * d_dets:   device float [batch][C*max_out][6] rows (top,left,bottom,right,score,class),
 *           class-major ascending, score-descending within a class (the reference's order).
 * d_counts: device int32 [batch].
 * Semantics restated from tensorflow 1.14 tf.image.non_max_suppression: score >= obj_thresh
 * kept, greedy, suppress iff IoU > iou_thresh, at most max_out (30) per class; equal scores
 * are ordered by ascending box index (TF leaves it unspecified). */
int yk_decode_py(const yk_decode_cfg_t *cfg, const float *const *d_pred, int batch, const float *d_image_hw,
                 float obj_thresh, float iou_thresh, int max_out, float *d_dets, int32_t *d_counts, void *stream);
/* The same, also returning which box each row is: d_box_index [batch][C*max_out] int32 (may be NULL), the row's index into the
 * reference's flattened (layer, h, w, anchor) box list (keras_inference.py:107-114: `tf.reshape(..., (-1, 4))` per layer, concatenated) -
 * what `tf.boolean_mask` + `tf.image.non_max_suppression` + `tf.gather` select (:116-131). */
int yk_decode_py_ex(const yk_decode_cfg_t *cfg, const float *const *d_pred, int batch, const float *d_image_hw,
                    float obj_thresh, float iou_thresh, int max_out, float *d_dets, int32_t *d_counts, int32_t *d_box_index,
                    void *stream);

/* The same selection, concatenated ACROSS the batch as well (keras_inference.py:133-135 concatenates per image): rows of image b
 * are rows[offsets[b] .. offsets[b+1]) (6 floats each, same order as d_dets), offsets[batch] = total.  `rows` (capacity
 * batch*C*max_out rows), `offsets` [batch+1] and `rows_index` (may be NULL; the box index of every row) may be device memory or
 * pinned, device-mapped HOST memory (hipHostMalloc): then the detections reach the host at their live size, with no device-to-host
 * copy whose size would have to be known first.  d_dets / d_counts (device, may be NULL) additionally receive the padded form. */
int yk_decode_py_packed(const yk_decode_cfg_t *cfg, const float *const *d_pred, int batch, const float *d_image_hw,
                        float obj_thresh, float iou_thresh, int max_out, float *rows, int32_t *offsets, int32_t *rows_index,
                        float *d_dets, int32_t *d_counts, void *stream);

/* ---- a step as one hipGraph (the kpu_run_kmodel(..., dma, ai_done_cb) shape of main.c:303-311: submit once, get called back) ----
 * Everything issued on `stream` (a created stream, not NULL) between yk_graph_begin and yk_graph_end - yk_run_*, yk_decode_py*,
 * yk_letterbox_u8, yk_memcpy_async - is recorded, not executed; yk_graph_launch replays the recording with ONE host call.
 * Rules: run the step once eagerly on the same stream first (scratch is allocated on first use, per stream); the replay uses the
 * pointers, batch and thresholds of the capture; keep the named buffers alive; one replay of a plan at a time.  The capture is
 * thread-local (other host threads may use HIP meanwhile). */
typedef struct yk_graph yk_graph_t; /* opaque */
int yk_graph_begin(void *stream);
int yk_graph_end(void *stream, yk_graph_t **out);
int yk_graph_launch(yk_graph_t *g, void *stream);
int yk_graph_node_count(const yk_graph_t *g);          /* nodes of the recording */
int yk_graph_kernel_node_count(const yk_graph_t *g);   /* ... of which kernels (the rest: copies / fills) */
void yk_graph_destroy(yk_graph_t *g);
/* hipMemcpyAsync(hipMemcpyDefault) on `stream`: the host<->device legs of a captured step (pinned host memory). */
int yk_memcpy_async(void *dst, const void *src, size_t bytes, void *stream);
/* Device address of pinned host memory (hipHostGetDevicePointer): the form of a host buffer a kernel can write. */
int yk_host_device_ptr(void *h_ptr, void **d_ptr);
/* A HIP stream created by the library itself (hipStreamCreateWithPriority, non-blocking; priority 0 = default, negative = higher):
 * the ROCm runtime binds a stream to one of its GPU_MAX_HW_QUEUES hardware queues in creation order, so a pipeline that creates its
 * own streams back to back gets distinct queues whatever pooled streams the host framework handed out before.
 * yk_stream_query_priority returns the priority the runtime reports for any stream handle. */
int yk_stream_create(void **stream_out, int priority);
int yk_stream_destroy(void *stream);
int yk_stream_query_priority(void *stream, int *priority_out);
/* The library keeps per-stream scratch buffers (decode / NMS work space) that grow on demand.  A captured step holds their addresses:
 * yk_scratch_generation(stream) counts how often a buffer of `stream` (on the current device) has MOVED; the holder of captured graphs
 * compares it before a replay and re-captures when it changed (engine.Pipeline does). */
unsigned long long yk_scratch_generation(void *stream);

/* C-mode (region_layer.c:121-283) for a batch of layer outputs without host round trips.
 * d_input element (b, anchor n, entry e, row y, col x) is at
 *   b*stride_b + n*stride_n + e*stride_e + y*stride_y + x*stride_x   (floats)
 * so both the K210 CHW layout and the engine's NHWC outputs can be consumed.
 * d_boxes: [batch][A*H*W][4] (x,y,w,h), d_probs: [batch][A*H*W][C+1]; box index = n*H*W + y*W + x. */
typedef struct {
    int32_t layer_w, layer_h, anchor_num, classes;
    int32_t net_w, net_h, image_w, image_h;
    float threshold, nms_value;
    float anchor[2 * YK_MAX_ANCHORS];
    int64_t stride_b, stride_n, stride_e, stride_y, stride_x;
} yk_region_cfg_t;
int yk_region_batched(const yk_region_cfg_t *cfg, const float *d_input, int batch, float *d_output /* may be NULL */,
                      float *d_boxes, float *d_probs, void *stream);

/* ---- pre-processing (the step immediately before the path) ------------------------------
 * Helper._process_img letterbox (tools/utils.py:378-399) for a batch of equally sized u8 frames:
 * d_src [batch][src_h][src_w][3] -> d_dst [batch][dst_h][dst_w][3], bilinear, zero fill, truncating cast.
 * (The `img / np.max(img)` that follows, utils.py:405, is fused into yk_run_u8.) */
int yk_letterbox_u8(const uint8_t *d_src, int batch, int src_h, int src_w, uint8_t *d_dst, int dst_h, int dst_w,
                    void *stream);
/* `img / np.max(img)` (tools/utils.py:405) for a batch of u8 frames of per_image bytes each -> fp32 (one correctly rounded quotient
 * per element, numpy's float64 division then the pipeline's float32 cast).  Used by the training input pipeline (N3); the inference
 * path fuses the normalisation into the stem conv (yk_run_u8). */
int yk_normalise_u8(const uint8_t *d_frames, int batch, size_t per_image, float *d_out, void *stream);

/* ---- training step, loss level (tools/utils.py:708-793 create_loss_fn, :662-705 calc_ignore_mask,
 *      tools/custom.py:13-75 Yolo_Precision/Yolo_Recall) for ONE output layer.
 * d_y_true / d_y_pred: device fp32 [batch][out_h][out_w][A][5+C] (labels from Helper.box_to_label / raw outputs).
 * d_loss[6]   = {total, xy, wh, obj, noobj, cls}, each already divided by cfg->batch_size like the reference.
 * d_grad      = dL/dy_pred, same shape as d_y_pred (NULL to skip) — what TF autodiff yields for this graph.
 * d_ignore    = ignore mask [batch][out_h][out_w][A] (NULL to skip).
 * d_counts[3] = running {tp, fp, fn}; this batch's counts are ADDED (Keras metric assign_add); NULL to skip. */
typedef struct {
    int32_t out_h, out_w, anchor_num, class_num;
    float anchors[YK_MAX_ANCHORS][2];       /* Helper.anchors[layer] */
    float obj_thresh, iou_thresh, obj_weight, noobj_weight, wh_weight;
    int32_t batch_size;                     /* Helper.batch_size (the divisor in utils.py:771-787) */
} yk_loss_cfg_t;
int yk_yolo_loss(const yk_loss_cfg_t *cfg, const float *d_y_true, const float *d_y_pred, int batch, float *d_loss,
                 float *d_grad, float *d_ignore, float *d_counts, void *stream);

/* ---- training step, network level (keras_train.py:73-98: model.fit = forward in training mode + TF autodiff +
 *      Adam; the reference gets every one of these ops from TensorFlow 1.14).  All tensors are device fp32, NHWC,
 *      dense (no channel padding); "M" is batch*height*width.  k210_yolo_framework_amd/train.py strings them into
 *      the step.
 *
 * yk_gemm_f32: C[M][N] = alpha * op(A) * op(B) + beta * C, row-major, op = transpose when trans* != 0
 *   (A is [M][K] or, transposed, [K][M]; B is [K][N] or [N][K]).  Conv2D 1x1 forward  Y = X * W^T,
 *   data gradient dX = dY * W, weight gradient dW = dY^T * X (tf Conv2DBackpropInput / ...Filter). */
int yk_gemm_f32(int transA, int transB, int M, int N, int K, float alpha, const float *A, int lda, const float *B, int ldb,
                float beta, float *C, int ldc, void *stream);
/* `count` independent GEMMs of one layout in as few launches as possible (the weight gradients of a whole backward pass): the problems'
 * tiles share one grid and the K slices are sized for the group (results reproducible run to run; equal to yk_gemm_f32's within fp32
 * summation-order differences). */
int yk_gemm_f32_grouped(int count, int transA, int transB, const int *M, const int *N, const int *K, float alpha, const float *const *A,
                        const int *lda, const float *const *B, const int *ldb, float beta, float *const *C, const int *ldc, void *stream);
/* 3x3 Conv2D through GEMM: col [B*Ho*Wo][9*C], k = (ky*3+kx)*C + c; col2im is the adjoint (sums overlaps). */
int yk_im2col3x3_f32(const float *x, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, int pad_t, int pad_l,
                     float *col, void *stream);
int yk_col2im3x3_f32(const float *col, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, int pad_t, int pad_l,
                     float *dx, void *stream);
/* 3x3 Conv2D as an implicit GEMM (no column matrix); weights [Co][9 * Ci] (k = (ky*3+kx)*Ci + c).  Needs Ci % 4 == 0 (Co % 4 == 0 too for the
 * gradients; stride 1 for the data gradient) and 16-byte aligned tensors - YK_ERR_UNSUPPORTED otherwise (the im2col path above covers those).
 * yk_conv3x3_bn_fwd_f32: z = conv(x); with gamma != NULL also BatchNormalization(training) + activation (+ residual) -> y, as yk_gemm_bn_fwd_f32.
 * Forward and weight gradient add in the order of im2col + yk_gemm_f32: bitwise the same result. */
int yk_conv3x3_bn_fwd_f32(const float *x, const float *w, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int stride, int pad_t, int pad_l, int Co,
                          float *z, const float *gamma, const float *beta, float eps, int act, float alpha, float *y, float *save_mean,
                          float *save_invstd, float *moving_mean, float *moving_var, float momentum, const float *res, void *stream);
int yk_conv3x3_bwd_weight_f32(const float *x, const float *dz, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int stride, int pad_t, int pad_l,
                              int Co, float *dw, void *stream);
int yk_conv3x3_bwd_data_f32(const float *dz, const float *w, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int stride, int pad_t, int pad_l, int Co,
                            float *dx, void *stream);
/* DepthwiseConv2D 3x3, weights [9][C] */
int yk_dw3x3_fwd_f32(const float *x, const float *w, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, int pad_t,
                     int pad_l, float *y, void *stream);
int yk_dw3x3_bwd_data_f32(const float *dy, const float *w, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride,
                          int pad_t, int pad_l, float *dx, void *stream);
int yk_dw3x3_bwd_weight_f32(const float *x, const float *dy, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride,
                            int pad_t, int pad_l, float *dw, void *stream);
/* the depthwise weight gradients of `count` layers in two launches; geom: 9 ints per problem (B, Hi, Wi, C, Ho, Wo, stride, pad_t, pad_l);
 * every problem computed exactly as by yk_dw3x3_bwd_weight_f32 (bitwise the same). */
int yk_dw3x3_bwd_weight_grouped_f32(int count, const float *const *x, const float *const *dy, const int *geom, float *const *dw, void *stream);
/* BatchNormalization(training=True) fused with the activation that follows it (act: YK_ACT_*).
 * fwd: batch mean / biased variance over M, y = act(gamma*(z-mean)*invstd + beta); saves mean and invstd and,
 *      when moving_mean/moving_var are given, updates them with `momentum` (Keras: 0.99).
 * bwd: dy is the gradient w.r.t. y; writes dz, dgamma, dbeta. */
int yk_bn_train_fwd_f32(const float *z, long long M, int C, const float *gamma, const float *beta, float eps, int act,
                        float alpha, float *y, float *save_mean, float *save_invstd, float *moving_mean,
                        float *moving_var, float momentum, void *stream);
/* ... with the residual of a following keras Add() folded into the apply pass: y = res + act(BN(z)) (keras_mobilenet_v2.py:483-484) */
int yk_bn_train_fwd_res_f32(const float *z, long long M, int C, const float *gamma, const float *beta, float eps, int act,
                            float alpha, float *y, float *save_mean, float *save_invstd, float *moving_mean,
                            float *moving_var, float momentum, const float *res, void *stream);
/* Conv2D / DepthwiseConv2D + BatchNormalization(training=True) + activation (+ residual) forward in ONE call: z = the convolution
 * (X [M][K] row-major with leading dimension ldx, W [N][K]: 1x1 convs directly, 3x3 through yk_im2col3x3_f32), y as yk_bn_train_fwd_res_f32.
 * The producer of z leaves the partial sums of the batch statistics, so z is not read again for them.  Same arithmetic per element as the
 * separate calls; the statistics are added in another (fixed) order in double. */
int yk_gemm_bn_fwd_f32(int M, int N, int K, const float *X, int ldx, const float *W, int ldw, float *z, const float *gamma, const float *beta,
                       float eps, int act, float alpha, float *y, float *save_mean, float *save_invstd, float *moving_mean,
                       float *moving_var, float momentum, const float *res, void *stream);
int yk_dw3x3_bn_fwd_f32(const float *x, const float *w, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, int pad_t, int pad_l,
                        float *z, const float *gamma, const float *beta, float eps, int act, float alpha, float *y, float *save_mean,
                        float *save_invstd, float *moving_mean, float *moving_var, float momentum, const float *res, void *stream);
int yk_bn_train_bwd_f32(const float *z, const float *dy, long long M, int C, const float *gamma, const float *beta,
                        const float *save_mean, const float *save_invstd, int act, float alpha, float *dz, float *dgamma,
                        float *dbeta, void *stream);
int yk_bias_add_f32(float *y, long long M, int C, const float *bias, void *stream);       /* y[m][c] += bias[c] */
int yk_colsum_f32(const float *x, long long M, int C, float *out, void *stream);          /* out[c] = sum_m x[m][c] */
int yk_upsample2x_bwd_f32(const float *dy, int B, int H, int W, int C, float *dx, void *stream); /* UpSampling2D(2) adjoint */
/* MaxPooling2D(2, stride, 'same'): argmax [B][Ho][Wo][C] u8 = winning tap (first maximum, row-major window) */
int yk_maxpool2_fwd_f32(const float *x, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, float *y, uint8_t *argmax,
                        void *stream);
int yk_maxpool2_bwd_f32(const float *dy, const uint8_t *argmax, int B, int Hi, int Wi, int C, int Ho, int Wo, int stride, float *dx,
                        void *stream);
int yk_dot_f32(long long n, const float *x, const float *y, float alpha, float beta, float *out, void *stream); /* *out = alpha*<x,y> + beta*(*out) */
/* keras.regularizers.l2(weight) of yolonet.py:245-250 over `nseg` segments of one flat parameter buffer in one pass: *out = weight * sum w^2
 * (want_value) and / or grads[j] += 2 * weight * params[j] (want_grad).  d_prefix [nseg + 1] = running sum of the segment lengths (device,
 * int64), d_offset [nseg] = start of every segment in the flat buffer, total = d_prefix[nseg]. */
int yk_l2_segments_f32(const float *params, float *grads, const long long *d_prefix, const long long *d_offset, int nseg, long long total,
                       float weight, int want_value, int want_grad, float *out, void *stream);
int yk_axpy_f32(long long n, float a, const float *x, float *y, void *stream);            /* y += a*x */
/* keras.optimizers.Adam as keras_train.py:74-76 configures it (lr, decay; beta 0.9/0.999, eps 1e-7):
 * lr_t = lr/(1+decay*iterations) * sqrt(1-b2^t)/(1-b1^t), t = iterations+1; p -= lr_t*m/(sqrt(v)+eps).
 * g is multiplied by grad_scale first (1/world_size after a sum all-reduce). */
int yk_adam_f32(long long n, float *p, const float *g, float *m, float *v, float lr, float decay, long long iterations,
                float beta1, float beta2, float eps, float grad_scale, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_HIP_H_ */
