/*
 * ty1.h -- C-ABI of libty1.so: the MI355X (gfx950) tiny-YOLOv1 hot path.
 *
 * The reference (AlucardNosferatu/YOLO) is pure Python on TensorFlow/Keras and has no
 * FFI/plugin layer of its own; each entry point below therefore cites the
 * reference Python lines whose arithmetic it replaces.  A reference maintainer binds these
 * with ctypes; yolo_amd/_lib.py is that binding.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; ty1_last_error() returns the
 *    message of the last failing call on the calling thread.  Nothing throws.
 *  - all tensor pointers are DEVICE pointers (hipMalloc / torch ROCm memory) unless the
 *    parameter is documented "host".  The library never allocates or frees caller memory.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *    asynchronous on that stream; the caller synchronises.
 *  - activations are NHWC float32; conv weights are OHWI ([Cout][3][3][Cin], i.e. the Keras
 *    HWIO kernel transposed once at load time); the Dense weight is stored transposed,
 *    [units][ldk] with ldk >= K (row = output unit).
 */
#ifndef TY1_H
#define TY1_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *ty1_stream_t;

/* Library ABI version (bumped on any signature change). */
int ty1_version(void);
/* Message of the last error on this thread ("" if none). */
const char *ty1_last_error(void);

/*
 * HOST helper.  Fold an inference-mode BatchNormalization into per-channel scale/shift:
 *   scale = gamma / sqrt(var + eps), shift = beta - mean * scale      (eps = 1e-3, Keras default)
 * Replaces: BatchNormalization(...) layers, /root/reference/models/model_tiny_yolov1.py:45,51,...,86
 * All pointers are HOST pointers, C floats each.
 */
int ty1_fold_bn(const float *gamma, const float *beta, const float *mean, const float *var,
                float eps, int C, float *scale, float *shift);

/*
 * y = [maxpool2x2/2]( leaky_relu( conv3x3_same(x, w) * scale + shift, alpha ) )
 * Replaces one Conv2D(3x3,'same',use_bias=False) + BatchNormalization + LeakyReLU(0.1)
 * [+ MaxPooling2D((2,2),strides=(2,2))] block: /root/reference/models/model_tiny_yolov1.py:43-87.
 *   x (N,H,W,Cin) f32 NHWC; w (Cout,3,3,Cin) f32 OHWI; scale/shift (Cout);
 *   y (N,H/2,W/2,Cout) if pool else (N,H,W,Cout).  pool requires even H and W.
 *   Cin must be 3 or a multiple of 16; Cout a multiple of 16.
 */
int ty1_conv3x3_bn_leaky_f32(const float *x, const float *w, const float *scale, const float *shift,
                             float *y, int N, int H, int W, int Cin, int Cout, int pool, float alpha,
                             ty1_stream_t stream);

/*
 * Same block with a caller-provided scratch workspace (device memory, 16-byte aligned).  With it, the
 * last partial round of output tiles is split along K into short blocks that fill every CU, and a
 * fix-up kernel sums the partial tiles and applies the epilogue (see csrc/conv_gemm.hip); results are
 * the same function of the inputs up to the f32 summation order of those tiles.  A NULL or too small
 * workspace selects the plain single-grid path.  ty1_conv3x3_workspace_bytes reports the size the
 * given problem wants (0 when it would not split).  No counterpart in the reference (Keras owns its
 * scratch memory).
 */
int ty1_conv3x3_workspace_bytes(int N, int H, int W, int Cin, int Cout, int pool, size_t *bytes);
int ty1_conv3x3_bn_leaky_ws_f32(const float *x, const float *w, const float *scale, const float *shift,
                                float *y, int N, int H, int W, int Cin, int Cout, int pool, float alpha,
                                void *workspace, size_t workspace_bytes, ty1_stream_t stream);

/*
 * The same block for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  No counterpart in the reference.  Same kernels and same results as
 * ty1_conv3x3_bn_leaky_ws_f32 given a workspace of ty1_conv3x3_workspace_bytes(), but the scratch for the K-split tail
 * is a 16 MiB array inside the library's code object (one copy per GPU, held until the process ends): the call makes
 * no hipMalloc / hipFree and needs no workspace argument.  A problem whose tail wants more than
 * ty1_static_scratch_bytes() (16 MiB; batch 64 of every layer of the network fits) is rejected with an error - use the
 * _ws_ entry point for it.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams,
 * two threads) race; calls on one stream are ordered.  Nothing in the array has to be zero: every byte that is read was
 * written by the same call.
 */
size_t ty1_static_scratch_bytes(void);
/*
 * Optional set-up of the entry point below on the CURRENT device: resolves the device address of the static scratch array
 * (hipGetSymbolAddress) so that the first compute call does not have to.  Call it once per device outside any HIP graph
 * capture; without it the first ty1_conv3x3_bn_leaky_static_f32 call on a device does the same work itself (also not
 * capturable: make one eager call per device and tile configuration before capturing, as with any HIP library whose
 * first launch loads its code object).  The 16 MiB array is part of libty1's code object: every process that loads the
 * library pays it on every GPU it touches, whether or not it uses this entry point.
 */
int ty1_static_scratch_init(void);
int ty1_conv3x3_bn_leaky_static_f32(const float *x, const float *w, const float *scale, const float *shift,
                                    float *y, int N, int H, int W, int Cin, int Cout, int pool, float alpha,
                                    ty1_stream_t stream);

/*
 * The same block for the bf16 network (BASELINE config "batch=1024 bf16 inference", 128 images per GPU):
 * activations and weights in bf16, fp32 accumulation on v_mfma_f32_32x32x16_bf16, folded BN + LeakyReLU + pool
 * in fp32, output rounded to bf16 (nearest even).  Same reference lines as above
 * (/root/reference/models/model_tiny_yolov1.py:43-87); the reference itself computes in fp32, parity for this path
 * is a stated tolerance (tests/test_gpu_bf16.py).
 *   Cin == 3:  x (N,H,W,3) and w (Cout,3,3,3) are passed in fp32 - the image enters the network in fp32 - and are
 *              rounded to bf16 on the fly (one v_mfma_f32_16x16x32_bf16 per 16 pixels x 16 channels); y is bf16.
 *              Needs W % 4 == 0, Cout <= 64.
 *   Cin == 16: x, w, y bf16; Cout 32 or 64.
 *   Cin % 32 == 0: x, w bf16; Cout % 64 == 0; y bf16, or fp32 when out_f32 != 0 (pool == 0 only).
 *   alpha >= 0.  workspace as for ty1_conv3x3_bn_leaky_ws_f32 (size from ty1_conv3x3_workspace_bytes_bf16; may be NULL).
 * ty1_cast_f32_to_bf16 converts n floats to bf16 (round to nearest even), e.g. the weights at model load.
 */
int ty1_conv3x3_workspace_bytes_bf16(int N, int H, int W, int Cin, int Cout, int pool, size_t *bytes);
int ty1_conv3x3_bn_leaky_bf16(const void *x, const void *w, const float *scale, const float *shift, void *y,
                              int N, int H, int W, int Cin, int Cout, int pool, float alpha, int out_f32,
                              void *workspace, size_t workspace_bytes, ty1_stream_t stream);
int ty1_cast_f32_to_bf16(const float *x, void *y, long long n, ty1_stream_t stream);

/*
 * Preprocessing of one decoded image on the GPU: uint8 (H0,W0,3) HWC -> cv2.resize INTER_LINEAR to (out_h,out_w)
 * -> optional channel swap (cv2.cvtColor BGR2RGB when the decoder delivered BGR) -> / 255 -> fp32 (out_h,out_w,3),
 * i.e. one image slot of the network's NHWC input batch.
 * Replaces /root/reference/utils.py:11-15 (cvtColor, resize, image / 255.); file decoding stays on the host.
 * Parity unpinned: OpenCV is absent and the reference holds no resized image; the kernel restates OpenCV's generic
 * 8-bit INTER_LINEAR arithmetic (float32 coordinates, 11-bit coefficients, truncating two-pass form, 2x2 area average
 * for an exact 2x downscale) and is bit-identical to the checker's restatement oracle/resize_ref.py.
 */
int ty1_preprocess_u8(const unsigned char *img, int H0, int W0, int swap_rb, float *out, int out_h, int out_w,
                      ty1_stream_t stream);
/*
 * The same for a batch of n equal-sized frames in one launch: imgs (n,H0,W0,3) uint8 -> out (n,out_h,out_w,3) fp32, frame by
 * frame the arithmetic (and the bits) of ty1_preprocess_u8.  Replaces utils.py:11-15 applied over the batch loop of
 * /root/reference/data_sequence.py:94-99 (for dataset in batch_datasets: ... X = np.array(images)) and the per-frame calls of
 * a camera loop, where every frame has the camera's size.
 */
int ty1_preprocess_batch_u8(const unsigned char *imgs, int n, int H0, int W0, int swap_rb, float *out, int out_h, int out_w,
                            ty1_stream_t stream);

/*
 * Split-K Dense: part[s][n][u] = sum_{k in slice s} x[n][k] * wt[u][k]      (no bias here)
 * Replaces the matmul of Flatten + Dense(1470,'linear',name='connected_0'):
 * /root/reference/models/model_tiny_yolov1.py:89,96.  The bias add and the split-K reduction
 * are fused into ty1_head_activate_f32.
 *   x (N,K) f32 (HWC-flattened features); wt (units, ldk) f32, row u = weights of output unit u;
 *   part (splits, N, ldz) f32 with ldz >= units.  K must be a multiple of 32, splits <= K/32
 *   (slice s reduces the 32-wide chunks [K/32*s/splits, K/32*(s+1)/splits)).
 */
int ty1_dense_splitk_f32(const float *x, const float *wt, float *part, int N, int K, int units,
                         int ldk, int ldz, int splits, ty1_stream_t stream);
/* The same for the bf16 network: x and wt hold bf16, part stays fp32; K a multiple of 64, splits <= K/64. */
int ty1_dense_splitk_bf16(const void *x, const void *wt, float *part, int N, int K, int units,
                          int ldk, int ldz, int splits, ty1_stream_t stream);

/*
 * z = sum_s part[s] + bias ; y = concat(softmax_C(z_cls), sigmoid(z_conf), sigmoid(z_box))
 * Replaces the Dense bias add and ReshapeYOLO.call: /root/reference/models/model_tiny_yolov1.py:14-34
 * (C=1 twin: /root/reference/faces/trainFace.py:38-58).
 *   part (splits,N,ldz) planar [S*S*C | S*S*B | S*S*4B]; bias (units) or NULL;
 *   y (N,S,S,C+5B) f32 [cls C | conf B | box 4B].
 */
int ty1_head_activate_f32(const float *part, const float *bias, float *y, int N, int S, int B, int C,
                          int ldz, int splits, ty1_stream_t stream);

/*
 * Decode + adaptive score threshold + class-agnostic greedy NMS ("yolo_eval"), per image.
 * Replaces /root/reference/tiny_yolov1.py:82-140 with YOLO_head :36-55, iou :58-72 and
 * utils.X_Y_W_H_To_Min_Max /root/reference/utils.py:19-23 (C=1 twin:
 * /root/reference/faces/testFace.py:51-110).  Bit-exact with the reference: scores in
 * float32, geometry in float64, ties kept per round, last tied slot in raster order
 * suppresses, IoU > iou_thr strict, NaN never suppresses.
 *   pred  (N,S,S,C+5B) f32;            requires S*S <= 64 and 1 <= B <= 4
 *   boxes (N,S*S*B,4) f64  x0,y0,x1,y1 pixels, zero unless kept      (:139-140)
 *   score (N,S*S*B) f32    confidence after the first filter          (:112)
 *   cls   (N,S*S*B) i32    argmax class, zero unless it passed the filter (:107)
 *   keep  (N,S*S*B) u8     nms_mask
 */
int ty1_decode_nms(const float *pred, int N, int S, int B, int C, int img_size, double iou_thr,
                   double *boxes, float *score, int32_t *cls, uint8_t *keep, ty1_stream_t stream);

/*
 * The inference tail behind the Dense GEMM in one launch: ty1_head_activate_f32 followed by ty1_decode_nms, same
 * arithmetic and same results bit for bit (one workgroup per image reduces and activates the logits into LDS, its first
 * wavefront decodes them).  Replaces /root/reference/models/model_tiny_yolov1.py:14-34 and
 * /root/reference/tiny_yolov1.py:82-140 together.  y (N,S,S,C+5B) may be NULL: the head tensor then never reaches HBM.
 */
int ty1_head_decode_nms_f32(const float *part, const float *bias, float *y, int N, int S, int B, int C, int ldz,
                            int splits, int img_size, double iou_thr, double *boxes, float *score, int32_t *cls,
                            uint8_t *keep, ty1_stream_t stream);

/*
 * Wire format after the path: fixed-size records for the multi-GPU detection gather.
 * The reference has no counterpart beyond drawing the kept boxes (/root/reference/tiny_yolov1.py:142-160);
 * the record keeps exactly what that loop reads: corners, confidence, class.
 *   rec (slots, 6) f64 = x0, y0, x1, y1, score, class (-1.0 where keep == 0); slots = N*S*S*B
 */
int ty1_pack_detections(const double *boxes, const float *score, const int32_t *cls, const uint8_t *keep,
                        double *rec, long long slots, ty1_stream_t stream);

/*
 * NumPy twins used by the reference's inference script (geometry in float64):
 *   YOLO_head: feats (N,S,S,B,4) f32 -> xy (N,S,S,B,2) f64, wh (N,S,S,B,2) f32
 *   iou:       n box pairs, mins/maxes (n,2) f64 -> (n) f64
 * Replace /root/reference/tiny_yolov1.py:36-55 and :58-72 (bit-exact with NumPy).
 */
int ty1_yolo_head_np_f64(const float *feats, double *xy, float *wh, int N, int S, int B, int img_size,
                         ty1_stream_t stream);
int ty1_iou_f64(const double *pred_mins, const double *pred_maxes, const double *true_mins,
                const double *true_maxes, double *out, int64_t n, ty1_stream_t stream);

/*
 * box_xy, box_wh = yolo_head(feats, img_size)   (float32 throughout)
 * Replaces /root/reference/yolo/yolo.py:23-45.   feats (N,S,S,B,4) -> xy (N,S,S,B,2), wh (N,S,S,B,2).
 */
int ty1_yolo_head_f32(const float *feats, float *xy, float *wh, int N, int S, int B, int img_size,
                      ty1_stream_t stream);

/*
 * Elementwise IoU of n box pairs (mins/maxes are (n,2) f32).  Replaces /root/reference/yolo/yolo.py:6-20
 * for equal-shaped operands.
 */
int ty1_iou_f32(const float *pred_mins, const float *pred_maxes, const float *true_mins,
                const float *true_maxes, float *out, int64_t n, ty1_stream_t stream);

/*
 * YOLOv1 loss forward + closed-form backward.
 * Replaces /root/reference/yolo/yolo.py:48-107 and the autodiff Keras derives from it
 * (C=1 twin /root/reference/faces/trainFace.py:66-126, img_size 224).
 *   y_true (N,S,S,C+5) f32 [one-hot C | x,y,w,h | response]; y_pred (N,S,S,C+5B) f32
 *   loss_per_image (N) f32; loss (1) f32 = batch SUM (not mean); grad (N,S,S,C+5B) f32 = dL/dy_pred
 *   (grad may be NULL for forward only).  Requires S*S <= 64, 1 <= B <= 4.
 */
int ty1_yolo_loss_fwd_bwd(const float *y_true, const float *y_pred, int N, int S, int B, int C,
                          int img_size, float *loss_per_image, float *loss, float *grad,
                          ty1_stream_t stream);

/*
 * Backward through ReshapeYOLO.call (softmax / sigmoid Jacobians), output in the planar Dense layout:
 *   dz (N, ldz) f32 from y (N,S,S,C+5B) and grad_y = dL/dy.  Columns >= units are zeroed.
 * Replaces the autodiff of /root/reference/models/model_tiny_yolov1.py:14-34.
 */
int ty1_head_activate_bwd_f32(const float *y, const float *grad_y, float *dz, int N, int S, int B, int C,
                              int ldz, ty1_stream_t stream);

/*
 * The training tail behind the Dense GEMM in one launch: ty1_head_activate_f32, the per-image part of
 * ty1_yolo_loss_fwd_bwd and ty1_head_activate_bwd_f32 with the head tensor and its gradient kept in LDS (one workgroup per
 * image), then the fixed-order batch sum of the loss.  Same arithmetic and the same bits as the three calls.
 * Replaces /root/reference/models/model_tiny_yolov1.py:14-34 (forward and autodiff) and /root/reference/yolo/yolo.py:48-107
 * as /root/reference/train.py:59-60,94-104 runs them inside fit().
 *   part (splits,N,ldz_part) Dense split-K partials; bias (units) or NULL; y_true (N,S,S,C+5);
 *   y (N,S,S,C+5B) or NULL (the head tensor then never reaches HBM); dz (N,ldz) planar dL/dz, zero behind the units;
 *   loss_per_image (N); loss (1) batch SUM.
 */
int ty1_train_tail_f32(const float *part, const float *bias, const float *y_true, float *y, float *dz, int N, int S,
                       int B, int C, int ldz_part, int splits, int img_size, int ldz, float *loss_per_image, float *loss,
                       ty1_stream_t stream);

/*
 * Dense weight gradient (the only trainable layer):
 *   dwt[u][k] = sum_n dz[n][u] * x[n][k]   (u < units, k < K),   db[u] = sum_n dz[n][u]
 * Replaces the autodiff of Dense(1470): /root/reference/models/model_tiny_yolov1.py:96.
 *   dz (N,ldz); x (N,K); dwt (units, ldk) same layout as the forward weight; db (units).
 *   K must be a multiple of 64; rows u in [units, ceil32(units)) of dwt are not touched.
 */
int ty1_dense_wgrad_f32(const float *dz, const float *x, float *dwt, float *db, int N, int K, int units,
                        int ldk, int ldz, ty1_stream_t stream);

/*
 * Adam step, Keras semantics (lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)).
 * Replaces optimizer Adam(lr=0.001) of /root/reference/train.py:59 (beta .9/.999, eps 1e-7).
 *   p, g, m, v: n floats each; t = 1-based step count.
 */
int ty1_adam_step_f32(float *p, const float *g, float *m, float *v, int64_t n, float lr, float beta1,
                      float beta2, float eps, int64_t t, ty1_stream_t stream);

/*
 * The first two blocks of the bf16 network in one launch - convolution_0 and convolution_1 with their BatchNormalization,
 * LeakyReLU and MaxPooling2D layers (/root/reference/models/model_tiny_yolov1.py:43-53) - so that the (H/2, W/2, 16) tensor
 * between them stays in LDS instead of crossing HBM twice:
 *   y = pool(leaky(bn1(conv3x3(pool(leaky(bn0(conv3x3(x, w0)))), w1))))
 * x (N,H,W,3) fp32 NHWC and w0 (16,3,3,3) fp32 OHWI are rounded to bf16 inside the kernel, w1 (32,3,3,16) is bf16 OHWI,
 * y (N,H/4,W/4,32) is bf16; accumulation, BN, LeakyReLU and pool are fp32, the intermediate tensor is rounded to bf16 -
 * the rounding points of ty1_conv3x3_bn_leaky_bf16 called twice.  H, W multiples of 4; 0 < alpha <= 1.
 */
int ty1_conv3x3_x2_c3_bf16(const float *x, const float *w0, const float *scale0, const float *shift0, const void *w1,
                           const float *scale1, const float *shift1, void *y, int N, int H, int W, float alpha,
                           ty1_stream_t stream);

/*
 * Room for a collective next to the persistent kernels.  No counterpart in the reference (it has no second device).
 * Every persistent launch of the library (conv blocks in fp32 and bf16) sizes its grid to the workgroups that stay
 * resident on (CU count - k) compute units; k = 0 (the default) fills the chip.  engine.Trainer can set k > 0 while its
 * gradient all-reduce is in flight on a side stream, so that RCCL's workgroups find free CUs instead of queueing behind
 * ~220 us conv kernels.  Only the number of workgroups changes - the work items, the K-split plan and every order of
 * summation are made for the full CU count - so results are bit-identical for every k.  Process-wide, read at launch
 * time on the host; at least 8 CUs are always used.
 */
int ty1_set_reserved_cus(int k);
int ty1_get_reserved_cus(int *k);

/*
 * Per-kernel device timing with HIP events recorded on the launch stream (used by bench.py
 * for the roofline figure).  ty1_prof_enable(1) makes every later launch of this thread
 * bracket itself with events; ty1_prof_query synchronises those events and returns the
 * launch count and summed milliseconds for kernels whose name starts with `prefix`.
 */
int ty1_prof_enable(int on);
/* Restrict the event brackets to launches whose name starts with `prefix` (NULL or "": all launches). */
int ty1_prof_filter(const char *prefix);
/* Bracket only every n-th launch that passes the filter (n <= 1: every one).  Two event records per launch cost the stream a
 * few microseconds; a sample of the launches gives the same average duration at a fraction of that. */
int ty1_prof_stride(int n);
int ty1_prof_reset(void);
int ty1_prof_query(const char *prefix, int *launches, double *total_ms);
/* Writes up to cap bytes of "name launches total_ms\n" lines for every kernel seen. */
int ty1_prof_dump(char *buf, size_t cap);

/*
 * Developer hook (no reference counterpart): when buf != NULL, every later persistent conv launch with at most
 * cap_blocks workgroups writes 8 int64 per workgroup: s_memrealtime (100 MHz) at start and end, cycles spent in the
 * K loops / epilogues / geometry, HW_ID, XCC_ID, item count (tools/stamp_trace.py).  Pass NULL to switch it off.
 */
int ty1_debug_stamp_buffer(long long *buf, long long cap_blocks);

#ifdef __cplusplus
}
#endif
#endif /* TY1_H */
