/*
 * libyolohead -- C ABI of the MI355X (gfx950) YOLOv1 detection head.
 *
 * The reference (KimRass/YOLO) has no FFI or operator interface: its head is a
 * set of Python methods on YOLOv1 (model.py:150-317, 379-394) built from
 * eval.py:14-65.  Each entry point below names the reference methods it
 * replaces.  All pointers are DEVICE pointers to contiguous buffers owned by
 * the caller; the library allocates nothing, keeps no state and never
 * synchronises: every call only enqueues kernels on `stream` (a hipStream_t
 * passed as void*; NULL = the default stream) for the current device.
 *
 * Return value: 0 = ok; < 0 = argument error found on the host before anything
 * was launched (YH_ERR_*); > 0 = the hipError_t of a failed launch.
 * yh_strerror() describes either kind.
 *
 * Tensor layouts (fp32 unless stated; S = grid size, Bx = predictors per cell,
 * C = classes, D = 5*Bx + C):
 *   pred      (batch, D, S, S)    raw logits, channel-major (model.py:140-148):
 *                                 ch 5k..5k+3 = x,y,w,h of predictor k,
 *                                 ch 5k+4 = its confidence, ch 5Bx.. = classes
 *   cell index                    i*S + j for pred[b, :, i, j]; the FIRST
 *                                 spatial axis carries x (model.py:171-180)
 *   gt_xywh   (batch, S*S, 1, 4)  normalised ground truth (data.py:185-194)
 *   gt_cls    (batch, S*S, 1, C)  class probabilities
 *   obj_mask  (batch, S*S, 1, 1)  uint8 / bool
 */
#ifndef YOLO_HEAD_H
#define YOLO_HEAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YH_VERSION 101 /* 0.1.1 */

#define YH_OK 0
#define YH_ERR_NULL (-1)        /* a required pointer is NULL */
#define YH_ERR_SHAPE (-2)       /* batch/S/Bx/C/img_size outside the supported range */
#define YH_ERR_WORKSPACE (-3)   /* workspace missing or smaller than yh_workspace_bytes() */
#define YH_ERR_ALIGN (-4)       /* a pointer is not aligned as documented */
#define YH_ERR_ARG (-5)         /* a scalar argument is invalid (mode, threshold is NaN, ...) */
#define YH_ERR_UNSUPPORTED (-6) /* more candidates per image than the kernels handle */
#define YH_ERR_SCRATCH (-7)     /* yh_loss_fwd_bwd_static: the batch does not fit the static scratch */

/* limits checked on the host */
#define YH_MAX_S 64
#define YH_MAX_BX 16
#define YH_MAX_C 4096
#define YH_MAX_CANDIDATES 2048     /* per image, for the NMS entry points without a workspace */
#define YH_MAX_CANDIDATES_WS 65536 /* per image, for yh_decode_nms_ws / yh_nms_ws (= 64*64*16) */

/* op codes for yh_workspace_bytes */
#define YH_OP_LOSS 0
#define YH_OP_DECODE_NMS 1
#define YH_OP_DECODE_GRID 2
#define YH_OP_NMS 3
#define YH_OP_MAP 4 /* batch = number of predictions */
#define YH_OP_LOSS_PIPELINED 5

/* candidate modes of the NMS entry points */
#define YH_MODE_REFERENCE 0 /* per cell the higher-confidence predictor (model.py:384-390) */
#define YH_MODE_DENSE 1     /* every predictor of every cell */

int yh_version(void);
const char* yh_strerror(int rc);

/* Bytes of device workspace the op needs for this shape (0 if none). */
size_t yh_workspace_bytes(int op, int64_t batch, int S, int Bx, int C);

/*
 * Loss forward + backward in one pass.  Replaces YOLOv1.get_loss and everything
 * it calls (model.py:296-317: _get_responsibility_mask :190-211,
 * get_coordinate_loss :224-240, get_confidence_loss :252-276,
 * get_classification_loss :288-294, get_ltrb :185-188, eval.get_iou
 * eval.py:53-65) and the autograd backward of loss.backward() (train.py:89,93).
 *
 * grad_pred (batch, D, S, S) receives d(total * inv_global_batch)/d(pred).
 * loss_terms[5] receives: coordinate, confidence and class SUMS over this
 * batch (not divided), their total, and total * inv_global_batch (the value
 * get_loss returns when inv_global_batch = 1/batch, model.py:317).
 * Alignment: gt_xywh 16 bytes; pred, grad_pred and gt_cls 4 bytes (the kernel tuned for
 * S=7, Bx=2, C=20 runs when grad_pred and gt_cls are 16-byte aligned, as any
 * allocation is; otherwise the run-time-shape kernel does the work).
 * workspace: yh_workspace_bytes(YH_OP_LOSS, ...) bytes, 16-byte aligned.
 * The result is bitwise reproducible from run to run (no float atomics).
 */
int yh_loss_fwd_bwd(const float* pred, const float* gt_xywh, const float* gt_cls,
                    const uint8_t* obj_mask, float* grad_pred, float* loss_terms,
                    void* workspace, size_t workspace_bytes, int64_t batch, int S, int Bx,
                    int C, int img_size, float lambda_coord, float lambda_noobj,
                    float inv_global_batch, void* stream);

/*
 * yh_loss_fwd_bwd without a workspace argument and without any allocation.  It
 * has no counterpart in the reference; it is for callers that cannot allocate
 * device memory or pass a workspace: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc.  The loss rows (16 bytes per
 * workgroup) go to an array of YH_STATIC_SCRATCH_BYTES that is part of the
 * library image on every GPU the library runs on, for the life of the process;
 * its device address is looked up once per device, on the first call made with
 * that device current (two host-side runtime queries: make that first call outside
 * graph capture; every later call only enqueues kernels and may be captured).
 * Same kernels and same bits as yh_loss_fwd_bwd.
 * Limit: yh_workspace_bytes(YH_OP_LOSS, batch, S, Bx, C) <= YH_STATIC_SCRATCH_BYTES
 * (batch <= 65,536 at S = 7; batch <= 16,384 for S >= 12), else YH_ERR_SCRATCH.
 * All calls on one GPU share that GPU's array: calls that overlap there (two
 * streams, two threads) race on the rows; order them.  Nothing in the array has
 * to start at zero -- a call writes every row it later reads.
 */
#define YH_STATIC_SCRATCH_BYTES 262144
int yh_loss_fwd_bwd_static(const float* pred, const float* gt_xywh, const float* gt_cls,
                           const uint8_t* obj_mask, float* grad_pred, float* loss_terms,
                           int64_t batch, int S, int Bx, int C, int img_size, float lambda_coord,
                           float lambda_noobj, float inv_global_batch, void* stream);

/*
 * The loss and the fused decode + NMS on a CELL-MAJOR head tensor: pred and grad_pred are
 * (batch, S, S, D) -- cell i*S + j of an image holds its D values consecutively, in the
 * channel order of model.py:140-142 -- which is the memory of a channels-last
 * (batch, D, S, S) tensor, for backbones that end in a channels-last convolution instead of
 * the reference's Linear + view (model.py:133-135, 146-148).  Everything else as
 * yh_loss_fwd_bwd / yh_decode_nms (same arithmetic per cell; the run-time-shape kernels do
 * the work for every shape).
 */
int yh_loss_fwd_bwd_nhwc(const float* pred, const float* gt_xywh, const float* gt_cls,
                         const uint8_t* obj_mask, float* grad_pred, float* loss_terms,
                         void* workspace, size_t workspace_bytes, int64_t batch, int S, int Bx,
                         int C, int img_size, float lambda_coord, float lambda_noobj,
                         float inv_global_batch, void* stream);
int yh_decode_nms_nhwc(const float* pred, float* dets, int32_t* keep_count, int32_t* keep_idx,
                       int mode, float conf_thresh, float iou_thresh, int64_t batch, int S, int Bx,
                       int C, int img_size, void* stream);

/*
 * The loss terms of every image by itself (forward only): per_image (batch, 4) receives the
 * coordinate, confidence and class sums of image b and their total -- the per-image values
 * whose sum over the batch, divided by it, is what get_loss returns (model.py:296-317; the
 * reference itself only forms the batch sums).  One wavefront per image and a wavefront
 * reduction when S*S <= 64, a workgroup per image above; fixed summation order, so the values
 * are reproducible from run to run.  The arithmetic per cell is that of yh_loss_fwd_bwd; the
 * order of the additions differs, so the batch totals agree to rounding, not to the bit.
 * Alignment: gt_xywh 16 bytes, the others 4.
 */
int yh_loss_per_image(const float* pred, const float* gt_xywh, const float* gt_cls,
                      const uint8_t* obj_mask, float* per_image, int64_t batch, int S, int Bx, int C,
                      int img_size, float lambda_coord, float lambda_noobj, void* stream);

/*
 * The two halves of yh_loss_fwd_bwd as separate launches, for callers that put
 * the scalar reduction on another stream (nothing downstream of the gradient
 * depends on it) or log the loss on the device.
 *
 * yh_loss_partials: everything yh_loss_fwd_bwd does except the final sum; one
 * row of four floats per workgroup is left in `workspace`.
 * yh_loss_finalize: adds those rows in a fixed order and writes loss_terms[5]
 * as yh_loss_fwd_bwd does (same batch and S as the yh_loss_partials call).  If
 * log_ring is not NULL it also appends the four sums to a device-side ring:
 * row (*log_cursor % log_capacity) of log_ring (log_capacity, 4) receives
 * coord, conf, class, total, then *log_cursor is incremented.  The cursor lives
 * on the device, so a captured graph can be replayed and still log every step.
 * No counterpart in the reference, which prints the terms (model.py:316).
 */
int yh_loss_partials(const float* pred, const float* gt_xywh, const float* gt_cls,
                     const uint8_t* obj_mask, float* grad_pred, void* workspace,
                     size_t workspace_bytes, int64_t batch, int S, int Bx, int C, int img_size,
                     float lambda_coord, float lambda_noobj, float inv_global_batch,
                     void* stream);
int yh_loss_finalize(const void* workspace, size_t workspace_bytes, int64_t batch, int S,
                     float inv_global_batch, float* loss_terms, float* log_ring,
                     int log_capacity, int32_t* log_cursor, void* stream);

/*
 * Pipelined loss step for training loops that only LOG the loss (the reference
 * reads it once per step for its running mean, train.py:84-94): one launch per
 * step.  The call writes grad_pred as yh_loss_fwd_bwd does and leaves its rows in
 * half `phase` (0 or 1) of the workspace; its first workgroup meanwhile adds the
 * rows the PREVIOUS call left in the other half -- bit for bit what
 * yh_loss_finalize computes -- and writes them to loss_terms_prev[5] (may be
 * NULL) and to the log ring (may be NULL).  So every step's sums reach the log,
 * one call late, and no second kernel sits between two steps.
 *
 * Contract: alternate phase 0, 1, 0, ... from call to call on the same workspace
 * pointer and size (a captured graph that is replayed needs an even number of
 * steps).  yh_loss_pipelined_flush(workspace, size, phase of the LAST call, ...)
 * logs the step that is still pending (a no-op if none is; the other half is
 * looked at as well); call it before reading the log to the end, and whenever
 * the alternation is to restart -- after it any phase may come first.
 * workspace: yh_workspace_bytes(YH_OP_LOSS_PIPELINED, ...) bytes, 16-byte aligned,
 * and -- unlike the other workspaces of this library -- its contents matter: clear
 * it once with yh_loss_workspace_init (or any zero fill) before the first call.
 * Calls that share a workspace must not overlap.  No counterpart in the reference.
 */
int yh_loss_workspace_init(void* workspace, size_t workspace_bytes, void* stream);
int yh_loss_fwd_bwd_pipelined(const float* pred, const float* gt_xywh, const float* gt_cls,
                              const uint8_t* obj_mask, float* grad_pred, float* loss_terms_prev,
                              float* log_ring, int log_capacity, int32_t* log_cursor,
                              void* workspace, size_t workspace_bytes, int phase, int64_t batch,
                              int S, int Bx, int C, int img_size, float lambda_coord,
                              float lambda_noobj, float inv_global_batch, void* stream);
int yh_loss_pipelined_flush(void* workspace, size_t workspace_bytes, int phase, float* loss_terms,
                            float* log_ring, int log_capacity, int32_t* log_cursor, void* stream);

/*
 * The last layer's bias and view folded into the head read (model.py:133-135,
 * 146-148: nn.Linear(4096, S*S*D) followed by .view(-1, D, S, S)).  `z`
 * (batch, S*S*D) is that layer's matrix product WITHOUT the bias, read in place as
 * (batch, D, S, S) -- the view is free, no cell-major staging copy exists in this
 * library -- and `bias` (S*S*D floats, the layer's bias) is added as each logit is
 * read: the result equals the unfused call on z + bias bit for bit.
 * yh_loss_fwd_bwd_linear: as yh_loss_fwd_bwd; grad_z is the gradient with respect
 * to z (equal to the one with respect to z + bias).
 * yh_decode_nms_linear: as yh_decode_nms.
 * yh_bias_grad: grad_bias[c] = sum over rows of grad (nrows, ncols) in a fixed
 * order (bitwise reproducible) -- the bias gradient of the layer's backward, with
 * nrows = batch, ncols = S*S*D.  workspace: yh_bias_grad_workspace_bytes bytes.
 */
int yh_loss_fwd_bwd_linear(const float* z, const float* bias, const float* gt_xywh,
                           const float* gt_cls, const uint8_t* obj_mask, float* grad_z,
                           float* loss_terms, void* workspace, size_t workspace_bytes,
                           int64_t batch, int S, int Bx, int C, int img_size, float lambda_coord,
                           float lambda_noobj, float inv_global_batch, void* stream);
int yh_decode_nms_linear(const float* z, const float* bias, float* dets, int32_t* keep_count,
                         int32_t* keep_idx, int mode, float conf_thresh, float iou_thresh,
                         int64_t batch, int S, int Bx, int C, int img_size, void* stream);
size_t yh_bias_grad_workspace_bytes(int64_t nrows, int ncols);
int yh_bias_grad(const float* grad, float* grad_bias, void* workspace, size_t workspace_bytes,
                 int64_t nrows, int ncols, void* stream);

/*
 * grad[i] *= *scale for i < n, in place; does not touch memory when *scale == 1.
 * `scale` is a device scalar: the grad_output that reaches the head's backward
 * (1 for loss.backward(), the GradScaler scale for
 * scaler.scale(loss).backward(), train.py:89).
 */
int yh_scale_grad(float* grad, const float* scale, int64_t n, void* stream);

/*
 * Pairwise box geometry.  Replaces eval.get_area (eval.py:22-32),
 * get_intersection_area (:35-50), get_iou (:53-65), get_smallest_enclosing_area (:68-73)
 * and get_giou (:76-83); the
 * fp32 results are bit-identical to the reference's PyTorch expressions (IEEE
 * division, no fused multiply-add).
 *   ltrb1 (N, 4), ltrb2 (M, 4), both 16-byte aligned; out (N, M) row-major.
 *   kind: YH_BOX_IOU, YH_BOX_GIOU, YH_BOX_INTERSECTION or YH_BOX_ENCLOSING.
 * yh_box_area: out[i] = max(r-l, 0) * max(b-t, 0) for n boxes.
 */
#define YH_BOX_IOU 0
#define YH_BOX_GIOU 1
#define YH_BOX_INTERSECTION 2
#define YH_BOX_ENCLOSING 3
int yh_pairwise_boxes(const float* ltrb1, const float* ltrb2, float* out, int64_t N, int64_t M,
                      int kind, void* stream);
int yh_box_area(const float* ltrb, float* out, int64_t n, void* stream);

/*
 * 11-point mean average precision exactly as eval.get_map / get_ap /
 * get_accum_prec_and_recall compute it (eval.py:86-149), quirks included: IoU is
 * multiplied by 8 before it meets the threshold (eval.py:99); a prediction is a
 * true positive when ANY ground truth of its class reaches the threshold (the
 * matched flags of eval.py:109-110 are never set); precision counts the
 * predictions of every class (eval.py:124); recall divides by ALL ground truths
 * (eval.py:125).  Predictions are ranked by confidence descending, equal
 * confidences by lower index (torch.argsort leaves ties unspecified).
 *   pred (P, 6) rows l,t,r,b,conf,class (the rows yh_decode_nms writes);
 *   gt (G, 5) rows l,t,r,b,class; P, G <= YH_MAP_MAX_ROWS (a whole dataset's
 *   detections in one call, as eval.get_map takes them).  A NaN confidence ranks first
 *   (torch.argsort(descending=True) counts NaN as the largest value, eval.py:121).
 *   ap_out (n_classes) float64, map_out (1) float64 (eval.py:142, :149).
 *   workspace: yh_map_workspace_bytes(P, G) bytes, 256-byte aligned: two buffers of
 *   64-bit sort words, the true-positive flags and the histogram of the library's own
 *   device radix sort.  Like every other call it only enqueues kernels (legal under
 *   graph capture).  A small call (P <= 1024, G <= 2048, n_classes <= 1024) is one
 *   launch of one workgroup that sorts in LDS and does not touch the workspace; same
 *   bits either way.
 */
#define YH_MAP_MAX_ROWS 16777216
size_t yh_map_workspace_bytes(int64_t P, int64_t G);
int yh_map_11point(const float* pred, const float* gt, int64_t P, int64_t G, int n_classes,
                   float iou_thresh, double* ap_out, double* map_out, void* workspace,
                   size_t workspace_bytes, void* stream);

/*
 * Target encoder: ground-truth boxes -> the three loss inputs, on the device.
 * Restates ltrb_to_xywh (data.py:173-183), normalize_xywh (data.py:185-194) and
 * get_object_mask (model.py:213-222); the reference leaves this step unfinished
 * on the host (data.py:196-203).
 *   gt_ltrb  (batch, M, 4) pixel boxes l,t,r,b;  gt_label (batch, M) int32;
 *   gt_count (batch) int32: boxes m >= gt_count[b] are padding and ignored.
 * The cell of a box is [cx // cell_size, cy // cell_size] (first grid axis = x,
 * flat index x_cell*S + y_cell, clamped into the grid); when several boxes fall
 * into one cell the LAST one wins.  Cells without a box get zeros.  A label
 * outside [0, C) leaves the class row all zero.  Outputs: gt_xywh (16-byte
 * aligned), gt_cls, obj_mask as yh_loss_fwd_bwd reads them.  Every value is
 * bit-identical to the PyTorch arithmetic of the reference (no transcendental).
 */
int yh_encode_targets(const float* gt_ltrb, const int32_t* gt_label, const int32_t* gt_count,
                      float* gt_xywh, float* gt_cls, uint8_t* obj_mask, int64_t batch, int M,
                      int S, int C, int img_size, void* stream);

/*
 * Dense grid decode.  Replaces get_ltrb (model.py:150-188), get_confidence
 * (:242-250), get_classification_prob (:278-286), _get_classification_index
 * (:392-394), get_confidence_mask (:379-382) and get_max_confidence_ltrb
 * (:384-390).  Every output pointer may be NULL (that output is skipped).
 *   ltrb      (batch, S*S, Bx, 4)  pixels, one-sided clips as xywh_to_ltrb
 *   conf      (batch, S*S, Bx, 1)
 *   cls_prob  (batch, S*S, 1, C)   softmax
 *   cls_idx   (batch, S*S, 1, 1)   int64, argmax of the class logits, first index on ties
 *   best_k    (batch, S*S, 1, 1)   int64, predictor with the higher confidence, first on ties
 *   best_ltrb (batch, S*S, 1, 4)
 *   conf_mask (batch, S*S, 1, 1)   uint8, max_k conf >= conf_thresh
 */
int yh_decode_grid(const float* pred, float* ltrb, float* conf, float* cls_prob,
                   int64_t* cls_idx, int64_t* best_k, float* best_ltrb, uint8_t* conf_mask,
                   float conf_thresh, int64_t batch, int S, int Bx, int C, int img_size,
                   void* stream);

/*
 * The box helpers of the reference as stand-alone element-wise kernels, for code
 * that calls them one by one (the fused entry points above do not go through
 * them).  Boxes are float4, 16-byte aligned.
 * yh_norm_xywh: get_norm_xywh (model.py:157-165) batched: norm_xywh
 *   (batch, S*S, Bx, 4) = sigmoid of channels 5k .. 5k+3 of cell i*S + j.
 * yh_denormalize_xywh: denormalize_xywh (model.py:167-183) on (batch, S*S, Bx, 4):
 *   x, y times the cell size plus the cell's offset (first grid index carries x),
 *   w, h times the image size.
 * yh_xywh_to_ltrb: xywh_to_ltrb (model.py:150-155) on n boxes, with the
 *   reference's one-sided clips (l, t at 0 from below; r, b at img_size from above).
 * yh_responsibility_mask: _get_responsibility_mask (model.py:190-211) as get_loss
 *   needs it: resp_mask (batch, S*S, Bx) u8 is 1 for the predictor with the highest
 *   IoU against the ground truth of a cell that holds an object (first index on
 *   ties), 0 elsewhere -- the mask yh_loss_fwd_bwd works with internally.
 */
int yh_norm_xywh(const float* pred, float* norm_xywh, int64_t batch, int S, int Bx, int C,
                 void* stream);
int yh_denormalize_xywh(const float* norm_xywh, float* xywh, int64_t batch, int S, int Bx,
                        int img_size, void* stream);
int yh_xywh_to_ltrb(const float* xywh, float* ltrb, int64_t n, int img_size, void* stream);
int yh_responsibility_mask(const float* pred, const float* gt_xywh, const uint8_t* obj_mask,
                           uint8_t* resp_mask, int64_t batch, int S, int Bx, int C, int img_size,
                           void* stream);

/*
 * NMS candidates exactly as yh_decode_nms forms them internally (same device
 * code): flat index n = cell*Bx + k, N = S*S*Bx per image.
 *   cand_ltrb (batch, N, 4), cand_score (batch, N) = confidence,
 *   cand_cls (batch, N) int32 = class of the cell, cand_present (batch, N) uint8
 *   (mode YH_MODE_REFERENCE: only the higher-confidence predictor of a cell is
 *   present; YH_MODE_DENSE: all are).
 */
int yh_decode_candidates(const float* pred, float* cand_ltrb, float* cand_score,
                         int32_t* cand_cls, uint8_t* cand_present, int mode, int64_t batch,
                         int S, int Bx, int C, int img_size, void* stream);

/*
 * Fused decode + confidence threshold + per-class greedy NMS.  The reference
 * stops at the confidence threshold (model.py:379-394, 406-419) and has no
 * NMS; the keep rule is specified here: a candidate is valid when present and
 * score >= conf_thresh; valid candidates are ordered by class ascending, score
 * descending, flat index ascending; walking each class in that order a
 * candidate is kept unless an already kept one of its class has
 * IoU > iou_thresh with it (IoU arithmetic of eval.py:53-65 in fp32).
 *   dets       (batch, N, 6)  kept rows l,t,r,b,score,class (the layout
 *              eval.py:89-91 consumes) in that order, zero rows after them
 *   keep_count (batch) int32
 *   keep_idx   (batch, N) int32 flat index of each kept row, -1 after them; may be NULL
 * N = S*S*Bx <= YH_MAX_CANDIDATES (more: yh_decode_nms_ws).  The candidates of an
 * image live in LDS: up to 36 bytes per candidate plus per-class tables, within the
 * 160 KB of a gfx950 CU (YH_ERR_UNSUPPORTED when a shape would not fit).
 */
int yh_decode_nms(const float* pred, float* dets, int32_t* keep_count, int32_t* keep_idx,
                  int mode, float conf_thresh, float iou_thresh, int64_t batch, int S, int Bx,
                  int C, int img_size, void* stream);

/*
 * yh_decode_nms with the row count of `dets` chosen by the caller: dets is
 * (batch, rows, 6) and keep_idx (batch, rows), rows >= the candidates of the mode
 * (S*S in YH_MODE_REFERENCE -- model.py:384-390 keeps one box per cell, so no image
 * can have more detections -- S*S*Bx in YH_MODE_DENSE), else YH_ERR_ARG.  Same keep
 * rule, order and bits in the rows both calls write; with rows = S*S the reference
 * mode writes half the bytes yh_decode_nms does at Bx = 2.
 */
int yh_decode_nms_rows(const float* pred, float* dets, int32_t* keep_count, int32_t* keep_idx,
                       int mode, float conf_thresh, float iou_thresh, int rows, int64_t batch, int S,
                       int Bx, int C, int img_size, void* stream);

/*
 * One call for both halves of a step on the same head tensor: yh_loss_fwd_bwd and
 * yh_decode_nms_rows, as an evaluation pass needs them together (the validation loss of
 * train.py:100-110 / model.py:296-317 and, for eval.get_map of eval.py:145, the detections
 * of model.py:379-394 from one forward pass).  At S = 7, Bx = 2, C = 20 in YH_MODE_REFERENCE (gt_cls and grad_pred
 * 16-byte aligned) ONE kernel reads the head tensor once for both; every other case
 * enqueues the two launches one after the other.  All outputs -- grad_pred, loss_terms,
 * dets, keep_count, keep_idx -- are bit for bit those of the two separate calls.
 * rows: rows of dets / keep_idx per image (0: S*S*Bx), as in yh_decode_nms_rows.
 * workspace: yh_workspace_bytes(YH_OP_LOSS, ...) bytes, 16-byte aligned.
 *
 * yh_head_step_pipelined: the same with the loss sums logged one call late, exactly as
 * yh_loss_fwd_bwd_pipelined does (its contract on phase, workspace --
 * yh_workspace_bytes(YH_OP_LOSS_PIPELINED, ...), cleared once -- and
 * yh_loss_pipelined_flush apply unchanged; calls of the two pipelined entry points may
 * alternate on one workspace): one launch per step.
 */
int yh_head_step(const float* pred, const float* gt_xywh, const float* gt_cls,
                 const uint8_t* obj_mask, float* grad_pred, float* loss_terms, float* dets,
                 int32_t* keep_count, int32_t* keep_idx, void* workspace, size_t workspace_bytes,
                 int mode, float conf_thresh, float iou_thresh, int rows, int64_t batch, int S, int Bx,
                 int C, int img_size, float lambda_coord, float lambda_noobj, float inv_global_batch,
                 void* stream);
int yh_head_step_pipelined(const float* pred, const float* gt_xywh, const float* gt_cls,
                           const uint8_t* obj_mask, float* grad_pred, float* dets,
                           int32_t* keep_count, int32_t* keep_idx, float* loss_terms_prev,
                           float* log_ring, int log_capacity, int32_t* log_cursor, void* workspace,
                           size_t workspace_bytes, int phase, int mode, float conf_thresh,
                           float iou_thresh, int rows, int64_t batch, int S, int Bx, int C,
                           int img_size, float lambda_coord, float lambda_noobj,
                           float inv_global_batch, void* stream);

/*
 * The same NMS over caller-supplied candidates (N per image, any order; the
 * flat index is the position in the input).  cand_present may be NULL (all
 * present).  Classes must be in [0, 32767].  Any class values give the same
 * result; with N <= 128, an image whose valid candidates all have classes below
 * 64 is settled without ranking its candidates, which is the faster path.
 */
int yh_nms(const float* cand_ltrb, const float* cand_score, const int32_t* cand_cls,
           const uint8_t* cand_present, float* dets, int32_t* keep_count, int32_t* keep_idx,
           float conf_thresh, float iou_thresh, int64_t batch, int N, void* stream);

/*
 * The NMS entry points for ANY shape the loss accepts (S <= 64, Bx <= 16: up to
 * YH_MAX_CANDIDATES_WS = 65,536 candidates per image; the reference's
 * post-processing has no size limit, model.py:379-394).  Up to
 * YH_MAX_CANDIDATES per image they forward to yh_decode_nms /
 * yh_decode_nms_linear (bias != NULL) / yh_nms, which keep an image's candidates in
 * LDS, and the workspace may be NULL.  Above, the candidates go through global
 * memory: boxes and 64-bit sort keys in the workspace, the library's own segmented
 * radix sort (one segment per image, stable, on the class and score bits), then one
 * workgroup per image that walks each class in order against the rows it has already
 * kept.  Same keep rule, row order and bits as yh_decode_nms / yh_nms.
 *   workspace: yh_nms_workspace_bytes(batch, N, own_boxes) bytes, 256-byte aligned;
 *   N = S*S*Bx and own_boxes = 1 for yh_decode_nms_ws, own_boxes = 0 for yh_nms_ws
 *   (0 bytes when N <= YH_MAX_CANDIDATES).  Launch-only like every other call
 *   (legal under graph capture).
 */
size_t yh_nms_workspace_bytes(int64_t batch, int N, int own_boxes);
int yh_decode_nms_ws(const float* pred, const float* bias, float* dets, int32_t* keep_count,
                     int32_t* keep_idx, int mode, float conf_thresh, float iou_thresh,
                     int64_t batch, int S, int Bx, int C, int img_size, void* workspace,
                     size_t workspace_bytes, void* stream);
int yh_nms_ws(const float* cand_ltrb, const float* cand_score, const int32_t* cand_cls,
              const uint8_t* cand_present, float* dets, int32_t* keep_count, int32_t* keep_idx,
              float conf_thresh, float iou_thresh, int64_t batch, int N, void* workspace,
              size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_HEAD_H */
