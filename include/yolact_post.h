/*
 * libyolact_post.so - C ABI of the MI355X (gfx950) YOLACT post-processing path.
 *
 * The reference (anshkumar/yolact, TensorFlow 2) has no FFI/plugin interface on
 * this path: the unit of replacement is the Python class `Detect`
 * (detection.py:6-117) that `Yolact.call` invokes at yolact.py:192.  The entry
 * points below are what a ctypes binding of that class needs; each one cites the
 * reference interface it stands in for.  yolact_amd/detect.py is that binding
 * and INTEGRATION.md shows the stub a maintainer would add to yolact.py.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on
 * success and a negative YP_E* code on failure (text: yp_last_error(), per
 * thread); no exceptions cross the ABI.  All tensor pointers are DEVICE
 * pointers on the handle's device, fp32, C-contiguous.  The caller owns every
 * input/output buffer; the handle owns its scratch.  yp_run is asynchronous on
 * `hip_stream` and allocates nothing.  One handle serves one stream at a time;
 * distinct handles are independent (one per GPU when sharding a batch).
 * Every entry point leaves the calling thread's current HIP device unchanged.
 * There is no CPU backend: without a HIP device yp_create fails.
 *
 * Experiment knobs (environment, read ONCE in yp_create, never on the launch
 * path): YP_TUNE_MASK_NT=1 (non-temporal mask stores), YP_TUNE_SF_WAVES=<n>
 * (score_filter workgroups per CU the plan assumes), YP_TUNE_PREFILL=0|1
 * (mask rows zero-filled by a launch of its own before mask_assemble / by fill
 * workgroups inside the NMS launch (default)).
 */
#ifndef YOLACT_POST_H
#define YOLACT_POST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YP_VERSION 310 /* 0.3.1 */

enum {
  YP_OK = 0,
  YP_EINVAL = -1,   /* bad argument / unsupported shape            */
  YP_ENODEV = -2,   /* no HIP device / bad device id               */
  YP_ENOMEM = -3,   /* scratch allocation failed                   */
  YP_EHIP = -4      /* a HIP runtime call failed                   */
};

enum {
  YP_NMS_CROSS_CLASS = 0, /* Detect._cc_fast_nms, detection.py:181-209 (the live path) */
  YP_NMS_PER_CLASS = 1    /* intent of Detect._fast_nms, detection.py:211-250 (dead code;
                             per-class top_k + cross-class top max_dets, 1-based class ids) */
};

/* Immutable after yp_create.  Mirrors Detect.__init__ (detection.py:13-22) plus the shapes
 * Detect.__call__ reads dynamically (detection.py:44-53) and the literals it hard-codes. */
typedef struct yp_config {
  int32_t num_anchors;      /* A: pred_cls.shape[1]                                   */
  int32_t num_classes;      /* C incl. background: pred_cls.shape[2] (2..256)         */
  int32_t mask_dim;         /* K: pred_mask_coef.shape[2]; must be 32                 */
  int32_t proto_h, proto_w; /* proto_out.shape[1:3]                                   */
  int32_t img_h, img_w;     /* img_shape[1], img_shape[2] (detection.py:92-98)        */
  float conf_thresh;        /* Detect.conf_thresh, strict '>' (detection.py:73)       */
  float nms_thresh;         /* IoU threshold, '<=' keeps (detection.py:202); the live
                               reference always uses 0.5 here                          */
  int32_t nms_mode;         /* YP_NMS_*                                               */
  int32_t top_k;            /* _cc_fast_nms top_k (15 in the reference); 1..256       */
  int32_t max_dets;         /* rows per image in every output (row stride); the first
                               min(kept, max_dets) score-descending detections are
                               written.  Reference layout: 300 (detection.py:21)       */
  int32_t use_cropped_mask; /* detection.py:102 (1 in every reference call)            */
  int32_t zero_pad_masks;   /* 1: every mask row >= num_det is written as zeros, which
                               is the reference's tf.zeros layout (detection.py:68);
                               0: rows >= num_det are zeros where a detection could
                               have landed (rows < min(max_dets, 256), and < top_k
                               in cross-class mode) and untouched beyond            */
  float mask_threshold;     /* < 0: float masks only.  >= 0: det_bitmaps = mask > thr
                               (the callers' binarisation, train.py:497)               */
  int32_t max_batch;        /* largest batch yp_run will be given                      */
  int32_t device_id;        /* HIP device ordinal, >= 0                                */
} yp_config;

typedef struct yp_handle yp_handle;

/* Detect.__init__ (detection.py:13-22).  Validates like it (nms_thresh <= 0 is
 * YP_EINVAL, cf. the ValueError at L17-18) and allocates the device scratch. */
int yp_create(const yp_config* cfg, yp_handle** out);

/* yp_create with bounded candidate lists (no counterpart in the reference, whose tf.boolean_mask allocates
 * per call).  yp_create sizes score_filter's key lists for the worst case - every anchor emitting
 * min(C-1, floor(1/conf)+1) keys in per-class mode: 94 MB at 19248 anchors x 32 images of which a real
 * batch touches under 1 MB.  Here the lists are sized for `keys_per_anchor` (>= 1) keys per anchor of a work
 * item instead.  If a work item (a run of a few hundred consecutive anchors of one image) ever emits more, its
 * LAST keys are dropped - the result is then that of the same image with those (anchor, class) candidates
 * absent - and the handle's overflow flag is set.  yp_overflowed() synchronises, returns 1 if the flag was
 * set since the last call (and clears it), 0 if not; with yp_create's worst-case lists it is always 0. */
int yp_create_bounded(const yp_config* cfg, int32_t keys_per_anchor, yp_handle** out);
int yp_overflowed(yp_handle* h);

/* Releases the scratch.  (Python: the object going out of scope.) */
int yp_destroy(yp_handle* h);

/* Detect.__call__ (detection.py:24-117) for `batch` images.
 *   inputs  (net_outs, detection.py:44-48):
 *     pred_offset [batch,A,4]  pred_cls [batch,A,C] raw logits  pred_mask_coef [batch,A,K]
 *     priors [A,4] = [ymin,xmin,ymax,xmax] pixels               proto [batch,ph,pw,K]
 *   outputs (result dict, detection.py:116), row stride max_dets:
 *     det_boxes   [batch,max_dets,4]  normalised [ymin,xmin,ymax,xmax]   (L93-98, L110)
 *     det_classes [batch,max_dets]    float, 1-based, 0 = empty          (L111)
 *     det_scores  [batch,max_dets]    descending in the filled prefix    (L112)
 *     det_masks   [batch,max_dets,ph,pw] float in [0,1]; may be NULL     (L113)
 *     det_bitmaps [batch,max_dets,ph,pw] uint8, NULL unless mask_threshold >= 0
 *     num_det     [batch] int32                                          (L114)
 *     kept_anchor_idx [batch,max_dets] int32, original anchor index of each kept
 *                 detection, -1 padded (not exposed by the reference; may be NULL)
 *   Rows >= num_det of boxes/classes/scores are zero.  Asynchronous on hip_stream
 *   (NULL = the default stream). */
int yp_run(yp_handle* h, int32_t batch,
           const float* pred_offset, const float* pred_cls, const float* pred_mask_coef,
           const float* priors, const float* proto,
           float* det_boxes, float* det_classes, float* det_scores,
           float* det_masks, uint8_t* det_bitmaps,
           int32_t* num_det, int32_t* kept_anchor_idx,
           void* hip_stream);

/* Output sets the library keeps track of.  NO COUNTERPART IN THE REFERENCE (its tf.zeros + scatter allocate fresh
 * tensors per call, detection.py:65-68,109-113): it is for callers that own a few output sets and reuse them - a serving
 * loop, an evaluation loop that consumes each result before the next batch.  yp_run must assume that det_masks holds
 * anything and writes every row (at config 2: 244 MB of zeros per batch for ~3 MB of mask values).  A yp_kept remembers
 * what the last call wrote into ONE set of mask buffers - the crop boxes of its detections - so that the next call
 * clears exactly those pixels and writes its own: same results, bit for bit, without the refill.
 *   yp_kept_create   device memory for the record (2 x max_batch x (rows x 16 + 4) bytes), allocated here, once
 *   yp_run_kept      yp_run on a tracked set.  The FIRST call with a yp_kept zero-fills the mask buffers completely
 *                    (every row yp_run would write, the padding rows of zero_pad_masks included) and binds the yp_kept
 *                    to these det_masks / det_bitmaps pointers and to this batch size; later calls must pass the same
 *                    pointers and batch (YP_EINVAL otherwise) and must not be overlapped with each other.  Nobody else
 *                    may write the mask buffers in between.  det_masks must not be NULL.
 *   yp_kept_reset    forget the record: the next call refills (after the caller wrote into the buffers itself)
 * The record's two slots swap on the HOST at every call: a yp_run_kept captured in a HIP graph would replay one
 * side for ever - capture yp_run / yp_run_static instead.  yp_kept_create allocates (hipMalloc); yp_run_kept does not. */
typedef struct yp_kept yp_kept;
int yp_kept_create(yp_handle* h, yp_kept** out);
int yp_kept_destroy(yp_kept* k);
int yp_kept_reset(yp_kept* k);
int yp_run_kept(yp_handle* h, yp_kept* k, int32_t batch,
                const float* pred_offset, const float* pred_cls, const float* pred_mask_coef,
                const float* priors, const float* proto,
                float* det_boxes, float* det_classes, float* det_scores,
                float* det_masks, uint8_t* det_bitmaps,
                int32_t* num_det, int32_t* kept_anchor_idx,
                void* hip_stream);

/* yp_run without a handle and without any allocation.  NO COUNTERPART IN THE REFERENCE: it is for
 * callers that cannot allocate device memory - code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  Same inputs, outputs and results as yp_run (cfg->max_batch is
 * ignored), but the scratch is a static array of YP_STATIC_SCRATCH_BYTES inside the library's code
 * object, one copy per GPU, held until the process ends.  A batch whose scratch exceeds the array is
 * cut into equal chunks of images that are processed one after the other on hip_stream; a
 * configuration of which a single image does not fit is rejected with YP_EINVAL (per-class mode
 * needs about 8 * num_anchors * min(num_classes-1, 1/conf_thresh + 1) bytes per image).  It calls no
 * hipMalloc / hipFree, creates no stream or event and does not synchronise.  Every call on a GPU uses that GPU's one array: calls
 * that overlap on one GPU (two streams, two threads) race and give wrong results - serialise them.
 * The FIRST call with a (device, configuration) pair makes that pair's launch plan (attribute / occupancy queries,
 * the YP_TUNE_* environment, and once per device hipGetSymbolAddress): make it eagerly, outside any graph capture.
 * The plans of the last four configurations per device are remembered; later calls with them query nothing. */
#define YP_STATIC_SCRATCH_BYTES (64u << 20)
int yp_run_static(const yp_config* cfg, int32_t batch,
                  const float* pred_offset, const float* pred_cls, const float* pred_mask_coef,
                  const float* priors, const float* proto,
                  float* det_boxes, float* det_classes, float* det_scores,
                  float* det_masks, uint8_t* det_bitmaps,
                  int32_t* num_det, int32_t* kept_anchor_idx,
                  void* hip_stream);

/* Blocks until the handle's last yp_run has finished.  (TF: fetching the tensors.) */
int yp_sync(yp_handle* h);

/* Number of anchors that passed the confidence filter (tf.boolean_mask, detection.py:73)
 * per image of the LAST yp_run; synchronises.  host_out: [batch] int32 on the host. */
int yp_last_survivors(yp_handle* h, int32_t* host_out, int32_t batch);

/* Per-kernel timing for benchmarks (no reference counterpart).  When enabled, yp_run brackets
 * each kernel with HIP events on hip_stream; yp_profile_read synchronises and returns the
 * last run's durations in milliseconds: [0] score_filter [1] nms (top-k + Fast NMS + final
 * selection, with the mask zero-fill riding in the same launch) [2] mask_assemble. */
#define YP_NUM_KERNELS 3
int yp_profile_enable(yp_handle* h, int32_t on);
int yp_profile_read(yp_handle* h, float* ms_out /* [YP_NUM_KERNELS] */);

/* Anchor._generate_anchors + Anchor.get_anchors (data/anchor.py:23-68): the `priors` tensor Detect
 * consumes, built on the device.  fmap_hw = [num_levels][2] feature-map (height, width)
 * (yolact.py:75-86), scales = one per level, aspect_ratios = num_ratios values (train.py:65-68);
 * 1..8 of each.  Writes [A,4] float32 [ymin,xmin,ymax,xmax] pixels in the reference's anchor
 * order (level, row, column, ratio) to `priors` (device, 16-byte aligned, room for `capacity`
 * anchors), asynchronously on hip_stream, and returns A.  With priors == NULL nothing is launched
 * and A is returned (size query; works without a device).  Negative YP_E* code on failure. */
int64_t yp_make_priors(int32_t img_h, int32_t img_w, int32_t num_levels, const int32_t* fmap_hw,
                       const double* scales, int32_t num_ratios, const double* aspect_ratios,
                       float* priors, int64_t capacity, int32_t device_id, void* hip_stream);

/* ---- the consumer step after Detect: resize + binarise + run-length encode ----
 * What every caller of the reference does with detection_masks on the CPU, per detection:
 *   YP_RESIZE_THEN_THRESHOLD  cv2.resize(mask, (out_w, out_h)) on the float mask, then `> threshold`
 *                             (coco_eval.py:133-135, infer.py:31-33)
 *   YP_THRESHOLD_THEN_RESIZE  `mask > threshold` first, astype(uint8), then cv2.resize of the 0/1
 *                             bytes (train.py:497,503-506)
 * followed by pycocotools.mask.encode of the Fortran-ordered bitmap (coco_eval.py:135-136,
 * utils/coco_tools.py:376-388).  cv2.resize = OpenCV INTER_LINEAR, restated (see csrc/rle.hip).
 *   det_masks [batch,max_dets,mask_h,mask_w] float, num_det [batch] int32: yp_run's outputs (device)
 *   bitmaps   [batch,max_dets,out_h,out_w] uint8 0/1, row-major; may be NULL
 *   rle_counts[batch,max_dets,rle_cap] uint32: the uncompressed COCO counts of every detection
 *             n < num_det[b] - lengths of alternating runs of 0s and 1s in column-major order,
 *             starting with the 0s (coco_eval.py:91-98 binary_mask_to_rle states exactly this list)
 *   rle_len   [batch,max_dets] int32: number of counts (0 for rows >= num_det).  Only the first
 *             rle_cap counts are stored; rle_len > rle_cap tells the caller to retry with more room
 *             (out_h*out_w + 1 always suffices).
 * Limits: mask_h <= 512, out_h <= 4096.  Output sizes smaller than the mask are resized as cv::resize's
 * INTER_LINEAR does (two taps, no anti-aliasing) - EXCEPT mask_h == 2*out_h && mask_w == 2*out_w, where
 * cv::resize silently switches to INTER_AREA: that one size pair is rejected with YP_EINVAL.
 * Asynchronous on hip_stream; allocates nothing. */
enum { YP_RESIZE_THEN_THRESHOLD = 0, YP_THRESHOLD_THEN_RESIZE = 1 };
int yp_masks_to_rle(int32_t device_id, int32_t batch, int32_t max_dets, int32_t mask_h, int32_t mask_w,
                    const float* det_masks, const int32_t* num_det,
                    int32_t out_h, int32_t out_w, float threshold, int32_t order,
                    uint8_t* bitmaps, uint32_t* rle_counts, int32_t* rle_len, int32_t rle_cap,
                    void* hip_stream);

/* The same with every image's OWN output size, as coco_eval.py:112-137 does it (`_h, _w` of the original image
 * per batch element): out_hw = HOST array [batch][2] of (height, width).  One call for a batch of differently
 * sized originals (launches of up to 32 images each, the sizes travel in the kernels' argument blocks: still no
 * allocation and no copy).  No bitmaps here; rle_counts / rle_len as above (rle_cap >= max(out_h*out_w) + 1
 * always suffices). */
int yp_masks_to_rle_sized(int32_t device_id, int32_t batch, int32_t max_dets, int32_t mask_h, int32_t mask_w,
                          const float* det_masks, const int32_t* num_det,
                          const int32_t* out_hw, float threshold, int32_t order,
                          uint32_t* rle_counts, int32_t* rle_len, int32_t rle_cap,
                          void* hip_stream);

/* ---- mask assembly of the training loss (forward - the gradients follow below) ----
 * loss/loss_yolact.py:239-253 (YOLACTLoss._loss_mask): mask_p = sigmoid(proto_out[i] @ pos_coef^T),
 * cropped by the positives' prior boxes (pixels, normalised by img_h / img_w at L247-252) with the
 * same utils.crop as Detect (L253), for every image of the batch in one launch of kernel 4.
 *   proto [batch,ph,pw,32]   coef [batch,rows,32]   boxes [batch,rows,4] = [ymin,xmin,ymax,xmax] pixels
 *   num_rows [batch] int32: positives of every image (<= rows <= 256; the loss caps them at 100)
 *   crop_workspace [batch,rows,4] float: scratch the caller provides (nothing is allocated)
 *   out_masks [batch,rows,ph,pw]: rows < num_rows[b] = the loss's `_mask_p` (L273), later rows zeros
 * All device pointers, 16-byte aligned; asynchronous on hip_stream. */
int yp_assemble_masks(int32_t device_id, int32_t batch, int32_t rows, int32_t proto_h, int32_t proto_w,
                      int32_t img_h, int32_t img_w, int32_t use_cropped_mask,
                      const float* proto, const float* coef, const float* boxes, const int32_t* num_rows,
                      float* crop_workspace, float* out_masks, void* hip_stream);

/* ---- gradients of that mask assembly (backward) ----
 * The reference has no code of its own here: TensorFlow's autodiff differentiates loss/loss_yolact.py:239-253
 * (tf.linalg.matmul L239, tf.sigmoid L241, utils.crop L253).  Given the forward's output `masks` and
 * grad_masks = dL/dmasks (both [batch,rows,ph,pw]):   dz = (g * m) * (1 - m)   (SigmoidGrad; m is 0 outside
 * the crop box, and so is dz);   grad_proto[b,p,k] = sum_n dz[n,p] coef[b,n,k];   grad_coef[b,n,k] =
 * sum_p dz[n,p] proto[b,p,k];   grad_coef of rows >= num_rows[b] is 0.  `masks` must be yp_assemble_masks'
 * output for the same proto / coef / boxes / num_rows (only the pixels inside a row's box are read).
 * Deterministic: fixed summation orders, no float atomics (csrc/loss_grad.hip states them;
 * oracle/detect_oracle.c restates them, and the GPU tests compare bit for bit).
 *   grad_proto [batch,ph,pw,32]   grad_coef [batch,rows,32]
 *   workspace: yp_assemble_masks_backward_workspace(batch, rows, proto_h) bytes of device memory the caller
 *              provides (= 4*batch*rows*(4 + 32*proto_h): the crop edges and grad_coef per pixel row)
 * All device pointers; proto, coef, boxes, workspace, grad_proto 16-byte aligned; proto_w <= 1000 (a prototype row
 * must fit in LDS; checked before anything is launched);
 * asynchronous on hip_stream; allocates nothing. */
int64_t yp_assemble_masks_backward_workspace(int32_t batch, int32_t rows, int32_t proto_h);
int yp_assemble_masks_backward(int32_t device_id, int32_t batch, int32_t rows, int32_t proto_h, int32_t proto_w,
                               int32_t img_h, int32_t img_w, int32_t use_cropped_mask,
                               const float* proto, const float* coef, const float* boxes, const int32_t* num_rows,
                               const float* masks, const float* grad_masks,
                               float* workspace, float* grad_proto, float* grad_coef, void* hip_stream);

/* pycocotools' compressed counts string (maskApi.c rleToString, what mask.encode puts in
 * rle['counts'], coco_eval.py:136-137) of n HOST counts.  Writes at most cap-1 characters and a
 * terminating NUL to `out`; returns the length the full string has (>= cap: it did not fit), or a
 * negative YP_E* code.  Pure host code, no device needed. */
int yp_rle_string(const uint32_t* counts, int32_t n, char* out, int32_t cap);

/* The same for n_lists lists at once: list i is the lens[i] counts from counts[starts[i]] on, in a HOST array of
 * n_counts counts (any layout: rows of a matrix, or the lists packed back to back).  The strings are written back
 * to back, each NUL-terminated, string i at out + offsets[i]; offsets has n_lists + 1 entries.  Returns the bytes
 * all strings need (> out_cap: they did not fit; 7*sum(lens) + n_lists always suffices), or a negative YP_E*
 * code.  One call per batch instead of one per detection. */
int64_t yp_rle_strings(const uint32_t* counts, const int64_t* starts, const int32_t* lens, int32_t n_lists,
                       int64_t n_counts, char* out, int64_t out_cap, int64_t* offsets);

/* Bytes of device scratch the handle holds. */
int64_t yp_scratch_bytes(const yp_handle* h);

/* Text of the calling thread's last error ("" if none). */
const char* yp_last_error(void);

/* YP_VERSION of the loaded library. */
int yp_version(void);

#ifdef __cplusplus
}
#endif
#endif /* YOLACT_POST_H */
