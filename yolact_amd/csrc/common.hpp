// Shared declarations of libyolact_post.so (gfx950 only).
//
// Arithmetic contract "YP-MATH" (DESIGN.md section 3): every value the decisions of the
// reference depend on (scores, decoded boxes, IoU) is computed with single IEEE f32
// operations in the order the reference's Python expressions evaluate them, an explicit-fma
// Cephes exp, a fixed softmax reduction tree, one IEEE reciprocal per softmax row and IEEE
// division elsewhere.  All translation units are
// built with -ffp-contract=off; fma is only ever written explicitly.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yolact_post.h"

#pragma clang fp contract(off)

typedef unsigned long long u64;

#define YP_WAVE 64
#define YP_SORT_MAX 256   // largest top_k / final selection handled by one workgroup sort
#define YP_MASK_DIM 32

// ---- candidate keys -----------------------------------------------------------------------
// A candidate is one 64-bit key whose unsigned order is the order tf.math.top_k imposes
// (detection.py:186: score descending, ties -> lower index first).
//   segment key (score_filter -> segment_nms):  [score bits:32][0xFFFFFF - anchor:24][class:8]
//   final key   (segment_nms -> final_select):  [score bits:32][255 - class:8][0xFFFFFF - anchor:24]
// Scores are positive floats, so their bit patterns order like their values.
__host__ __device__ inline u64 yp_seg_key(float score, int anchor, int cls) {
  return ((u64)__builtin_bit_cast(uint32_t, score) << 32) | ((u64)(0xFFFFFFu - (uint32_t)anchor) << 8) | (u64)(uint32_t)cls;
}
__host__ __device__ inline float yp_key_score(u64 k) { return __builtin_bit_cast(float, (uint32_t)(k >> 32)); }
__host__ __device__ inline int yp_seg_key_anchor(u64 k) { return (int)(0xFFFFFFu - (uint32_t)((k >> 8) & 0xFFFFFFu)); }
__host__ __device__ inline int yp_seg_key_class(u64 k) { return (int)(k & 0xFFu); }
__host__ __device__ inline u64 yp_fin_key(float score, int anchor, int cls) {
  return ((u64)__builtin_bit_cast(uint32_t, score) << 32) | ((u64)(255u - (uint32_t)cls) << 24) | (u64)(0xFFFFFFu - (uint32_t)anchor);
}
__host__ __device__ inline int yp_fin_key_anchor(u64 k) { return (int)(0xFFFFFFu - (uint32_t)(k & 0xFFFFFFu)); }
__host__ __device__ inline int yp_fin_key_class(u64 k) { return (int)(255u - (uint32_t)((k >> 24) & 0xFFu)); }

#ifdef __HIPCC__
// ---- YP-MATH device functions -------------------------------------------------------------
template <bool NONPOS>
__device__ __forceinline__ float yp_exp_t(float x) {
  x = fmaxf(x, -87.0f);
  if (!NONPOS) x = fminf(x, 87.0f);   // a no-op for x <= 0 (softmax arguments), so it is skipped there
  const float fx = floorf(__builtin_fmaf(x, 1.44269504088896341f, 0.5f));
  float r = __builtin_fmaf(fx, -0.693359375f, x);
  r = __builtin_fmaf(fx, 2.12194440e-4f, r);
  const float z = r * r;
  float y = 1.9875691500E-4f;
  y = __builtin_fmaf(y, r, 1.3981999507E-3f);
  y = __builtin_fmaf(y, r, 8.3334519073E-3f);
  y = __builtin_fmaf(y, r, 4.1665795894E-2f);
  y = __builtin_fmaf(y, r, 1.6666665459E-1f);
  y = __builtin_fmaf(y, r, 5.0000001201E-1f);
  y = __builtin_fmaf(y, z, r);
  y = y + 1.0f;
  const int n = (int)fx;
  return y * __builtin_bit_cast(float, (uint32_t)(n + 127) << 23);
}

// Two exponentials of non-positive arguments at once, on gfx950's packed fp32 pipe (v_pk_fma_f32,
// v_pk_mul_f32, v_pk_add_f32): each half is the same sequence of IEEE operations as yp_exp_t<true>, so the
// bits are those of two scalar calls; the instruction count is ~55 % of theirs.
typedef float yp_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ yp_f2 yp_exp2_le0(yp_f2 x) {
  x.x = fmaxf(x.x, -87.0f);
  x.y = fmaxf(x.y, -87.0f);
  const yp_f2 t = __builtin_elementwise_fma(x, (yp_f2){1.44269504088896341f, 1.44269504088896341f}, (yp_f2){0.5f, 0.5f});
  const yp_f2 fx = {floorf(t.x), floorf(t.y)};
  yp_f2 r = __builtin_elementwise_fma(fx, (yp_f2){-0.693359375f, -0.693359375f}, x);
  r = __builtin_elementwise_fma(fx, (yp_f2){2.12194440e-4f, 2.12194440e-4f}, r);
  const yp_f2 z = r * r;
  yp_f2 y = {1.9875691500E-4f, 1.9875691500E-4f};
  y = __builtin_elementwise_fma(y, r, (yp_f2){1.3981999507E-3f, 1.3981999507E-3f});
  y = __builtin_elementwise_fma(y, r, (yp_f2){8.3334519073E-3f, 8.3334519073E-3f});
  y = __builtin_elementwise_fma(y, r, (yp_f2){4.1665795894E-2f, 4.1665795894E-2f});
  y = __builtin_elementwise_fma(y, r, (yp_f2){1.6666665459E-1f, 1.6666665459E-1f});
  y = __builtin_elementwise_fma(y, r, (yp_f2){5.0000001201E-1f, 5.0000001201E-1f});
  y = __builtin_elementwise_fma(y, z, r);
  y = y + (yp_f2){1.0f, 1.0f};
  const int nx = (int)fx.x, ny = (int)fx.y;
  const yp_f2 sc = {__builtin_bit_cast(float, (uint32_t)(nx + 127) << 23), __builtin_bit_cast(float, (uint32_t)(ny + 127) << 23)};
  return y * sc;
}

__device__ __forceinline__ float yp_exp(float x) { return yp_exp_t<false>(x); }
__device__ __forceinline__ float yp_exp_le0(float x) { return yp_exp_t<true>(x); }   // caller guarantees x <= 0

__device__ __forceinline__ float yp_sigmoid(float x) { return 1.0f / (1.0f + yp_exp(-x)); }

__device__ __forceinline__ float yp_clip(float t, float lo, float hi) { return fminf(fmaxf(t, lo), hi); }

// utils._decode for one anchor (utils/utils.py:193-213).  pr = prior (ymin,xmin,ymax,xmax) pixels,
// of = (dx,dy,dw,dh); returns the box as (ymin,xmin,ymax,xmax) pixels.
__device__ __forceinline__ float4 yp_decode(const float4 pr, const float4 of) {
  const float ph = pr.z - pr.x;                    // L193
  const float pw = pr.w - pr.y;                    // L194
  const float cx = pr.y + (pw / 2.0f);             // L196
  const float cy = pr.x + (ph / 2.0f);             // L197
  const float bx = cx + (of.x * pw) * 0.1f;        // L201
  const float by = cy + (of.y * ph) * 0.1f;
  const float bw = pw * yp_exp(of.z * 0.2f);       // L202
  const float bh = ph * yp_exp(of.w * 0.2f);
  float4 box;
  box.y = bx - bw / 2.0f;   // xmin                   L210
  box.x = by - bh / 2.0f;   // ymin
  box.w = bw / 2.0f + bx;   // xmax
  box.z = bh / 2.0f + by;   // ymax
  return box;                                      // L213
}

// utils.sanitize_coordinates (utils/utils.py:89-104).
__device__ __forceinline__ void yp_sanitize(float a, float b, float img_size, bool normalized, float padding,
                                            float& x1, float& x2) {
  if (normalized) { a = a * img_size; b = b * img_size; }
  const float lo = fminf(a, b), hi = fmaxf(a, b);
  x1 = yp_clip(lo - padding, 0.0f, img_size);
  x2 = yp_clip(hi + padding, 0.0f, img_size);
}
#endif  // __HIPCC__

// ---- device scratch owned by a handle -----------------------------------------------------
#ifndef YP_SF_TA
#define YP_SF_TA 32   // anchors per score_filter tile (a multiple of 4, at most 32)
#endif

#define YP_NLISTS 16       // candidate lists per image: one per class group (per-class mode), or the image's
                           // score_filter items dealt round-robin (cross-class mode)
#define YP_CNT_STRIDE 32   // ints per list counter in the counter block (16 x 32 ints per image)
// List l's counter inside its image's block: lists 2i and 2i+1 are the two halves of one 64-bit word - score_filter
// adds to both with one atomic - and every word has 256 bytes of its own.
#define YP_CNT_IDX(l) (((l) >> 1) * (2 * YP_CNT_STRIDE) + ((l) & 1))

struct yp_scratch {
  u64* lkeys;           // [batch][YP_NLISTS][lcap] candidate keys, appended by score_filter in no particular order
  float4* lbox;         // [batch][YP_NLISTS][lcap] their decoded boxes (ymin,xmin,ymax,xmax pixels), same positions
  int32_t* lcount;      // [max_batch][YP_NLISTS] x YP_CNT_STRIDE: entries appended (can exceed lcap: the excess went to
                        // the overflow list); zero between launches - the NMS workgroup that reads a counter resets it
  u64* okeys;           // [batch][ocap] per-class mode: keys that found their list full (no boxes); read by every class
                        // group of the image
  int32_t* ocount;      // [max_batch] x YP_CNT_STRIDE; zero between launches (reset by the image's final selection)
  int32_t* seg_surv;    // [nitems] anchors past the confidence filter per score_filter item (rewritten every run)
  int32_t* stat_surv;   // [max_batch] anchors past the confidence filter, for yp_last_survivors
  u64* pool;            // [max_batch][ngroups][gcap] per-class NMS survivors (per-class mode), dense per class group
  float4* pool_box;     // [max_batch][ngroups][gcap] their decoded boxes
  int32_t* pool_count;  // [max_batch][ngroups] (rewritten every run)
  int32_t* arrive;      // [max_batch] class groups of an image that have finished; zero between launches
  int32_t* overflow;    // [1] set by score_filter when an image had more keys than its lists hold (yp_create_bounded)
  float4* crop;         // [max_batch][sel_rows] (x1,x2,y1,y2) crop edges in proto pixels
  int32_t* sel_anchor;  // [max_batch][sel_rows] anchor of every kept detection: mask_assemble reads its coefficient
                        // row straight from pred_mask_coef
};

// Experiment knobs, read from the environment ONCE, in yp_create (never on the launch path):
//   YP_TUNE_MASK_NT=1        mask_assemble stores its rows non-temporally
//   YP_TUNE_SF_WAVES=<n>     score_filter: single-wave workgroups per CU the plan assumes
//   YP_TUNE_PREFILL=0|1      0: mask_assemble zero-fills its rows itself (pass Z); 1 (default): the zeros
//                            are written by fill workgroups that ride in the NMS launch
struct yp_tune {
  int mask_nt;
  int sf_waves;
  int prefill;
};

struct yp_handle {
  yp_config cfg;
  yp_tune tune;
  int nseg;             // segments per image: 1 (cross-class) or C-1 (per-class)
  int sel_rows;         // min(max_dets, YP_SORT_MAX): rows a detection can land in
  int lds_stride;       // floats per anchor row in score_filter's LDS tile
  float pre_thresh;     // score_filter's early-reject bound on x_bg - x_fgmax
  float jen_thr;        // score_filter's second reject: bound on 1 + exp(x_bg-xf) + (C-2) exp(mean-xf); NaN = off
  int num_cus;          // compute units of the device (256 on MI355X)
  int sf_bufb;          // score_filter: bytes of a wave's LDS tile buffer
  int sf_waves_per_cu;  // score_filter: single-wave workgroups resident per CU
  int keys_per_anchor;  // keys per anchor the lists are sized for: the worst case below, or yp_create_bounded's bound
  int keys_worst;       // most keys one anchor can emit: 1 (cross-class) or min(C-1, floor(1/conf)+1)
#ifdef YP_STAMPS
  void* dbg;            // diagnostic build only: cycle stamps
#endif
  void* scratch_base;
  int64_t scratch_bytes;
  yp_scratch s;
  hipStream_t last_stream;
  int last_batch;
  int profile;
  hipEvent_t ev[YP_NUM_KERNELS + 1];
};

// How score_filter cuts a batch into work items: `parts` items per image, as many as keep every item
// resident at once, each a run of consecutive tiles (`longest` tiles at most).
// (Tried: shrinking the tile to 29 anchors so that 95 items x 7 tiles cover A = 19248 instead of
// 86 x 7 x 32 with ten idle wave slots per image - 49.1 us against 48.0: more resident waves do not help
// this kernel, and tiles that start off a 16-byte boundary cost more than the balance gains.)
static inline void yp_plan_items(const yp_handle* h, int batch, int* parts, int* longest_out) {
  const int ntiles = (h->cfg.num_anchors + YP_SF_TA - 1) / YP_SF_TA;
  const int resident = h->num_cus * h->sf_waves_per_cu;
  int pt = batch > 0 ? resident / batch : 1;
  pt = pt < 1 ? 1 : (pt > ntiles ? ntiles : pt);
  // the longest item sets the kernel's length: keep its tile count, but use only as many items as
  // that needs, so that (nearly) all items are equally long (602 tiles: 86 x 7 instead of 96 x 6..7)
  const int longest = (ntiles + pt - 1) / pt;
  pt = (ntiles + longest - 1) / longest;
  *parts = pt;
  *longest_out = longest;
}

// Per-class mode: the classes of an image are dealt round-robin to `ngroups` workgroups; a group's
// pool list holds at most gcap keys (top_k per class).
static inline void yp_plan_groups(const yp_handle* h, int* ngroups, int* gcap) {
  const int ng = h->nseg < YP_NLISTS ? h->nseg : YP_NLISTS;
  *ngroups = ng;
  *gcap = ((h->nseg + ng - 1) / ng) * h->cfg.top_k;
}

// Where score_filter's candidates go: YP_NLISTS lists of lcap entries (key + box) per image, and - per-class
// mode only - one overflow list of ocap keys per image for the keys that find their list full.
//   cross-class  item `part` of an image appends to list part % YP_NLISTS; an anchor emits at most one key, so
//                lcap = the anchors of the items that share a list: nothing can overflow
//   per-class    list g takes the classes c with (c-1) % ngroups == g.  Its honest worst case is one key per
//                (anchor, class of the group) - 37 MB per image at config 2 - so the lists are sized for what a
//                class group can usefully hold (max(2048, A/8) entries, the whole worst case if that is less)
//                and the overflow list for everything an image can emit (A * keys_per_anchor keys, the bound
//                of yp_create / yp_create_bounded); a class group that finds overflow keys takes its slow path
//                (nms.hip).  Only when the overflow list itself is full is a key dropped (bounded lists only).
static inline void yp_plan_lists(const yp_handle* h, int batch, long long* lcap, long long* ocap) {
  const long long A = h->cfg.num_anchors;
  if (h->cfg.nms_mode == YP_NMS_CROSS_CLASS) {
    int parts, longest;
    yp_plan_items(h, batch, &parts, &longest);
    long long c = (long long)((parts + YP_NLISTS - 1) / YP_NLISTS) * longest * YP_SF_TA;
    *lcap = c < A ? c : A;
    *ocap = 0;
  } else {
    int ng, gcap;
    yp_plan_groups(h, &ng, &gcap);
    long long kpl = (h->nseg + ng - 1) / ng;                       // classes per group
    if (h->keys_per_anchor < kpl) kpl = h->keys_per_anchor;
    const long long worst = A * kpl;
    long long room = A / 8 > 2048 ? A / 8 : 2048;
    *lcap = worst < room ? worst : room;
    *ocap = A * h->keys_per_anchor;
  }
}


int yp_score_filter_waves_per_cu(const yp_handle* h);

// Kernel launchers (one per translation unit).  All asynchronous on `st`.
long long yp_priors_count(int num_levels, const int32_t* fmap_hw, int num_ratios);
hipError_t yp_launch_make_priors(int img_h, int img_w, int num_levels, const int32_t* fmap_hw, const double* scales,
                                 int num_ratios, const double* sqrt_ratios, float* priors, hipStream_t st);
hipError_t yp_launch_score_filter(const yp_handle* h, int batch, const float* pred_cls, const float* pred_offset,
                                  const float* priors, hipStream_t st);
// A tracked output set's record for one call (yp_run_kept): the NMS launch writes the kept detections' crop edges
// to crop_new instead of the handle's scratch; mask_assemble clears what crop_prev / num_prev say the call before
// wrote, writes its own and leaves the count in num_new.
struct yp_track {
  float4* crop_new;          // [batch][sel_rows]
  const float4* crop_prev;   // [batch][sel_rows]
  const int32_t* num_prev;   // [batch]
  int32_t* num_new;          // [batch]
};
hipError_t yp_launch_nms(const yp_handle* h, int batch, const float* pred_offset, const float* priors, float* det_boxes,
                         float* det_classes, float* det_scores, int32_t* num_det, int32_t* kept_idx,
                         float* fill_masks, uint8_t* fill_bitmaps, int fill_rows, hipStream_t st, float4* crop_out = nullptr);
int yp_mask_fill_rows(const yp_handle* h);
hipError_t yp_launch_fill(float* masks, uint8_t* bitmaps, long long P, int fill_rows, int max_dets, int batch, hipStream_t st);
int yp_mask_prefill(const yp_handle* h, int batch);
hipError_t yp_launch_mask_assemble(const yp_handle* h, int batch, const float* proto, const float* pred_mask_coef,
                                   const int32_t* num_det, float* det_masks, uint8_t* det_bitmaps, hipStream_t st,
                                   const yp_track* track = nullptr);
hipError_t yp_launch_mask_rle(int batch, int max_dets, int ph, int pw, const float* det_masks, const int32_t* num_det,
                              int out_h, int out_w, const int32_t* out_hw, float threshold, int order, uint8_t* bitmaps,
                              uint32_t* counts, int32_t* rle_len, int cap, hipStream_t st);
int yp_rle_limits(int* max_src_h, int* max_out_h);
hipError_t yp_launch_loss_masks(int batch, int rows, int ph, int pw, int img_h, int img_w, int use_crop, const float* proto,
                                const float* coef, const float* boxes, const int32_t* num_rows, float* crop_ws, float* out_masks,
                                hipStream_t st);
hipError_t yp_launch_crop_edges(int total, int ph, int pw, int img_h, int img_w, int use_crop, const float* boxes, float* crop_ws,
                                hipStream_t st);
size_t yp_loss_grads_workspace_floats(int batch, int rows, int ph);
hipError_t yp_launch_loss_grads(int batch, int rows, int ph, int pw, int img_h, int img_w, int use_crop, const float* proto,
                                const float* coef, const float* boxes, const int32_t* num_rows, const float* masks,
                                const float* grad_masks, float* workspace, float* grad_proto, float* grad_coef, hipStream_t st);
