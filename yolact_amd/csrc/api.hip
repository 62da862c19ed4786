// C ABI of libyolact_post.so (declared in include/yolact_post.h).  Host side only: argument
// validation, scratch carving, stream-ordered launches.  No CPU compute path exists.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <thread>
#include <vector>
#include <new>

#include "common.hpp"

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define YP_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) return fail(YP_EHIP, "%s failed: %s", #call, hipGetErrorString(e_));    \
  } while (0)

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

extern "C" const char* yp_last_error(void) { return g_err; }
extern "C" int yp_version(void) { return YP_VERSION; }

// Every entry point leaves the calling thread's current HIP device as it found it.
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) err = hipSetDevice(dev); else prev = -1;   // nothing to restore
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define YP_ON_DEVICE(dev)                                                                          \
  DeviceGuard guard_(dev);                                                                         \
  if (guard_.err != hipSuccess) return fail(YP_EHIP, "hipSetDevice(%d) failed: %s", (dev), hipGetErrorString(guard_.err))

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}

// Where the handle's buffers sit inside one scratch allocation of `total` bytes.
struct ScratchLayout {
  size_t lkeys, lbox, okeys, pool, pool_box, crop, sel_anchor, cnt, n_cnt, max_items, total;
};

// Detect.__init__'s validation (detection.py:17-18) plus the limits of the kernels; fills the
// host-side plan of `h` (no device memory is touched apart from attribute queries).
static int plan_handle(const yp_config& c, yp_handle* h, int kpa_bound = 0) {
  if (c.nms_thresh <= 0.0f) return fail(YP_EINVAL, "nms_threshold must be non negative.");   // detection.py:17-18
  if (c.num_anchors < 1 || c.num_anchors > (1 << 24)) return fail(YP_EINVAL, "num_anchors must be in [1, 2^24]");
  if (c.num_classes < 2 || c.num_classes > 256) return fail(YP_EINVAL, "num_classes (incl. background) must be in [2, 256]");
  if (c.mask_dim != YP_MASK_DIM) return fail(YP_EINVAL, "mask_dim must be %d", YP_MASK_DIM);
  if (c.proto_h < 1 || c.proto_w < 1 || c.img_h < 1 || c.img_w < 1) return fail(YP_EINVAL, "bad proto/image size");
  if (c.nms_mode != YP_NMS_CROSS_CLASS && c.nms_mode != YP_NMS_PER_CLASS) return fail(YP_EINVAL, "bad nms_mode");
  if (c.top_k < 1 || c.top_k > YP_SORT_MAX) return fail(YP_EINVAL, "top_k must be in [1, %d]", YP_SORT_MAX);
  if (c.max_dets < 1) return fail(YP_EINVAL, "max_dets must be >= 1");
  if (c.nms_mode == YP_NMS_PER_CLASS && c.max_dets > YP_SORT_MAX)
    return fail(YP_EINVAL, "per-class mode: max_dets must be <= %d", YP_SORT_MAX);
  if (c.max_batch < 1) return fail(YP_EINVAL, "max_batch must be >= 1");
  if (!(c.conf_thresh >= 0.0f && c.conf_thresh < 1.0f)) return fail(YP_EINVAL, "conf_thresh must be in [0, 1)");
  // mask_assemble addresses a detection's row as a 32-bit byte offset from the image's first row
  const long long P = (long long)c.proto_h * c.proto_w;
  const long long rows = c.max_dets < YP_SORT_MAX ? c.max_dets : YP_SORT_MAX;
  if (P > (1ll << 28) || rows * P * 4 + 4096 >= (1ll << 32))
    return fail(YP_EINVAL, "proto_h*proto_w = %lld with %lld detection rows exceeds the 4 GiB of mask rows per image the kernels address",
                P, rows);
  if ((long long)c.max_batch * c.num_anchors >= (1ll << 31))
    return fail(YP_EINVAL, "max_batch*num_anchors must be below 2^31");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(YP_ENODEV, "no HIP device available");
  if (c.device_id < 0 || c.device_id >= ndev) return fail(YP_ENODEV, "device_id %d out of range (%d devices)", c.device_id, ndev);

  YP_ON_DEVICE(c.device_id);   // the occupancy / attribute queries below are about this device
  memset(h, 0, sizeof(*h));
  h->cfg = c;
  h->tune.mask_nt = env_int("YP_TUNE_MASK_NT", 0);
  h->tune.sf_waves = env_int("YP_TUNE_SF_WAVES", 0);
  h->tune.prefill = env_int("YP_TUNE_PREFILL", -1);
  h->nseg = c.nms_mode == YP_NMS_CROSS_CLASS ? 1 : c.num_classes - 1;
  h->sel_rows = c.max_dets < YP_SORT_MAX ? c.max_dets : YP_SORT_MAX;
  h->lds_stride = (c.num_classes & 1) ? c.num_classes : c.num_classes + 1;   // odd stride: conflict-free rows
  {
    // p_fgmax <= 1/(1+exp(x_bg-x_fgmax)) > conf  <=>  x_bg-x_fgmax < ln(1/conf-1); +1e-3 covers rounding
    const double cf = (double)c.conf_thresh;
    h->pre_thresh = cf > 0.0 ? (float)(log(1.0 / cf - 1.0) + 1e-3) : INFINITY;
    // Second reject (score_filter.hip, phase 1): R >= 1/conf means p_fgmax <= conf.  Margins: 1e-3 relative
    // for the hardware exp2, the products and the <= 1e-5 by which the computed probability can exceed the
    // true one; an absolute one for the float sum behind the mean of the n = C-2 other logits: its error
    // is at most (n-1) 2^-24 (xf - mean + |xf|), which moves the term n exp(mean - xf) by less than
    // n^2 2^-24 (0.37 + |xf|); the kernel applies the test for |xf| <= 128 only, and the margin is twice that.
    const double n = (double)c.num_classes - 2.0;
    h->jen_thr = (cf > 0.0 && cf < 1.0 && n >= 1.0) ? (float)((1.0 / cf) * (1.0 + 1e-3) + 2.0 * n * n * 0x1p-24 * 128.37) : NAN;
  }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c.device_id) != hipSuccess || cus < 1) cus = 256;
    h->num_cus = cus;
  }
  h->sf_bufb = (YP_SF_TA * h->lds_stride * (int)sizeof(float) + 15) / 16 * 16;
  h->sf_waves_per_cu = yp_score_filter_waves_per_cu(h);
  {
    // Sum of probabilities <= 1 (+ rounding), so fewer than 1/conf classes of one anchor exceed conf.
    int kpa = 1;
    if (c.nms_mode == YP_NMS_PER_CLASS) {
      kpa = c.num_classes - 1;
      if (c.conf_thresh > 0.0f) {
        const double lim = floor(1.0 / (double)c.conf_thresh) + 1.0;
        if (lim < (double)kpa) kpa = (int)lim;
      }
    }
    h->keys_worst = kpa;
    if (kpa_bound > 0 && kpa_bound < kpa) kpa = kpa_bound;   // yp_create_bounded: shorter lists, overflow reported
    h->keys_per_anchor = kpa;
  }
  return YP_OK;
}

// Scratch a planned handle needs to run batches of up to `max_batch` images.
static ScratchLayout scratch_layout(const yp_handle* h, int max_batch) {
  const bool per_class = h->cfg.nms_mode == YP_NMS_PER_CLASS;
  const size_t B = (size_t)max_batch;
  size_t list_entries = 0, ovf_keys = 0, max_items = 0;   // the largest any batch size up to max_batch needs
  for (int bs = 1; bs <= max_batch; ++bs) {
    int parts, longest;
    long long lcap, ocap;
    yp_plan_items(h, bs, &parts, &longest);
    yp_plan_lists(h, bs, &lcap, &ocap);
    const size_t items = (size_t)bs * parts;
    if (items > max_items) max_items = items;
    if ((size_t)bs * YP_NLISTS * (size_t)lcap > list_entries) list_entries = (size_t)bs * YP_NLISTS * (size_t)lcap;
    if ((size_t)bs * (size_t)ocap > ovf_keys) ovf_keys = (size_t)bs * (size_t)ocap;
  }
  int ngroups, gcap;
  yp_plan_groups(h, &ngroups, &gcap);
  const size_t pool_per_img = per_class ? (size_t)ngroups * gcap : 0;
  ScratchLayout L;
  size_t off = 0;
  L.lkeys = off;      off = align_up(off + list_entries * sizeof(u64), 256);
  L.lbox = off;       off = align_up(off + list_entries * sizeof(float4), 256);
  L.okeys = off;      off = align_up(off + ovf_keys * sizeof(u64), 256);
  L.pool = off;       off = align_up(off + B * pool_per_img * sizeof(u64), 256);
  L.pool_box = off;   off = align_up(off + B * pool_per_img * sizeof(float4), 256);
  L.crop = off;       off = align_up(off + B * h->sel_rows * sizeof(float4), 256);
  L.sel_anchor = off; off = align_up(off + B * h->sel_rows * sizeof(int32_t), 256);
  L.cnt = off;   // all counters contiguous: lcount | ocount | seg_surv | pool_count | stat_surv | arrive | overflow
  L.max_items = max_items;
  L.n_cnt = B * YP_NLISTS * YP_CNT_STRIDE + B * YP_CNT_STRIDE + max_items + B * YP_NLISTS + 2 * B + 1;
  off = align_up(off + L.n_cnt * sizeof(int32_t), 256);
  L.total = off;
  return L;
}

static void carve_scratch(yp_handle* h, const ScratchLayout& L, void* base_ptr, int max_batch) {
  char* base = (char*)base_ptr;
  h->s.lkeys = (u64*)(base + L.lkeys);
  h->s.lbox = (float4*)(base + L.lbox);
  h->s.okeys = (u64*)(base + L.okeys);
  h->s.pool = (u64*)(base + L.pool);
  h->s.pool_box = (float4*)(base + L.pool_box);
  h->s.crop = (float4*)(base + L.crop);
  h->s.sel_anchor = (int32_t*)(base + L.sel_anchor);
  h->s.lcount = (int32_t*)(base + L.cnt);                                   // (256-byte aligned: a line per counter)
  h->s.ocount = h->s.lcount + (size_t)max_batch * YP_NLISTS * YP_CNT_STRIDE;
  h->s.seg_surv = h->s.ocount + (size_t)max_batch * YP_CNT_STRIDE;
  h->s.pool_count = h->s.seg_surv + L.max_items;
  h->s.stat_surv = h->s.pool_count + (size_t)max_batch * YP_NLISTS;
  h->s.arrive = h->s.stat_surv + max_batch;
  h->s.overflow = h->s.arrive + max_batch;
}

static int create_handle(const yp_config* cfg, int kpa_bound, yp_handle** out) {
  if (!cfg || !out) return fail(YP_EINVAL, "yp_create: null argument");
  *out = nullptr;
  yp_handle* h = new (std::nothrow) yp_handle();
  if (!h) return fail(YP_ENOMEM, "host allocation failed");
  const int rc = plan_handle(*cfg, h, kpa_bound);
  if (rc != YP_OK) { delete h; return rc; }
  DeviceGuard guard_(cfg->device_id);
  if (guard_.err != hipSuccess) { delete h; return fail(YP_EHIP, "hipSetDevice(%d) failed", cfg->device_id); }
  const ScratchLayout L = scratch_layout(h, cfg->max_batch);
  h->scratch_bytes = (int64_t)L.total;
  if (hipMalloc(&h->scratch_base, L.total) != hipSuccess) {
    delete h;
    return fail(YP_ENOMEM, "hipMalloc of %zu scratch bytes failed", L.total);
  }
  carve_scratch(h, L, h->scratch_base, cfg->max_batch);
  if (hipMemset(h->s.lcount, 0, L.n_cnt * sizeof(int32_t)) != hipSuccess) {   // (the list and arrival counters must start at zero)
    (void)hipFree(h->scratch_base);
    delete h;
    return fail(YP_EHIP, "hipMemset of the counters failed");
  }
  (void)hipDeviceSynchronize();
#ifdef YP_STAMPS
  (void)hipMalloc(&h->dbg, 3 * 4096 * 16 * 8); (void)hipMemset(h->dbg, 0, 3 * 4096 * 16 * 8);
  (void)hipMemset((char*)h->dbg + (size_t)2 * 4096 * 16 * 8, 0xff, 8);   // fill start: running minimum
#endif
  *out = h;
  g_err[0] = 0;
  return YP_OK;
}

extern "C" int yp_create(const yp_config* cfg, yp_handle** out) { return create_handle(cfg, 0, out); }

extern "C" int yp_create_bounded(const yp_config* cfg, int32_t keys_per_anchor, yp_handle** out) {
  if (keys_per_anchor < 1) return fail(YP_EINVAL, "yp_create_bounded: keys_per_anchor must be >= 1");
  return create_handle(cfg, keys_per_anchor, out);
}

extern "C" int yp_overflowed(yp_handle* h) {
  if (!h) return fail(YP_EINVAL, "yp_overflowed: null handle");
  YP_ON_DEVICE(h->cfg.device_id);
  YP_HIP(hipStreamSynchronize(h->last_stream));
  int32_t flag = 0;
  YP_HIP(hipMemcpy(&flag, h->s.overflow, sizeof(flag), hipMemcpyDeviceToHost));
  if (flag) YP_HIP(hipMemset(h->s.overflow, 0, sizeof(flag)));
  return flag ? 1 : 0;
}

extern "C" int yp_destroy(yp_handle* h) {
  if (!h) return YP_OK;
  DeviceGuard guard_(h->cfg.device_id);
  (void)hipDeviceSynchronize();
  if (h->profile) for (int i = 0; i <= YP_NUM_KERNELS; ++i) (void)hipEventDestroy(h->ev[i]);
  (void)hipFree(h->scratch_base);
  delete h;
  return YP_OK;
}

static int check_run_args(const yp_handle* h, const float* pred_offset, const float* pred_cls, const float* pred_mask_coef,
                          const float* priors, const float* proto, float* det_boxes, float* det_classes, float* det_scores,
                          float* det_masks, uint8_t* det_bitmaps, int32_t* num_det) {
  if (!pred_offset || !pred_cls || !pred_mask_coef || !priors || !det_boxes || !det_classes || !det_scores || !num_det)
    return fail(YP_EINVAL, "yp_run: null tensor pointer");
  if (det_masks && !proto) return fail(YP_EINVAL, "yp_run: det_masks given without proto");
  if (det_bitmaps && (!det_masks || h->cfg.mask_threshold < 0.0f))
    return fail(YP_EINVAL, "yp_run: det_bitmaps needs det_masks and mask_threshold >= 0");
  if ((((uintptr_t)pred_offset | (uintptr_t)priors | (uintptr_t)pred_mask_coef | (uintptr_t)det_boxes | (uintptr_t)proto) & 15u) != 0)
    return fail(YP_EINVAL, "yp_run: pred_offset, priors, pred_mask_coef, proto and det_boxes must be 16-byte aligned");
  return YP_OK;
}

// The launches of one batch: score_filter -> NMS (+ the mask zero-fill as extra workgroups of the same
// launch: it depends on nothing the NMS computes) -> mask_assemble.
static int run_batch(yp_handle* h, int batch, const float* pred_offset, const float* pred_cls, const float* pred_mask_coef,
                     const float* priors, const float* proto, float* det_boxes, float* det_classes, float* det_scores,
                     float* det_masks, uint8_t* det_bitmaps, int32_t* num_det, int32_t* kept_anchor_idx, hipStream_t st) {
  const bool prof = h->profile != 0;
  const bool prefill = det_masks && yp_mask_prefill(h, batch);
  if (prof) YP_HIP(hipEventRecord(h->ev[0], st));
  YP_HIP(yp_launch_score_filter(h, batch, pred_cls, pred_offset, priors, st));
  if (prof) YP_HIP(hipEventRecord(h->ev[1], st));
  YP_HIP(yp_launch_nms(h, batch, pred_offset, priors, det_boxes, det_classes, det_scores, num_det, kept_anchor_idx,
                       prefill ? det_masks : nullptr, prefill ? det_bitmaps : nullptr, yp_mask_fill_rows(h), st));
  if (prof) YP_HIP(hipEventRecord(h->ev[2], st));
  if (det_masks) YP_HIP(yp_launch_mask_assemble(h, batch, proto, pred_mask_coef, num_det, det_masks, det_bitmaps, st));
  if (prof) YP_HIP(hipEventRecord(h->ev[3], st));
  return YP_OK;
}

extern "C" int yp_run(yp_handle* h, int32_t batch, const float* pred_offset, const float* pred_cls,
                      const float* pred_mask_coef, const float* priors, const float* proto, float* det_boxes,
                      float* det_classes, float* det_scores, float* det_masks, uint8_t* det_bitmaps,
                      int32_t* num_det, int32_t* kept_anchor_idx, void* hip_stream) {
  if (!h) return fail(YP_EINVAL, "yp_run: null handle");
  if (batch < 0 || batch > h->cfg.max_batch) return fail(YP_EINVAL, "batch %d outside [0, max_batch=%d]", batch, h->cfg.max_batch);
  const int rc = check_run_args(h, pred_offset, pred_cls, pred_mask_coef, priors, proto, det_boxes, det_classes, det_scores,
                                det_masks, det_bitmaps, num_det);
  if (rc != YP_OK) return rc;
  if (batch == 0) return YP_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  YP_ON_DEVICE(h->cfg.device_id);
  h->last_stream = st;
  h->last_batch = batch;
  return run_batch(h, batch, pred_offset, pred_cls, pred_mask_coef, priors, proto, det_boxes, det_classes, det_scores,
                   det_masks, det_bitmaps, num_det, kept_anchor_idx, st);
}

// ---- output sets the library keeps track of ----------------------------------------------------------
// The record of one set of mask buffers: two slots of (crop edges per row, count per image), the one the last call
// wrote and the one the next call writes; `side` says which is which.
struct yp_kept {
  int device;
  int max_batch, sel_rows;
  void* base;
  float4* crop[2];
  int32_t* num[2];
  int side;              // the slot the NEXT call writes
  bool fresh;            // nothing recorded yet: the next call refills the buffers
  int batch;             // bound at the first call
  const void* masks;     // ... as are the buffers
  const void* bitmaps;
};

extern "C" int yp_kept_create(yp_handle* h, yp_kept** out) {
  if (!h || !out) return fail(YP_EINVAL, "yp_kept_create: null argument");
  *out = nullptr;
  yp_kept* k = new (std::nothrow) yp_kept();
  if (!k) return fail(YP_ENOMEM, "host allocation failed");
  k->device = h->cfg.device_id; k->max_batch = h->cfg.max_batch; k->sel_rows = h->sel_rows;
  YP_ON_DEVICE(k->device);
  const size_t crop_bytes = align_up((size_t)k->max_batch * k->sel_rows * sizeof(float4), 256);
  const size_t num_bytes = align_up((size_t)k->max_batch * sizeof(int32_t), 256);
  const size_t total = 2 * (crop_bytes + num_bytes);
  if (hipMalloc(&k->base, total) != hipSuccess) { delete k; return fail(YP_ENOMEM, "hipMalloc of %zu bytes failed", total); }
  if (hipMemset(k->base, 0, total) != hipSuccess) { (void)hipFree(k->base); delete k; return fail(YP_EHIP, "hipMemset failed"); }
  char* b = (char*)k->base;
  k->crop[0] = (float4*)b; k->crop[1] = (float4*)(b + crop_bytes);
  k->num[0] = (int32_t*)(b + 2 * crop_bytes); k->num[1] = (int32_t*)(b + 2 * crop_bytes + num_bytes);
  k->side = 0; k->fresh = true; k->batch = 0; k->masks = nullptr; k->bitmaps = nullptr;
  *out = k;
  return YP_OK;
}

extern "C" int yp_kept_destroy(yp_kept* k) {
  if (!k) return YP_OK;
  YP_ON_DEVICE(k->device);
  (void)hipFree(k->base);
  delete k;
  return YP_OK;
}

extern "C" int yp_kept_reset(yp_kept* k) {
  if (!k) return fail(YP_EINVAL, "yp_kept_reset: null argument");
  k->fresh = true;
  return YP_OK;
}

extern "C" int yp_run_kept(yp_handle* h, yp_kept* k, int32_t batch, const float* pred_offset, const float* pred_cls,
                           const float* pred_mask_coef, const float* priors, const float* proto, float* det_boxes,
                           float* det_classes, float* det_scores, float* det_masks, uint8_t* det_bitmaps,
                           int32_t* num_det, int32_t* kept_anchor_idx, void* hip_stream) {
  if (!h || !k) return fail(YP_EINVAL, "yp_run_kept: null handle");
  if (k->device != h->cfg.device_id || k->max_batch != h->cfg.max_batch || k->sel_rows != h->sel_rows)
    return fail(YP_EINVAL, "yp_run_kept: this yp_kept was made for another handle's configuration");
  if (batch < 0 || batch > h->cfg.max_batch) return fail(YP_EINVAL, "batch %d outside [0, max_batch=%d]", batch, h->cfg.max_batch);
  if (!det_masks) return fail(YP_EINVAL, "yp_run_kept: det_masks must not be NULL (without masks there is nothing to keep track of)");
  const int rc = check_run_args(h, pred_offset, pred_cls, pred_mask_coef, priors, proto, det_boxes, det_classes, det_scores,
                                det_masks, det_bitmaps, num_det);
  if (rc != YP_OK) return rc;
  if (batch == 0) return YP_OK;
  if (k->fresh && k->masks == nullptr) { k->masks = det_masks; k->bitmaps = det_bitmaps; k->batch = batch; }
  if (k->masks != det_masks || k->bitmaps != det_bitmaps)
    return fail(YP_EINVAL, "yp_run_kept: a yp_kept belongs to the mask buffers of its first call");
  if (k->batch != batch) return fail(YP_EINVAL, "yp_run_kept: batch %d, but this yp_kept is bound to batch %d", batch, k->batch);
  hipStream_t st = (hipStream_t)hip_stream;
  YP_ON_DEVICE(h->cfg.device_id);
  h->last_stream = st;
  h->last_batch = batch;
  const bool prof = h->profile != 0;
  const int nw = k->side, od = nw ^ 1;
  if (k->fresh)   // nothing is known about the buffers: the record says "no detections" and the launch below refills
    YP_HIP(hipMemsetAsync(k->num[od], 0, (size_t)batch * sizeof(int32_t), st));
  if (prof) YP_HIP(hipEventRecord(h->ev[0], st));
  YP_HIP(yp_launch_score_filter(h, batch, pred_cls, pred_offset, priors, st));
  if (prof) YP_HIP(hipEventRecord(h->ev[1], st));
  YP_HIP(yp_launch_nms(h, batch, pred_offset, priors, det_boxes, det_classes, det_scores, num_det, kept_anchor_idx,
                       k->fresh ? det_masks : nullptr, k->fresh ? det_bitmaps : nullptr, yp_mask_fill_rows(h), st, k->crop[nw]));
  if (prof) YP_HIP(hipEventRecord(h->ev[2], st));
  const yp_track trk{k->crop[nw], k->crop[od], k->num[od], k->num[nw]};
  YP_HIP(yp_launch_mask_assemble(h, batch, proto, pred_mask_coef, num_det, det_masks, det_bitmaps, st, &trk));
  if (prof) YP_HIP(hipEventRecord(h->ev[3], st));
  k->side = od;
  k->fresh = false;
  return YP_OK;
}

// ---- the allocation-free entry point ----------------------------------------------------------------
// Scratch of yp_run_static: one array per GPU inside the code object (every device that loads the
// library holds its own copy, at its own address, until the process ends).  Host code never uses
// the array's NAME as an address: g_static_ptr[dev] is what hipGetSymbolAddress returned with that
// device current, and only that pointer is handed to kernels and memsets.
static __device__ __align__(256) unsigned char scratch[YP_STATIC_SCRATCH_BYTES];

#define YP_STATIC_MAX_DEVICES 64
static std::mutex g_static_mu;
static unsigned char* g_static_ptr[YP_STATIC_MAX_DEVICES];    // per device; null = not resolved yet
#define YP_STATIC_PLANS 4                                     // configs remembered per device (round-robin replacement)
static yp_handle g_static_plan[YP_STATIC_MAX_DEVICES][YP_STATIC_PLANS];   // host-side plans of the last configs seen per device
static bool g_static_plan_valid[YP_STATIC_MAX_DEVICES][YP_STATIC_PLANS];
static int g_static_plan_next[YP_STATIC_MAX_DEVICES];

extern "C" int yp_run_static(const yp_config* cfg, int32_t batch, const float* pred_offset, const float* pred_cls,
                             const float* pred_mask_coef, const float* priors, const float* proto, float* det_boxes,
                             float* det_classes, float* det_scores, float* det_masks, uint8_t* det_bitmaps,
                             int32_t* num_det, int32_t* kept_anchor_idx, void* hip_stream) {
  if (!cfg) return fail(YP_EINVAL, "yp_run_static: null config");
  if (batch < 0) return fail(YP_EINVAL, "yp_run_static: negative batch");
  if (cfg->device_id < 0 || cfg->device_id >= YP_STATIC_MAX_DEVICES)
    return fail(YP_ENODEV, "yp_run_static: device_id %d outside [0, %d)", cfg->device_id, YP_STATIC_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_static_mu);   // one call at a time plans and launches (calls that OVERLAP on a GPU race on the scratch anyway)
  const int dev = cfg->device_id;
  yp_config want = *cfg;
  want.max_batch = 1;   // not part of this entry point's contract: any batch is cut to fit
  // The plan of a (device, config) pair is made on the FIRST call with it - validation, attribute and occupancy
  // queries, and once per device hipGetSymbolAddress: that call must be made eagerly, outside any graph capture -
  // and remembered (a few configs per device, so callers that alternate configs do not re-plan every call).
  yp_handle* h = nullptr;
  for (int i = 0; i < YP_STATIC_PLANS; ++i)
    if (g_static_plan_valid[dev][i] && memcmp(&g_static_plan[dev][i].cfg, &want, sizeof(want)) == 0) h = &g_static_plan[dev][i];
  if (!h) {
    const int slot = g_static_plan_next[dev];
    g_static_plan_valid[dev][slot] = false;
    const int rc = plan_handle(want, &g_static_plan[dev][slot]);   // validation + host-side plan; queries only, nothing allocated
    if (rc != YP_OK) return rc;
    g_static_plan_valid[dev][slot] = true;
    g_static_plan_next[dev] = (slot + 1) % YP_STATIC_PLANS;
    h = &g_static_plan[dev][slot];
  }
  const int rc = check_run_args(h, pred_offset, pred_cls, pred_mask_coef, priors, proto, det_boxes, det_classes, det_scores,
                                det_masks, det_bitmaps, num_det);
  if (rc != YP_OK) return rc;
  if (batch == 0) return YP_OK;
  YP_ON_DEVICE(dev);
  if (!g_static_ptr[dev]) {   // once per device, with that device current
    unsigned char* ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) return fail(YP_EHIP, "hipGetSymbolAddress(scratch) failed on device %d: %s", dev, hipGetErrorString(e));
    g_static_ptr[dev] = ptr;
  }
  unsigned char* const base = g_static_ptr[dev];
  // the largest chunk of images whose scratch fits the array, then equal chunks
  const size_t one = scratch_layout(h, 1).total;
  if (one > (size_t)YP_STATIC_SCRATCH_BYTES)
    return fail(YP_EINVAL, "yp_run_static: one image of this configuration needs %zu bytes of scratch, the static scratch holds %zu "
                           "(YP_STATIC_SCRATCH_BYTES); use yp_create / yp_run", one, (size_t)YP_STATIC_SCRATCH_BYTES);
  int fit = 1;
  for (int lo = 1, hi = batch; lo <= hi;) {   // scratch grows with the batch: binary search
    const int mid = lo + (hi - lo) / 2;
    if (scratch_layout(h, mid).total <= (size_t)YP_STATIC_SCRATCH_BYTES) { fit = mid; lo = mid + 1; } else hi = mid - 1;
  }
  const int nchunks = (batch + fit - 1) / fit;
  const int per = (batch + nchunks - 1) / nchunks;
  const yp_config& c = h->cfg;
  const size_t A = (size_t)c.num_anchors, C = (size_t)c.num_classes, P = (size_t)c.proto_h * c.proto_w, R = (size_t)c.max_dets;
  hipStream_t st = (hipStream_t)hip_stream;
  for (int lo = 0; lo < batch; lo += per) {
    const int nb = batch - lo < per ? batch - lo : per;
    const ScratchLayout L = scratch_layout(h, nb);
    if (L.total > (size_t)YP_STATIC_SCRATCH_BYTES) return fail(YP_EINVAL, "yp_run_static: internal: chunk of %d images does not fit", nb);
    carve_scratch(h, L, base, nb);
    h->cfg.max_batch = nb;
    // the array keeps the previous call's (or chunk's) contents: the counters (the images' arrival
    // counters among them) start from zero
    YP_HIP(hipMemsetAsync(h->s.lcount, 0, L.n_cnt * sizeof(int32_t), st));
    const int rcb = run_batch(h, nb, pred_offset + lo * A * 4, pred_cls + lo * A * C, pred_mask_coef + lo * A * YP_MASK_DIM, priors,
                              proto ? proto + lo * P * YP_MASK_DIM : nullptr, det_boxes + lo * R * 4, det_classes + lo * R,
                              det_scores + lo * R, det_masks ? det_masks + lo * R * P : nullptr,
                              det_bitmaps ? det_bitmaps + lo * R * P : nullptr, num_det + lo,
                              kept_anchor_idx ? kept_anchor_idx + lo * R : nullptr, st);
    h->cfg.max_batch = 1;
    if (rcb != YP_OK) return rcb;
  }
  return YP_OK;
}

extern "C" int yp_sync(yp_handle* h) {
  if (!h) return fail(YP_EINVAL, "yp_sync: null handle");
  YP_ON_DEVICE(h->cfg.device_id);
  YP_HIP(hipStreamSynchronize(h->last_stream));
  return YP_OK;
}

extern "C" int yp_last_survivors(yp_handle* h, int32_t* host_out, int32_t batch) {
  if (!h || !host_out) return fail(YP_EINVAL, "yp_last_survivors: null argument");
  if (batch < 0 || batch > h->cfg.max_batch) return fail(YP_EINVAL, "bad batch");
  YP_ON_DEVICE(h->cfg.device_id);
  YP_HIP(hipStreamSynchronize(h->last_stream));
  YP_HIP(hipMemcpy(host_out, h->s.stat_surv, sizeof(int32_t) * (size_t)batch, hipMemcpyDeviceToHost));
  return YP_OK;
}

extern "C" int yp_profile_enable(yp_handle* h, int32_t on) {
  if (!h) return fail(YP_EINVAL, "yp_profile_enable: null handle");
  YP_ON_DEVICE(h->cfg.device_id);
  if (on && !h->profile) {
    for (int i = 0; i <= YP_NUM_KERNELS; ++i) YP_HIP(hipEventCreate(&h->ev[i]));
    h->profile = 1;
  } else if (!on && h->profile) {
    YP_HIP(hipStreamSynchronize(h->last_stream));
    for (int i = 0; i <= YP_NUM_KERNELS; ++i) (void)hipEventDestroy(h->ev[i]);
    h->profile = 0;
  }
  return YP_OK;
}

extern "C" int yp_profile_read(yp_handle* h, float* ms_out) {
  if (!h || !ms_out) return fail(YP_EINVAL, "yp_profile_read: null argument");
  if (!h->profile) return fail(YP_EINVAL, "profiling is not enabled");
  YP_ON_DEVICE(h->cfg.device_id);
  YP_HIP(hipEventSynchronize(h->ev[YP_NUM_KERNELS]));
  for (int i = 0; i < YP_NUM_KERNELS; ++i) YP_HIP(hipEventElapsedTime(&ms_out[i], h->ev[i], h->ev[i + 1]));
  return YP_OK;
}

#ifdef YP_STAMPS
// diagnostic build only: copies the [3][4096][16] stamp table to the host
extern "C" int yp_debug_stamps(yp_handle* h, unsigned long long* host_out) {
  if (!h || !host_out) return YP_EINVAL;
  (void)hipDeviceSynchronize();
  return hipMemcpy(host_out, h->dbg, 3 * 4096 * 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? YP_OK : YP_EHIP;
}
#endif

extern "C" int64_t yp_make_priors(int32_t img_h, int32_t img_w, int32_t num_levels, const int32_t* fmap_hw, const double* scales,
                                  int32_t num_ratios, const double* aspect_ratios, float* priors, int64_t capacity,
                                  int32_t device_id, void* hip_stream) {
  if (!fmap_hw || !scales || !aspect_ratios) return fail(YP_EINVAL, "yp_make_priors: null argument");
  if (img_h < 1 || img_w < 1) return fail(YP_EINVAL, "yp_make_priors: bad image size");
  const long long n = yp_priors_count(num_levels, fmap_hw, num_ratios);
  if (n < 0) return fail(YP_EINVAL, "yp_make_priors: 1..8 levels with positive sizes and 1..8 aspect ratios are supported");
  double sq[8];
  for (int k = 0; k < num_ratios; ++k) {
    if (!(aspect_ratios[k] > 0.0)) return fail(YP_EINVAL, "yp_make_priors: aspect ratios must be positive");
    sq[k] = sqrt(aspect_ratios[k]);                                    // anchor.py:40
  }
  if (!priors) return n;                                               // size query
  if (capacity < n) return fail(YP_EINVAL, "yp_make_priors: %lld anchors do not fit a buffer of %lld", n, (long long)capacity);
  if ((((uintptr_t)priors) & 15u) != 0) return fail(YP_EINVAL, "yp_make_priors: priors must be 16-byte aligned");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(YP_ENODEV, "no HIP device available");
  if (device_id < 0 || device_id >= ndev) return fail(YP_ENODEV, "device_id %d out of range (%d devices)", device_id, ndev);
  YP_ON_DEVICE(device_id);
  YP_HIP(yp_launch_make_priors(img_h, img_w, num_levels, fmap_hw, scales, num_ratios, sq, priors, (hipStream_t)hip_stream));
  return n;
}

// cv::resize replaces INTER_LINEAR by INTER_AREA when both axes shrink by exactly 2 (imgproc resize.cpp:
// `interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2`), and its 2x2 float average
// sums in an order that depends on the SIMD width of the build: not restated here, rejected instead.
static int rle_check_size(const char* who, int mask_h, int mask_w, int out_h, int out_w, int max_out_h) {
  if (out_h < 1 || out_w < 1) return fail(YP_EINVAL, "%s: output sizes must be positive", who);
  if (out_h > max_out_h || out_w > 65535) return fail(YP_EINVAL, "%s: out_h <= %d and out_w <= 65535 are supported", who, max_out_h);
  if (mask_h == 2 * out_h && mask_w == 2 * out_w)
    return fail(YP_EINVAL, "%s: %dx%d -> %dx%d halves both axes exactly: cv::resize switches to INTER_AREA there, which this library "
                           "does not restate", who, mask_h, mask_w, out_h, out_w);
  return YP_OK;
}

static int masks_to_rle(const char* who, int32_t device_id, int32_t batch, int32_t max_dets, int32_t mask_h, int32_t mask_w,
                        const float* det_masks, const int32_t* num_det, int32_t out_h, int32_t out_w, const int32_t* out_hw,
                        float threshold, int32_t order, uint8_t* bitmaps, uint32_t* rle_counts, int32_t* rle_len, int32_t rle_cap,
                        void* hip_stream) {
  if (!det_masks || !num_det || !rle_counts || !rle_len) return fail(YP_EINVAL, "%s: null pointer", who);
  if (batch < 0 || max_dets < 1 || mask_h < 1 || mask_w < 1 || rle_cap < 1) return fail(YP_EINVAL, "%s: sizes must be positive", who);
  if (order != YP_RESIZE_THEN_THRESHOLD && order != YP_THRESHOLD_THEN_RESIZE) return fail(YP_EINVAL, "%s: bad order", who);
  int max_src_h, max_out_h;
  yp_rle_limits(&max_src_h, &max_out_h);
  if (mask_h > max_src_h) return fail(YP_EINVAL, "%s: mask_h <= %d is supported", who, max_src_h);
  if (mask_w > 32767 || max_dets > 65535 || batch > 65535) return fail(YP_EINVAL, "%s: size out of range", who);
  if (out_hw) {
    if (bitmaps) return fail(YP_EINVAL, "%s: bitmaps are not written with per-image sizes", who);
    for (int b = 0; b < batch; ++b) {
      const int rc = rle_check_size(who, mask_h, mask_w, out_hw[2 * b], out_hw[2 * b + 1], max_out_h);
      if (rc != YP_OK) return rc;
    }
  } else {
    const int rc = rle_check_size(who, mask_h, mask_w, out_h, out_w, max_out_h);
    if (rc != YP_OK) return rc;
  }
  if (batch == 0) return YP_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(YP_ENODEV, "no HIP device available");
  if (device_id < 0 || device_id >= ndev) return fail(YP_ENODEV, "device_id %d out of range (%d devices)", device_id, ndev);
  YP_ON_DEVICE(device_id);
  YP_HIP(yp_launch_mask_rle(batch, max_dets, mask_h, mask_w, det_masks, num_det, out_h, out_w, out_hw, threshold, order, bitmaps,
                            rle_counts, rle_len, rle_cap, (hipStream_t)hip_stream));
  return YP_OK;
}

extern "C" int yp_masks_to_rle(int32_t device_id, int32_t batch, int32_t max_dets, int32_t mask_h, int32_t mask_w,
                               const float* det_masks, const int32_t* num_det, int32_t out_h, int32_t out_w, float threshold,
                               int32_t order, uint8_t* bitmaps, uint32_t* rle_counts, int32_t* rle_len, int32_t rle_cap,
                               void* hip_stream) {
  return masks_to_rle("yp_masks_to_rle", device_id, batch, max_dets, mask_h, mask_w, det_masks, num_det, out_h, out_w, nullptr,
                      threshold, order, bitmaps, rle_counts, rle_len, rle_cap, hip_stream);
}

extern "C" int yp_masks_to_rle_sized(int32_t device_id, int32_t batch, int32_t max_dets, int32_t mask_h, int32_t mask_w,
                                     const float* det_masks, const int32_t* num_det, const int32_t* out_hw, float threshold,
                                     int32_t order, uint32_t* rle_counts, int32_t* rle_len, int32_t rle_cap, void* hip_stream) {
  if (!out_hw) return fail(YP_EINVAL, "yp_masks_to_rle_sized: null out_hw");
  return masks_to_rle("yp_masks_to_rle_sized", device_id, batch, max_dets, mask_h, mask_w, det_masks, num_det, 0, 0, out_hw,
                      threshold, order, nullptr, rle_counts, rle_len, rle_cap, hip_stream);
}

extern "C" int yp_assemble_masks(int32_t device_id, int32_t batch, int32_t rows, int32_t proto_h, int32_t proto_w, int32_t img_h,
                                 int32_t img_w, int32_t use_cropped_mask, const float* proto, const float* coef, const float* boxes,
                                 const int32_t* num_rows, float* crop_workspace, float* out_masks, void* hip_stream) {
  if (!proto || !coef || !boxes || !num_rows || !crop_workspace || !out_masks) return fail(YP_EINVAL, "yp_assemble_masks: null pointer");
  if (batch < 0 || rows < 1 || rows > YP_SORT_MAX) return fail(YP_EINVAL, "yp_assemble_masks: rows must be in [1, %d]", YP_SORT_MAX);
  if (proto_h < 1 || proto_w < 1 || img_h < 1 || img_w < 1) return fail(YP_EINVAL, "yp_assemble_masks: bad proto/image size");
  const long long P = (long long)proto_h * proto_w;
  if (P > (1ll << 28) || (long long)rows * P * 4 + 4096 >= (1ll << 32))
    return fail(YP_EINVAL, "yp_assemble_masks: %d rows of %lld pixels exceed the 4 GiB of mask rows per image the kernel addresses", rows, P);
  if ((long long)batch * rows >= (1ll << 31)) return fail(YP_EINVAL, "yp_assemble_masks: batch*rows too large");
  if ((((uintptr_t)proto | (uintptr_t)coef | (uintptr_t)boxes | (uintptr_t)crop_workspace) & 15u) != 0)
    return fail(YP_EINVAL, "yp_assemble_masks: proto, coef, boxes and crop_workspace must be 16-byte aligned");
  if (batch == 0) return YP_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(YP_ENODEV, "no HIP device available");
  if (device_id < 0 || device_id >= ndev) return fail(YP_ENODEV, "device_id %d out of range (%d devices)", device_id, ndev);
  YP_ON_DEVICE(device_id);
  YP_HIP(yp_launch_loss_masks(batch, rows, proto_h, proto_w, img_h, img_w, use_cropped_mask, proto, coef, boxes, num_rows,
                              crop_workspace, out_masks, (hipStream_t)hip_stream));
  return YP_OK;
}

extern "C" int64_t yp_assemble_masks_backward_workspace(int32_t batch, int32_t rows, int32_t proto_h) {
  if (batch < 0 || rows < 1 || proto_h < 1) return fail(YP_EINVAL, "yp_assemble_masks_backward_workspace: bad argument");
  return (int64_t)(yp_loss_grads_workspace_floats(batch, rows, proto_h) * sizeof(float));
}

extern "C" int yp_assemble_masks_backward(int32_t device_id, int32_t batch, int32_t rows, int32_t proto_h, int32_t proto_w,
                                          int32_t img_h, int32_t img_w, int32_t use_cropped_mask, const float* proto,
                                          const float* coef, const float* boxes, const int32_t* num_rows, const float* masks,
                                          const float* grad_masks, float* workspace, float* grad_proto, float* grad_coef,
                                          void* hip_stream) {
  if (!proto || !coef || !boxes || !num_rows || !masks || !grad_masks || !workspace || !grad_proto || !grad_coef)
    return fail(YP_EINVAL, "yp_assemble_masks_backward: null pointer");
  if (batch < 0 || rows < 1 || rows > YP_SORT_MAX) return fail(YP_EINVAL, "yp_assemble_masks_backward: rows must be in [1, %d]", YP_SORT_MAX);
  if (proto_h < 1 || proto_w < 1 || img_h < 1 || img_w < 1) return fail(YP_EINVAL, "yp_assemble_masks_backward: bad proto/image size");
  // grad_coef keeps one prototype row in LDS: 160 bytes per pixel + 1 KiB, of the CU's 160 KiB.  Checked here,
  // before anything is launched (above ~1017 the third launch failed after two had already run).
  if (proto_w > 1000) return fail(YP_EINVAL, "yp_assemble_masks_backward: proto_w %d exceeds 1000 (a prototype row, 160 bytes per pixel, must fit in LDS)", proto_w);
  const long long P = (long long)proto_h * proto_w;
  if (P > (1ll << 26)) return fail(YP_EINVAL, "yp_assemble_masks_backward: %lld pixels per prototype exceed 2^26", P);
  if ((long long)batch * rows >= (1ll << 31) || (long long)batch * ((P + 255) / 256) >= (1ll << 31) || (long long)batch * proto_h >= (1ll << 31))
    return fail(YP_EINVAL, "yp_assemble_masks_backward: batch too large");
  if ((((uintptr_t)proto | (uintptr_t)coef | (uintptr_t)boxes | (uintptr_t)workspace | (uintptr_t)grad_proto) & 15u) != 0)
    return fail(YP_EINVAL, "yp_assemble_masks_backward: proto, coef, boxes, workspace and grad_proto must be 16-byte aligned");
  if (batch == 0) return YP_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(YP_ENODEV, "no HIP device available");
  if (device_id < 0 || device_id >= ndev) return fail(YP_ENODEV, "device_id %d out of range (%d devices)", device_id, ndev);
  YP_ON_DEVICE(device_id);
  YP_HIP(yp_launch_loss_grads(batch, rows, proto_h, proto_w, img_h, img_w, use_cropped_mask, proto, coef, boxes, num_rows, masks,
                              grad_masks, workspace, grad_proto, grad_coef, (hipStream_t)hip_stream));
  return YP_OK;
}

// pycocotools maskApi.c rleToString: every count (from the fourth on: its difference to the count two
// places before) as little-endian groups of 5 bits, bit 5 = "more groups follow", offset by '0'.
// Writes at most cap-1 characters (no terminator) and returns the full length.
static long long rle_encode(const uint32_t* counts, int n, char* out, long long cap) {
  long long len = 0;
  for (int i = 0; i < n; ++i) {
    long long x = (long long)counts[i];
    if (i > 2) x -= (long long)counts[i - 2];
    bool more = true;
    while (more) {
      long long c = x & 0x1f;
      x >>= 5;
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      c += 48;
      if (len < cap - 1) out[len] = (char)c;
      ++len;
    }
  }
  return len;
}

extern "C" int yp_rle_string(const uint32_t* counts, int32_t n, char* out, int32_t cap) {
  if (!counts || !out || n < 0 || cap < 1) return fail(YP_EINVAL, "yp_rle_string: bad argument");
  const long long len = rle_encode(counts, n, out, cap);
  out[len < cap - 1 ? len : cap - 1] = 0;
  return (int)len;
}

extern "C" int64_t yp_rle_strings(const uint32_t* counts, const int64_t* starts, const int32_t* lens, int32_t n_lists,
                                  int64_t n_counts, char* out, int64_t out_cap, int64_t* offsets) {
  if (!counts || !starts || !lens || !out || !offsets || n_lists < 0 || n_counts < 0 || out_cap < 1)
    return fail(YP_EINVAL, "yp_rle_strings: bad argument");
  int64_t total = 0;
  for (int i = 0; i < n_lists; ++i) {
    if (lens[i] < 0 || starts[i] < 0 || starts[i] + lens[i] > n_counts)
      return fail(YP_EINVAL, "yp_rle_strings: list %d (start %lld, %d counts) leaves the array of %lld counts", i, (long long)starts[i], lens[i],
                  (long long)n_counts);
    total += lens[i];
  }
  // A batch's worth of lists (config 2: 3200 lists, 0.9 M counts, 5 ms on one core) is cut into runs of lists with
  // about equal numbers of counts, one host thread each: first every string's length, then - the offsets known -
  // the strings themselves.  Same bytes as the one-thread loop, truncation of a full buffer included.
  unsigned hw = std::thread::hardware_concurrency();
  int nthreads = (int)(hw == 0 ? 1 : (hw > 8 ? 8 : hw));
  if (total < 65536 || n_lists < 2 * nthreads) nthreads = 1;
  std::vector<int> cut(nthreads + 1, n_lists);
  cut[0] = 0;
  {
    int64_t acc = 0;
    int k = 1;
    for (int i = 0; i < n_lists && k < nthreads; ++i) {
      acc += lens[i];
      if (acc * nthreads >= total * k) cut[k++] = i + 1;
    }
  }
  std::vector<long long> slen((size_t)n_lists);
  auto run = [&](auto&& body) {
    if (nthreads == 1) { body(0, n_lists); return; }
    std::vector<std::thread> th;
    int started = 0;
    try {                                       // (no exception may leave a C entry point)
      for (; started < nthreads; ++started) th.emplace_back(body, cut[started], cut[started + 1]);
    } catch (...) {
    }
    for (auto& t : th) t.join();
    if (started < nthreads) body(cut[started], n_lists);   // the runs no thread could be started for: here
  };
  char dummy;
  run([&](int lo, int hi) { for (int i = lo; i < hi; ++i) slen[i] = rle_encode(counts + starts[i], lens[i], &dummy, 1); });
  int64_t pos = 0;
  for (int i = 0; i < n_lists; ++i) { offsets[i] = pos; pos += slen[i] + 1; }
  offsets[n_lists] = pos;
  run([&](int lo, int hi) {
    for (int i = lo; i < hi; ++i) {
      const int64_t room = out_cap - offsets[i];   // (<= 0 once the buffer is full: nothing more is written)
      if (room <= 0) continue;
      const long long len = rle_encode(counts + starts[i], lens[i], out + offsets[i], room);
      out[offsets[i] + (len < room - 1 ? len : room - 1)] = 0;
    }
  });
  return pos;
}

extern "C" int64_t yp_scratch_bytes(const yp_handle* h) { return h ? h->scratch_bytes : 0; }
