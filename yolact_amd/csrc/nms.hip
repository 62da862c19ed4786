// Kernels 2+3: top-k selection + Fast NMS per segment, and the final detection rows.
//
// Replaces detection.py:186-207 (_cc_fast_nms: top_k, gather, pairwise IoU, strict upper triangle,
// column max, threshold, gathers), detection.py:217-248 (the per-class intent of _fast_nms),
// detection.py:92-98 (_sanitize + normalise), the edge computation of utils.crop
// (utils/utils.py:116-117) and the box/class/score scatters of detection.py:109-114.
//
// A "segment" is one image (cross-class mode) or one (image, class) pair (per-class mode).  One
// workgroup owns a segment: it selects the top_k candidate keys (radix select on the 64-bit keys
// when the segment holds more than 256, otherwise directly), sorts them with a bitonic network in
// LDS, gathers the <= 256 decoded boxes into LDS and lets thread j reduce max_{i<j} IoU(i,j) in
// registers - the k x k IoU matrix (160 KB at k=200) is never materialised.  max is order
// independent, so the result equals the reference's reduce_max over the strict upper triangle.
#include "common.hpp"

struct NmsArgs {
  const u64* cand;            // [B,nseg,A]
  int32_t* cand_count;        // [B,nseg]
  int32_t* surv_count;        // [B]
  int32_t* stat_surv;         // [B]
  const float4* dec_box;      // [B,A]
  u64* pool;                  // [B,pool_cap]
  int32_t* pool_count;        // [B]
  float4* crop;               // [B,sel_rows]
  float* coef_out;            // [B,sel_rows,32]
  const float* coef_in;       // [B,A,32]
  float* det_boxes;           // [B,max_dets,4]
  float* det_classes;         // [B,max_dets]
  float* det_scores;          // [B,max_dets]
  int32_t* num_det;           // [B]
  int32_t* kept_idx;          // [B,max_dets] (nullable)
  int A, nseg, top_k, max_dets, sel_rows, pool_cap;
  float nms_thresh, img_h, img_w, proto_h, proto_w;
  int use_crop;
};

struct SelectShared {
  u64 key[YP_SORT_MAX];
  int hist[256];
  int scan[256];
  int sel, need, cnt, fill;
};

// Leaves the k = min(want, n) largest keys of list[0..n) in sh.key[0..k), descending; returns k.
// All threads of the workgroup must call it.  Keys are unique and non-zero.
template <int THREADS>
__device__ int select_topk_sorted(const u64* __restrict__ list, int n, int want, SelectShared& sh) {
  const int t = threadIdx.x;
  const int k = min(want, n);
  int m;  // keys placed in sh.key
  if (n <= YP_SORT_MAX) {
    if (t < YP_SORT_MAX) sh.key[t] = t < n ? list[t] : 0ull;
    m = n;
    __syncthreads();
  } else {
    // MSB-first radix select of the k-th largest key, 8 bits per pass.
    u64 prefix = 0, mask = 0;
    int need = k;
    for (int shift = 56; shift >= 0; shift -= 8) {
      if (t < 256) sh.hist[t] = 0;
      __syncthreads();
      for (int i = t; i < n; i += THREADS) {
        const u64 key = list[i];
        if ((key & mask) == prefix) atomicAdd(&sh.hist[(int)((key >> shift) & 0xFFull)], 1);
      }
      __syncthreads();
      if (t < 256) sh.scan[t] = sh.hist[t];
      __syncthreads();
      for (int o = 1; o < 256; o <<= 1) {   // inclusive suffix sums
        int v = 0;
        if (t < 256 && t + o < 256) v = sh.scan[t + o];
        __syncthreads();
        if (t < 256) sh.scan[t] += v;
        __syncthreads();
      }
      if (t < 256) {
        const int above = sh.scan[t] - sh.hist[t];
        if (above < need && need <= above + sh.hist[t]) { sh.sel = t; sh.need = need - above; sh.cnt = sh.hist[t]; }
      }
      __syncthreads();
      prefix |= (u64)sh.sel << shift;
      mask |= 0xFFull << shift;
      need = sh.need;
      const bool done = sh.cnt == need;
      __syncthreads();
      if (done) break;
    }
    if (t == 0) sh.fill = 0;
    if (t < YP_SORT_MAX) sh.key[t] = 0ull;
    __syncthreads();
    for (int i = t; i < n; i += THREADS) {
      const u64 key = list[i];
      if ((key & mask) >= prefix) {
        const int pos = atomicAdd(&sh.fill, 1);
        if (pos < YP_SORT_MAX) sh.key[pos] = key;
      }
    }
    m = k;
    __syncthreads();
  }
  // bitonic sort, descending, over the next power of two >= m
  int sz = 1;
  while (sz < m) sz <<= 1;
  for (int k2 = 2; k2 <= sz; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      if (t < sz) {
        const int ixj = t ^ j;
        if (ixj > t) {
          const u64 x = sh.key[t], y = sh.key[ixj];
          const bool desc = (t & k2) == 0;
          if (desc ? (x < y) : (x > y)) { sh.key[t] = y; sh.key[ixj] = x; }
        }
      }
      __syncthreads();
    }
  }
  return k;
}

// Writes output row r of image b (detection.py:92-98, 110-112) and the row's crop edges
// (utils/utils.py:116-117 with padding 1.0, normalised coordinates: SURVEY.md A12).
__device__ __forceinline__ void emit_row(const NmsArgs& p, int b, int r, float score, int cls, int anchor) {
  const float4 bx = p.dec_box[(size_t)b * p.A + anchor];      // ymin,xmin,ymax,xmax pixels
  float x1, x2, y1, y2;
  yp_sanitize(bx.y, bx.w, p.img_w, false, 0.0f, x1, x2);     // detection.py:126
  yp_sanitize(bx.x, bx.z, p.img_h, false, 0.0f, y1, y2);     // detection.py:127
  float4 nb;
  nb.x = y1 / p.img_h; nb.y = x1 / p.img_w; nb.z = y2 / p.img_h; nb.w = x2 / p.img_w;   // L93-98
  const size_t o = (size_t)b * p.max_dets + r;
  reinterpret_cast<float4*>(p.det_boxes)[o] = nb;
  p.det_classes[o] = (float)cls;
  p.det_scores[o] = score;
  if (p.kept_idx) p.kept_idx[o] = anchor;
  float4 cb;   // x1,x2,y1,y2 in proto pixels
  if (p.use_crop) {
    yp_sanitize(nb.y, nb.w, p.proto_w, true, 1.0f, cb.x, cb.y);
    yp_sanitize(nb.x, nb.z, p.proto_h, true, 1.0f, cb.z, cb.w);
  } else {
    cb.x = 0.0f; cb.y = p.proto_w; cb.z = 0.0f; cb.w = p.proto_h;
  }
  p.crop[(size_t)b * p.sel_rows + r] = cb;
}

// Rows [n, max_dets) of image b: zeros / -1 (detection.py:65-67); then num_det (L114).
template <int THREADS>
__device__ void finish_image(const NmsArgs& p, int b, int n, const int* s_anchor) {
  const int t = threadIdx.x;
  for (int r = n + t; r < p.max_dets; r += THREADS) {
    const size_t o = (size_t)b * p.max_dets + r;
    reinterpret_cast<float4*>(p.det_boxes)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    p.det_classes[o] = 0.0f;
    p.det_scores[o] = 0.0f;
    if (p.kept_idx) p.kept_idx[o] = -1;
  }
  // coefficients of the kept anchors, compacted for mask_assemble (detection.py:206)
  for (int i = t; i < n * (YP_MASK_DIM / 4); i += THREADS) {
    const int r = i / (YP_MASK_DIM / 4), q = i % (YP_MASK_DIM / 4);
    const float4 v = reinterpret_cast<const float4*>(p.coef_in + ((size_t)b * p.A + s_anchor[r]) * YP_MASK_DIM)[q];
    reinterpret_cast<float4*>(p.coef_out + ((size_t)b * p.sel_rows + r) * YP_MASK_DIM)[q] = v;
  }
  if (t == 0) {
    p.num_det[b] = n;
    p.stat_surv[b] = p.surv_count[b];
    p.surv_count[b] = 0;
  }
}

template <int THREADS, int MODE>
__global__ __launch_bounds__(THREADS) void segment_nms_kernel(const NmsArgs p) {
  __shared__ SelectShared sh;
  __shared__ float4 sbox[YP_SORT_MAX];
  __shared__ int smax[YP_SORT_MAX];
  __shared__ int s_anchor[YP_SORT_MAX];
  __shared__ int s_wave_cnt[YP_SORT_MAX / YP_WAVE];
  __shared__ int s_base;

  const int t = threadIdx.x;
  const int seg = blockIdx.x, b = blockIdx.y;
  const size_t sidx = (size_t)b * p.nseg + seg;
  const int n = p.cand_count[sidx];
  __syncthreads();
  if (t == 0) p.cand_count[sidx] = 0;   // leave the counter ready for the next run
  if (n == 0) {
    if (MODE == YP_NMS_CROSS_CLASS) finish_image<THREADS>(p, b, 0, s_anchor);   // detection.py:81
    return;
  }
  const int k = select_topk_sorted<THREADS>(p.cand + sidx * p.A, n, p.top_k, sh);   // detection.py:186

  if (t < YP_SORT_MAX) {
    smax[t] = 0;   // +0.0f: the zeroed diagonal/lower triangle take part in reduce_max
    if (t < k) sbox[t] = p.dec_box[(size_t)b * p.A + yp_seg_key_anchor(sh.key[t])];   // detection.py:187
  }
  __syncthreads();

  // iou_max[j] = max_{i<j} iou(i,j)                                          (detection.py:190-198)
  {
    constexpr int G = THREADS / YP_SORT_MAX;
    const int j = t % YP_SORT_MAX, g = t / YP_SORT_MAX;
    if (j < k) {
      const float4 bj = sbox[j];
      float mx = 0.0f;
      for (int i = g; i < j; i += G) mx = fmaxf(mx, yp_iou(sbox[i], bj));
      if (G == 1) smax[j] = __float_as_int(mx);
      else atomicMax(&smax[j], __float_as_int(mx));   // iou >= 0: int order == float order
    }
  }
  __syncthreads();

  // keep[j] = iou_max[j] <= thr, order preserved                             (detection.py:202)
  bool keep = false;
  if (t < YP_SORT_MAX && t < k) keep = __int_as_float(smax[t]) <= p.nms_thresh;
  int pos = 0;
  if (t < YP_SORT_MAX) {
    const u64 ball = __ballot(keep);
    pos = __popcll(ball & ((1ull << (t & 63)) - 1ull));
    if ((t & 63) == 0) s_wave_cnt[t >> 6] = __popcll(ball);
  }
  __syncthreads();
  int total = 0;
  if (t < YP_SORT_MAX) {
#pragma unroll
    for (int w = 0; w < YP_SORT_MAX / YP_WAVE; ++w) {
      if (w < (t >> 6)) pos += s_wave_cnt[w];
      total += s_wave_cnt[w];
    }
  }

  if (MODE == YP_NMS_CROSS_CLASS) {
    if (t >= YP_SORT_MAX) total = 0;
    // every thread needs `total`
    if (t == 0) s_base = total;
    __syncthreads();
    const int nk = min(s_base, p.max_dets);
    if (keep && pos < nk) {
      const u64 key = sh.key[t];
      const int anchor = yp_seg_key_anchor(key);
      s_anchor[pos] = anchor;
      emit_row(p, b, pos, yp_key_score(key), yp_seg_key_class(key), anchor);   // detection.py:204-207
    }
    __syncthreads();
    finish_image<THREADS>(p, b, nk, s_anchor);
  } else {
    // class-major pool of survivors for the cross-class top-N               (detection.py:234-239)
    if (t == 0) s_base = total ? atomicAdd(&p.pool_count[b], total) : 0;
    __syncthreads();
    if (keep) {
      const u64 key = sh.key[t];
      p.pool[(size_t)b * p.pool_cap + s_base + pos] =
          yp_fin_key(yp_key_score(key), yp_seg_key_anchor(key), seg + 1);
    }
  }
}

// Per-class mode only: cross-class top max_dets over the pooled survivors (detection.py:242-248).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void final_select_kernel(const NmsArgs p) {
  __shared__ SelectShared sh;
  __shared__ int s_anchor[YP_SORT_MAX];
  const int t = threadIdx.x;
  const int b = blockIdx.x;
  const int n = p.pool_count[b];
  __syncthreads();
  if (t == 0) p.pool_count[b] = 0;
  int k = 0;
  if (n > 0) {
    k = select_topk_sorted<THREADS>(p.pool + (size_t)b * p.pool_cap, n, min(p.max_dets, YP_SORT_MAX), sh);
    if (t < k) {
      const u64 key = sh.key[t];
      const int anchor = yp_fin_key_anchor(key);
      s_anchor[t] = anchor;
      emit_row(p, b, t, yp_key_score(key), yp_fin_key_class(key), anchor);
    }
    __syncthreads();
  }
  finish_image<THREADS>(p, b, k, s_anchor);
}

static NmsArgs make_args(const yp_handle* h, const float* pred_mask_coef, float* det_boxes, float* det_classes,
                         float* det_scores, int32_t* num_det, int32_t* kept_idx) {
  NmsArgs a;
  a.cand = h->s.cand; a.cand_count = h->s.cand_count; a.surv_count = h->s.surv_count; a.stat_surv = h->s.stat_surv;
  a.dec_box = h->s.dec_box; a.pool = h->s.pool; a.pool_count = h->s.pool_count; a.crop = h->s.crop;
  a.coef_out = h->s.coef; a.coef_in = pred_mask_coef;
  a.det_boxes = det_boxes; a.det_classes = det_classes; a.det_scores = det_scores; a.num_det = num_det;
  a.kept_idx = kept_idx;
  a.A = h->cfg.num_anchors; a.nseg = h->nseg; a.top_k = h->cfg.top_k; a.max_dets = h->cfg.max_dets;
  a.sel_rows = h->sel_rows; a.pool_cap = h->nseg * h->cfg.top_k;
  a.nms_thresh = h->cfg.nms_thresh;
  a.img_h = (float)h->cfg.img_h; a.img_w = (float)h->cfg.img_w;
  a.proto_h = (float)h->cfg.proto_h; a.proto_w = (float)h->cfg.proto_w;
  a.use_crop = h->cfg.use_cropped_mask;
  return a;
}

hipError_t yp_launch_segment_nms(const yp_handle* h, int batch, const float* pred_mask_coef, float* det_boxes,
                                 float* det_classes, float* det_scores, int32_t* num_det, int32_t* kept_idx,
                                 hipStream_t st) {
  const NmsArgs a = make_args(h, pred_mask_coef, det_boxes, det_classes, det_scores, num_det, kept_idx);
  if (h->cfg.nms_mode == YP_NMS_CROSS_CLASS)
    hipLaunchKernelGGL((segment_nms_kernel<1024, YP_NMS_CROSS_CLASS>), dim3(1, batch), dim3(1024), 0, st, a);
  else
    hipLaunchKernelGGL((segment_nms_kernel<256, YP_NMS_PER_CLASS>), dim3(h->nseg, batch), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t yp_launch_final_select(const yp_handle* h, int batch, const float* pred_mask_coef, float* det_boxes,
                                  float* det_classes, float* det_scores, int32_t* num_det, int32_t* kept_idx,
                                  hipStream_t st) {
  const NmsArgs a = make_args(h, pred_mask_coef, det_boxes, det_classes, det_scores, num_det, kept_idx);
  hipLaunchKernelGGL((final_select_kernel<1024>), dim3(batch), dim3(1024), 0, st, a);
  return hipGetLastError();
}
