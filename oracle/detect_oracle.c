/*
 * CPU oracle, plain-C restatement of YOLACT's Detect post-processing.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the shipped library
 * (yolact_amd/csrc) contains no CPU path and never links or calls it.
 *
 * Follows, in order (all paths relative to /root/reference):
 *   detection.py:44-117   Detect.__call__        -> yo_detect()
 *   utils/utils.py:173-213 _decode               -> yo_decode()
 *   detection.py:181-209  _cc_fast_nms           -> nms_cc()
 *   detection.py:211-250  _fast_nms (dead; spec) -> nms_per_class()
 *   utils/utils.py:7-70   _area/_intersection/_iou -> yo_iou()
 *   detection.py:119-131 + utils/utils.py:89-104 _sanitize / sanitize_coordinates
 *   utils/utils.py:108-133 crop (padding=1, normalized)
 *
 * PARITY UNPINNED: TensorFlow cannot be imported in the build container and the
 * reference holds no golden vectors for this path.  This file
 * is cross-checked against the independent numpy restatement
 * (oracle/detect_oracle.py, libm exp, numpy reductions) by tests/test_oracle.py.
 *
 * Arithmetic contract ("YP-MATH").  TF leaves the exp
 * polynomial, the softmax reduction order and the matmul accumulation order to
 * Eigen, none of which is pinned by the reference.  This oracle fixes them so that
 * a second implementation can be compared BIT FOR BIT:
 *   exp      Cephes expf (the polynomial Eigen's pexp<float> is derived from),
 *            every multiply-add an explicit fmaf, argument clamped to [-87, 87];
 *   softmax  e_c = exp(x_c - max), S = 16 strided partial sums
 *            (s_j = e_j + e_{j+16} + ...) folded by a binary tree (8,4,2,1) -
 *            the shape of a packet reduction -, p_c = e_c * (1/S): ONE IEEE reciprocal
 *            per row, then a multiply per class (Eigen's form);
 *   matmul   K-ordered fmaf chain starting from 0 (what v_mfma_f32_*_f32 and a
 *            scalar loop both compute);
 *   sigmoid  1 / (1 + exp(-x)) with the exp above and an IEEE divide;
 *   every other line of the reference is single IEEE f32 operations in the
 *   order the Python expression evaluates them (no contraction: build with
 *   -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define YO_NMS_CC 0
#define YO_NMS_PER_CLASS 1

static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ---- YP-MATH exp ------------------------------------------------------- */
float yo_exp(float x) {
  x = fminf(fmaxf(x, -87.0f), 87.0f);
  float fx = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
  float r = fmaf(fx, -0.693359375f, x);
  r = fmaf(fx, 2.12194440e-4f, r);
  float z = r * r;
  float y = 1.9875691500E-4f;
  y = fmaf(y, r, 1.3981999507E-3f);
  y = fmaf(y, r, 8.3334519073E-3f);
  y = fmaf(y, r, 4.1665795894E-2f);
  y = fmaf(y, r, 1.6666665459E-1f);
  y = fmaf(y, r, 5.0000001201E-1f);
  y = fmaf(y, z, r);
  y = y + 1.0f;
  int n = (int)fx;
  return y * bits2f((uint32_t)(n + 127) << 23);
}

float yo_sigmoid(float x) { return 1.0f / (1.0f + yo_exp(-x)); }

/* tf.nn.softmax over one row (detection.py:56), YP-MATH reduction order. */
void yo_softmax_row(const float* x, int C, float* p) {
  float m = x[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
  float s[16];
  for (int j = 0; j < 16; ++j) s[j] = 0.0f;
  for (int c = 0; c < C; ++c) {
    p[c] = yo_exp(x[c] - m);
    s[c & 15] = s[c & 15] + p[c];
  }
  for (int w = 8; w >= 1; w >>= 1)
    for (int j = 0; j < w; ++j) s[j] = s[j] + s[j + w];
  const float rS = 1.0f / s[0];                  /* exp(x - max) * reciprocal-of-sum */
  for (int c = 0; c < C; ++c) p[c] = p[c] * rS;
}

/* utils._decode, utils/utils.py:193-213.  prior/out are [ymin,xmin,ymax,xmax]. */
void yo_decode(const float* off, const float* prior, float* out) {
  float ph = prior[2] - prior[0];                 /* L193 */
  float pw = prior[3] - prior[1];                 /* L194 */
  float cx = prior[1] + (pw / 2.0f);              /* L196 */
  float cy = prior[0] + (ph / 2.0f);              /* L197 */
  float bx = cx + (off[0] * pw) * 0.1f;           /* L201 */
  float by = cy + (off[1] * ph) * 0.1f;
  float bw = pw * yo_exp(off[2] * 0.2f);          /* L202 */
  float bh = ph * yo_exp(off[3] * 0.2f);
  float xmin = bx - bw / 2.0f;                    /* L210 */
  float ymin = by - bh / 2.0f;
  float xmax = bw / 2.0f + bx;
  float ymax = bh / 2.0f + by;
  out[0] = ymin; out[1] = xmin; out[2] = ymax; out[3] = xmax;   /* L213 */
}

/* utils._iou for one pair, utils/utils.py:19-21,40-48,63-70. */
float yo_iou(const float* a, const float* b) {
  float area1 = (a[2] - a[0]) * (a[3] - a[1]);
  float area2 = (b[2] - b[0]) * (b[3] - b[1]);
  float ih = fmaxf(0.0f, fminf(a[2], b[2]) - fmaxf(a[0], b[0]));
  float iw = fmaxf(0.0f, fminf(a[3], b[3]) - fmaxf(a[1], b[1]));
  float inter = ih * iw;
  float uni = (area1 + area2) - inter;
  return inter == 0.0f ? 0.0f : inter / uni;
}

static inline float clipf(float t, float lo, float hi) { return fminf(fmaxf(t, lo), hi); }

/* utils.sanitize_coordinates, utils/utils.py:89-104. */
void yo_sanitize(float a, float b, float img_size, int normalized, float padding, float* x1, float* x2) {
  if (normalized) { a = a * img_size; b = b * img_size; }
  float lo = fminf(a, b), hi = fmaxf(a, b);
  *x1 = clipf(lo - padding, 0.0f, img_size);
  *x2 = clipf(hi + padding, 0.0f, img_size);
}

/* ---- sorting: tf.math.top_k order = score desc, ties lower index first ---- */
typedef struct { float score; int32_t idx; } yo_item;
static int cmp_item(const void* pa, const void* pb) {
  const yo_item* a = (const yo_item*)pa; const yo_item* b = (const yo_item*)pb;
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  return (a->idx > b->idx) - (a->idx < b->idx);
}

/* Fast NMS column max (detection.py:190-202): keep[j] = max_{i<j} iou(i,j) <= thr */
static void fast_nms_keep(const float* boxes /*[k,4] sorted*/, int k, float thr, uint8_t* keep) {
  for (int j = 0; j < k; ++j) {
    float mx = 0.0f;   /* the zeroed diagonal / lower triangle take part in reduce_max */
    for (int i = 0; i < j; ++i) mx = fmaxf(mx, yo_iou(boxes + 4 * i, boxes + 4 * j));
    keep[j] = mx <= thr;
  }
}

typedef struct {
  int n;
  int32_t* surv_idx;   /* index into the survivor list */
  float* score;
  float* cls;          /* 1-based, float */
} yo_dets;

/* Detect._cc_fast_nms, detection.py:181-209.  probs: [S, Cfg] foreground probabilities. */
static void nms_cc(const float* probs, const float* boxes, int S, int Cfg, float thr, int top_k, yo_dets* out) {
  yo_item* it = (yo_item*)malloc(sizeof(yo_item) * (size_t)S);
  int32_t* cls = (int32_t*)malloc(sizeof(int32_t) * (size_t)S);
  for (int s = 0; s < S; ++s) {
    const float* p = probs + (size_t)s * Cfg;
    int arg = 0; float mx = p[0];
    for (int c = 1; c < Cfg; ++c) if (p[c] > mx) { mx = p[c]; arg = c; }   /* L184-185: first max */
    it[s].score = mx; it[s].idx = s; cls[s] = arg + 1;
  }
  int k = top_k < S ? top_k : S;                                           /* L186 */
  qsort(it, (size_t)S, sizeof(yo_item), cmp_item);
  float* sb = (float*)malloc(sizeof(float) * 4 * (size_t)(k > 0 ? k : 1));
  uint8_t* keep = (uint8_t*)malloc((size_t)(k > 0 ? k : 1));
  for (int j = 0; j < k; ++j) memcpy(sb + 4 * j, boxes + 4 * (size_t)it[j].idx, 16);   /* L187 */
  fast_nms_keep(sb, k, thr, keep);                                         /* L190-202 */
  int n = 0;
  for (int j = 0; j < k; ++j) if (keep[j]) {
    out->surv_idx[n] = it[j].idx; out->score[n] = it[j].score; out->cls[n] = (float)cls[it[j].idx]; ++n;
  }
  out->n = n;
  free(it); free(cls); free(sb); free(keep);
}

/* Intent of the dead Detect._fast_nms, detection.py:211-250, 1-based class ids. */
static void nms_per_class(const float* probs, const float* boxes, int S, int Cfg, float conf, float thr, int top_k,
                          int max_num, yo_dets* out) {
  const int kmax = top_k < S ? top_k : S;
  yo_item* it = (yo_item*)malloc(sizeof(yo_item) * (size_t)(S > 0 ? S : 1));
  float* sb = (float*)malloc(sizeof(float) * 4 * (size_t)(kmax > 0 ? kmax : 1));
  uint8_t* keep = (uint8_t*)malloc((size_t)(kmax > 0 ? kmax : 1));
  size_t cap = (size_t)Cfg * (size_t)(kmax > 0 ? kmax : 1);
  yo_item* flat = (yo_item*)malloc(sizeof(yo_item) * cap);     /* class-major kept list */
  int32_t* flat_surv = (int32_t*)malloc(sizeof(int32_t) * cap);
  int32_t* flat_cls = (int32_t*)malloc(sizeof(int32_t) * cap);
  int nf = 0;
  for (int c = 0; c < Cfg; ++c) {
    /* tf.math.top_k over all S scores of the class (L217), Fast NMS among the k boxes (L219-229), then
     * `scores > conf` (L233-234).  Only the entries with score > conf are sorted here: they come first in the
     * descending order, a box is suppressed only by boxes ranked above it (so never by an entry <= conf), and
     * entries <= conf are dropped at L233 anyway - the kept list is the same as with all S entries sorted,
     * which is what this function did at first (80 qsorts of S items per image: most of the port's time). */
    int m = 0;
    for (int s = 0; s < S; ++s) {
      const float sc = probs[(size_t)s * Cfg + c];
      if (sc > conf) { it[m].score = sc; it[m].idx = s; ++m; }
    }
    qsort(it, (size_t)m, sizeof(yo_item), cmp_item);                       /* L217 */
    const int k = top_k < m ? top_k : m;
    for (int j = 0; j < k; ++j) memcpy(sb + 4 * j, boxes + 4 * (size_t)it[j].idx, 16);   /* L219 */
    fast_nms_keep(sb, k, thr, keep);                                       /* L221-229 */
    for (int j = 0; j < k; ++j)
      if (keep[j] && it[j].score > conf) {                                 /* L233-234 */
        flat[nf].score = it[j].score; flat[nf].idx = nf;
        flat_surv[nf] = it[j].idx; flat_cls[nf] = c + 1; ++nf;
      }
  }
  int n = max_num < nf ? max_num : nf;                                     /* L242 */
  qsort(flat, (size_t)nf, sizeof(yo_item), cmp_item);                      /* L243: ties -> lower flat index */
  for (int r = 0; r < n; ++r) {
    int f = flat[r].idx;
    out->surv_idx[r] = flat_surv[f]; out->score[r] = flat[r].score; out->cls[r] = (float)flat_cls[f];
  }
  out->n = n;
  free(it); free(sb); free(keep); free(flat); free(flat_surv); free(flat_cls);
}

/* One image of Detect.__call__'s loop body, detection.py:73-114. */
static void detect_image(int A, int C, int K, int ph, int pw, int img_h, int img_w, float conf, float nms_thr,
                         int nms_mode, int top_k, int max_dets, int use_crop, const float* off, const float* cls,
                         const float* coef, const float* priors, const float* proto, float* det_boxes,
                         float* det_classes, float* det_scores, float* det_masks, int32_t* num_det,
                         int32_t* anchor_idx, int32_t* num_surv) {
  const int Cfg = C - 1;
  const size_t P = (size_t)ph * pw;
  float* prob = (float*)malloc(sizeof(float) * (size_t)C);
  /* pass 1: softmax (L56), drop background (L59), max (L62), threshold (L73) */
  int32_t* surv = (int32_t*)malloc(sizeof(int32_t) * (size_t)A);
  float* sprob = NULL; size_t sprob_cap = 0; int S = 0;
  for (int a = 0; a < A; ++a) {
    yo_softmax_row(cls + (size_t)a * C, C, prob);
    float mx = prob[1];
    for (int c = 2; c < C; ++c) mx = fmaxf(mx, prob[c]);
    if (mx > conf) {                                                       /* strict >, L73 */
      if ((size_t)(S + 1) * Cfg > sprob_cap) {
        sprob_cap = sprob_cap ? sprob_cap * 2 : (size_t)1024 * Cfg;
        sprob = (float*)realloc(sprob, sizeof(float) * sprob_cap);
      }
      memcpy(sprob + (size_t)S * Cfg, prob + 1, sizeof(float) * (size_t)Cfg);
      surv[S++] = a;
    }
  }
  *num_surv = S;
  for (int r = 0; r < max_dets; ++r) {
    det_boxes[4 * r] = det_boxes[4 * r + 1] = det_boxes[4 * r + 2] = det_boxes[4 * r + 3] = 0.0f;
    det_classes[r] = 0.0f; det_scores[r] = 0.0f; anchor_idx[r] = -1;
  }
  *num_det = 0;
  if (S == 0) { free(prob); free(surv); free(sprob); return; }              /* L81 */

  float* boxes = (float*)malloc(sizeof(float) * 4 * (size_t)S);             /* L79 */
  for (int s = 0; s < S; ++s) yo_decode(off + 4 * (size_t)surv[s], priors + 4 * (size_t)surv[s], boxes + 4 * s);

  int cap = nms_mode == YO_NMS_CC ? (top_k < S ? top_k : S) : max_dets;
  if (cap < 1) cap = 1;
  yo_dets d; d.n = 0;
  d.surv_idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
  d.score = (float*)malloc(sizeof(float) * (size_t)cap);
  d.cls = (float*)malloc(sizeof(float) * (size_t)cap);
  if (nms_mode == YO_NMS_CC) nms_cc(sprob, boxes, S, Cfg, nms_thr, top_k, &d);       /* L83 */
  else nms_per_class(sprob, boxes, S, Cfg, conf, nms_thr, top_k, max_dets, &d);
  int n = d.n < max_dets ? d.n : max_dets;   /* caller-side [:det_num] slice of a score-descending list */
  *num_det = n;                                                             /* L87, L114 */

  const float fH = (float)img_h, fW = (float)img_w, fph = (float)ph, fpw = (float)pw;
  for (int r = 0; r < n; ++r) {
    const int s = d.surv_idx[r];
    const int a = surv[s];
    const float* bx = boxes + 4 * (size_t)s;
    float x1, x2, y1, y2;
    yo_sanitize(bx[1], bx[3], fW, 0, 0.0f, &x1, &x2);                       /* L92, L126-127 */
    yo_sanitize(bx[0], bx[2], fH, 0, 0.0f, &y1, &y2);
    float nb[4] = { y1 / fH, x1 / fW, y2 / fH, x2 / fW };                   /* L93-98 */
    memcpy(det_boxes + 4 * r, nb, 16);                                      /* L110 */
    det_classes[r] = d.cls[r];                                              /* L111 */
    det_scores[r] = d.score[r];                                             /* L112 */
    anchor_idx[r] = a;
    if (det_masks) {
      float cx1 = 0, cx2 = fpw, cy1 = 0, cy2 = fph;
      if (use_crop) {                                                       /* utils.crop L116-117 */
        yo_sanitize(nb[1], nb[3], fpw, 1, 1.0f, &cx1, &cx2);
        yo_sanitize(nb[0], nb[2], fph, 1, 1.0f, &cy1, &cy2);
      }
      const float* cf = coef + (size_t)a * K;
      float* m = det_masks + (size_t)r * P;
      for (int row = 0; row < ph; ++row)
        for (int col = 0; col < pw; ++col) {
          const float* pr = proto + ((size_t)row * pw + col) * K;
          float acc = 0.0f;
          for (int k = 0; k < K; ++k) acc = fmaf(pr[k], cf[k], acc);        /* L89 */
          float v = yo_sigmoid(acc);                                        /* L90 */
          if (use_crop) {                                                   /* L125-133 */
            float fr = (float)row, fc = (float)col;
            int inside = (fc >= cx1) && (fc < cx2) && (fr >= cy1) && (fr < cy2);
            v = v * (inside ? 1.0f : 0.0f);
          }
          m[(size_t)row * pw + col] = clipf(v, 0.0f, 1.0f);                 /* L105, L107 */
        }
    }
  }
  free(prob); free(surv); free(sprob); free(boxes); free(d.surv_idx); free(d.score); free(d.cls);
}

/*
 * Detect.__call__ over a batch.  Compact outputs: [B,max_dets,...]; rows >= num_det[b] of
 * boxes/classes/scores are zero and of anchor_idx are -1; mask rows >= num_det[b] are left
 * untouched (caller pre-zeroes if the reference's padded layout is wanted).
 * det_masks may be NULL.  nthreads > 1 parallelises over images (independent, detection.py:71).
 */
int yo_detect(int batch, int A, int C, int K, int ph, int pw, int img_h, int img_w, float conf, float nms_thr,
              int nms_mode, int top_k, int max_dets, int use_crop, const float* off, const float* cls,
              const float* coef, const float* priors, const float* proto, float* det_boxes, float* det_classes,
              float* det_scores, float* det_masks, int32_t* num_det, int32_t* anchor_idx, int32_t* num_surv,
              int nthreads) {
  if (batch < 0 || A <= 0 || C < 2 || K <= 0 || max_dets <= 0) return -1;
  const size_t P = (size_t)ph * pw;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int b = 0; b < batch; ++b) {
    detect_image(A, C, K, ph, pw, img_h, img_w, conf, nms_thr, nms_mode, top_k, max_dets, use_crop,
                 off + (size_t)b * A * 4, cls + (size_t)b * A * C, coef + (size_t)b * A * K, priors,
                 proto + (size_t)b * P * K, det_boxes + (size_t)b * max_dets * 4, det_classes + (size_t)b * max_dets,
                 det_scores + (size_t)b * max_dets, det_masks ? det_masks + (size_t)b * max_dets * P : NULL,
                 num_det + b, anchor_idx + (size_t)b * max_dets, num_surv + b);
  }
  return 0;
}

/*
 * Mask assembly of the training loss, loss/loss_yolact.py:239-253 (forward): for one image
 *   mask_p = sigmoid(proto @ pos_coef^T)                                   L239-241
 *   boxes  = pos_prior_box / [img_h, img_w, img_h, img_w]                  L247-252
 *   mask_p = utils.crop(mask_p, boxes)                                     L253 (utils/utils.py:108-133)
 * Output [num_pos, ph, pw] (the transpose of L273).  No clip (L257-258 are commented out there).
 * Same arithmetic contract as detect_image above.
 */
void yo_loss_masks(int num_pos, int K, int ph, int pw, int img_h, int img_w, int use_crop, const float* proto,
                   const float* pos_coef, const float* pos_box, float* out) {
  const float fH = (float)img_h, fW = (float)img_w, fph = (float)ph, fpw = (float)pw;
  const size_t P = (size_t)ph * pw;
  for (int r = 0; r < num_pos; ++r) {
    const float* bx = pos_box + 4 * (size_t)r;
    const float nb[4] = { bx[0] / fH, bx[1] / fW, bx[2] / fH, bx[3] / fW };
    float cx1 = 0, cx2 = fpw, cy1 = 0, cy2 = fph;
    if (use_crop) {
      yo_sanitize(nb[1], nb[3], fpw, 1, 1.0f, &cx1, &cx2);
      yo_sanitize(nb[0], nb[2], fph, 1, 1.0f, &cy1, &cy2);
    }
    const float* cf = pos_coef + (size_t)r * K;
    float* m = out + (size_t)r * P;
    for (int row = 0; row < ph; ++row)
      for (int col = 0; col < pw; ++col) {
        const float* pr = proto + ((size_t)row * pw + col) * K;
        float acc = 0.0f;
        for (int k = 0; k < K; ++k) acc = fmaf(pr[k], cf[k], acc);
        float v = yo_sigmoid(acc);
        if (use_crop) {
          float fr = (float)row, fc = (float)col;
          int inside = (fc >= cx1) && (fc < cx2) && (fr >= cy1) && (fr < cy2);
          v = v * (inside ? 1.0f : 0.0f);
        }
        m[(size_t)row * pw + col] = v;
      }
  }
}

/* ---------------------------------------------------------------------------------------------
 * Gradients of that mask assembly w.r.t. proto and pos_coef, given dL/d mask_p.
 * The reference has no code for this step - TensorFlow's autodiff differentiates L239-253:
 *   crop (a product with a constant 0/1 mask): g passes inside the box, 0 outside;
 *   tf.sigmoid: SigmoidGrad = dy * y * (1 - y);   tf.linalg.matmul: dA = dC B, dB = A^T dC.
 * Since the cropped forward output is 0 outside the box, dz = (g * m) * (1 - m) with m = mask_p as the
 * forward returned it covers the crop too.  TF does not fix a summation order; the HIP kernels do, and
 * this restatement follows them so that the comparison can be bit-exact (YP-MATH, fp-contract off):
 *   grad_proto[p][k] = fmaf chain over r = 0..num_pos-1 (ascending) of dz[r][p] * pos_coef[r][k]
 *   grad_coef[r][k]  = per pixel row y one fmaf chain over x ascending of dz[r][p] * proto[p][k]; the rows'
 *                      values added up in ascending y, starting from +0.
 * Terms with dz == 0 change nothing (finite inputs), so the kernels skip the pixels outside the box.
 */
void yo_loss_masks_backward(int num_pos, int K, int ph, int pw, const float* proto, const float* pos_coef,
                            const float* masks, const float* grad_masks, float* grad_proto, float* grad_coef) {
  const size_t P = (size_t)ph * pw;
  for (size_t p = 0; p < P; ++p)
    for (int k = 0; k < K; ++k) {
      float acc = 0.0f;
      for (int r = 0; r < num_pos; ++r) {
        const float m = masks[(size_t)r * P + p], g = grad_masks[(size_t)r * P + p];
        const float dz = (g * m) * (1.0f - m);
        acc = fmaf(dz, pos_coef[(size_t)r * K + k], acc);
      }
      grad_proto[p * K + k] = acc;
    }
  for (int r = 0; r < num_pos; ++r)
    for (int k = 0; k < K; ++k) {
      float total = 0.0f;
      for (int y = 0; y < ph; ++y) {
        float acc = 0.0f;
        for (int x = 0; x < pw; ++x) {
          const size_t p = (size_t)y * pw + x;
          const float m = masks[(size_t)r * P + p], g = grad_masks[(size_t)r * P + p];
          const float dz = (g * m) * (1.0f - m);
          acc = fmaf(dz, proto[p * K + k], acc);
        }
        total = total + acc;
      }
      grad_coef[(size_t)r * K + k] = total;
    }
}

int yo_has_openmp(void) {
#ifdef _OPENMP
  return 1;
#else
  return 0;
#endif
}

/* The library is built with -mfma; callers check this before any compute call. */
int yo_cpu_ok(void) { return __builtin_cpu_supports("fma") ? 1 : 0; }
