// Consumer step after Detect: every kept mask resized to the image size,
// binarised and run-length encoded, on the device.
//
// Replaces, per detection, coco_eval.py:133-136 / infer.py:31-33 (cv2.resize of the float mask to
// (w, h), `> 0.5`, Fortran-order bitmap, pycocotools encode) and train.py:497,503-506 (`> 0.5` first,
// uint8, cv2.resize of the 0/1 bytes).  The resize is OpenCV's INTER_LINEAR restated operation by
// operation (half-pixel centres computed in double and rounded to float, border clamps, horizontal
// pass first, plain float multiplies and adds / 11-bit fixed point for bytes); the run lengths are the
// column-major counts of coco_eval.py:91-98 (binary_mask_to_rle) = pycocotools' rleEncode.
//
// One 1024-thread workgroup per detection walks the output image in column-major order, 16 columns
// at a time, one wave per column:
//   A  the column's source columns sx, sx+1 and weights are wave-uniform: the wave builds the
//      horizontally interpolated source column (proto_h values) in LDS - exactly OpenCV's row buffer;
//   B  64 output rows at a time: two LDS reads, the vertical interpolation, the threshold, one ballot
//      -> 64 bits of the bitmap.  Run boundaries are the set bits of m ^ (m << 1 | carry): counting
//      and locating them is popcount / clz on 64-bit masks;
//   C  after a barrier every wave knows how many boundaries precede its column and where the last one
//      was, and writes its run lengths straight to their final places: count j = position of boundary j
//      minus position of boundary j-1.  No second pass, no atomics.
// HBM traffic is the source mask (read through L2, reused by ~4 neighbouring columns) plus the counts
// (and the optional bitmap): the kernel is bound by instruction issue, ~25 per output pixel.
#include "common.hpp"

#define YP_RLE_THREADS 1024
#define YP_RLE_COLS (YP_RLE_THREADS / YP_WAVE)   // columns in flight = waves
#define YP_RLE_MAX_SRC_H 512
#define YP_RLE_MAX_OUT_H 4096

struct RleArgs {
  const float* masks;       // [B,max_dets,ph,pw]
  const int32_t* num_det;   // [B]
  uint8_t* bitmaps;         // [B,max_dets,out_h,out_w] row-major 0/1, or null
  uint32_t* counts;         // [B,max_dets,cap]
  int32_t* rle_len;         // [B,max_dets]
  int max_dets, ph, pw, out_h, out_w, cap, order;
  float threshold;
  double scale_x, scale_y;  // 1. / ((double)dst / src), as cv::resize computes them
};

struct RleShared {
  union { float f[YP_RLE_COLS][YP_RLE_MAX_SRC_H]; int i[YP_RLE_COLS][YP_RLE_MAX_SRC_H]; } hcol;   // OpenCV's row buffer, per column
  u64 bits[YP_RLE_COLS][YP_RLE_MAX_OUT_H / 64];   // the column's bitmap
  u64 trans[YP_RLE_COLS][YP_RLE_MAX_OUT_H / 64];  // its run boundaries, the column's first pixel excluded
  float fy[YP_RLE_MAX_OUT_H];                     // vertical weight of every output row
  short sy[YP_RLE_MAX_OUT_H];                     // and its (unclipped) upper source row
  int col_first[YP_RLE_COLS], col_lastbit[YP_RLE_COLS], col_count[YP_RLE_COLS];
  long long col_lastpos[YP_RLE_COLS];             // position of the column's last inner boundary, -1 if none
};

// cv::resize's coefficient of destination index d: source index (clamped at the borders, which zero
// the weight) and weight of the second tap.  `right` = the taps collapse onto the last source element.
__device__ __forceinline__ void linear_coeff_x(int d, double scale, int src, int& s, float& f, bool& right) {
  f = (float)(((double)d + 0.5) * scale - 0.5);
  s = (int)floorf(f);
  f = f - (float)s;
  if (s < 0) { f = 0.0f; s = 0; }
  right = s >= src - 1;
  if (right) { f = 0.0f; s = src - 1; }
}

__global__ __launch_bounds__(YP_RLE_THREADS) void mask_rle_kernel(const RleArgs p) {
  __shared__ RleShared sh;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = blockIdx.x, b = blockIdx.y;
  const size_t slot = (size_t)b * p.max_dets + n;
  if (n >= p.num_det[b]) {                       // not a detection: an empty encoding
    if (t == 0) p.rle_len[slot] = 0;
    return;
  }
  const float* __restrict__ S = p.masks + slot * (size_t)p.ph * p.pw;
  uint32_t* const counts = p.counts + slot * (size_t)p.cap;
  uint8_t* const bitmap = p.bitmaps ? p.bitmaps + slot * (size_t)p.out_h * p.out_w : nullptr;
  const bool identity = p.ph == p.out_h && p.pw == p.out_w;     // cv::resize copies when the sizes agree
  const bool fixed = p.order == YP_THRESHOLD_THEN_RESIZE;
  const float one = 1.0f, cscale = 2048.0f;                      // INTER_RESIZE_COEF_SCALE
  const int HB = (p.out_h + 63) >> 6;

  // vertical coefficients of every output row (weights kept, row indices clipped at use)
  for (int y = t; y < p.out_h; y += YP_RLE_THREADS) {
    float f = (float)(((double)y + 0.5) * p.scale_y - 0.5);
    const int s = (int)floorf(f);
    f = f - (float)s;
    sh.fy[y] = f;
    sh.sy[y] = (short)s;
  }
  __syncthreads();

  const u64 lt_mask = (1ull << lane) - 1ull;
  long long total = 0;        // boundaries before the current column group (identical in every thread)
  long long last_pos = 0;     // position of the last boundary so far (0: none yet)
  int prev_lastbit = 0;       // bit of the last pixel of the previous column
  for (int x0 = 0; x0 < p.out_w; x0 += YP_RLE_COLS) {
    const int x = x0 + wave;
    const bool active = x < p.out_w;             // wave-uniform
    const long long col_start = (long long)x * p.out_h;
    if (active) {
      // ---- A: horizontal pass for this column ------------------------------------------------------
      if (!identity) {
        int sx; float fx; bool right;
        linear_coeff_x(x, p.scale_x, p.pw, sx, fx, right);
        const float a0 = one - fx, a1 = fx;
        const int ia0 = (int)rintf(a0 * cscale), ia1 = (int)rintf(a1 * cscale);   // saturate_cast<short>: cvRound
        const int sx1 = min(sx + 1, p.pw - 1);
        for (int r = lane; r < p.ph; r += YP_WAVE) {
          const float s0 = S[(size_t)r * p.pw + sx], s1 = S[(size_t)r * p.pw + sx1];
          if (!fixed) {
            sh.hcol.f[wave][r] = right ? s0 * one : s0 * a0 + s1 * a1;
          } else {
            const int b0 = s0 > p.threshold ? 1 : 0, b1 = s1 > p.threshold ? 1 : 0;            // train.py:497
            sh.hcol.i[wave][r] = right ? b0 * 2048 : b0 * ia0 + b1 * ia1;
          }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): single wave, LDS in order
        __builtin_amdgcn_wave_barrier();
      }
      // ---- B: vertical pass, threshold, bits and boundaries --------------------------------------
      int cnt = 0, carry = 0;
      long long lastpos = -1;
      for (int c = 0; c < HB; ++c) {
        const int y = c * 64 + lane;
        const bool valid = y < p.out_h;
        bool bit = false;
        if (valid) {
          if (identity) {
            bit = S[(size_t)y * p.pw + x] > p.threshold;
          } else {
            const int s = sh.sy[y];
            const float f = sh.fy[y];
            const int r0 = min(max(s, 0), p.ph - 1), r1 = min(max(s + 1, 0), p.ph - 1);
            if (!fixed) {
              const float b0 = one - f, b1 = f;
              const float v = sh.hcol.f[wave][r0] * b0 + sh.hcol.f[wave][r1] * b1;
              bit = v > p.threshold;                                                           // coco_eval.py:135
            } else {
              const int ib0 = (int)rintf((one - f) * cscale), ib1 = (int)rintf(f * cscale);
              const int v = (((ib0 * (sh.hcol.i[wave][r0] >> 4)) >> 16) + ((ib1 * (sh.hcol.i[wave][r1] >> 4)) >> 16) + 2) >> 2;
              bit = (v & 0xff) != 0;
            }
          }
        }
        const u64 m = __ballot(bit);
        const int nvalid = min(64, p.out_h - c * 64);
        const u64 vmask = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        // a boundary wherever a pixel differs from the one before it; the column's first pixel is
        // compared with the previous COLUMN's last pixel later (phase C), so it is excluded here
        u64 tr = (m ^ ((m << 1) | (u64)(c == 0 ? (int)(m & 1ull) : carry))) & vmask;
        carry = (int)((m >> (nvalid - 1)) & 1ull);
        cnt += __popcll(tr);
        if (tr != 0ull) lastpos = col_start + c * 64 + (63 - __clzll((long long)tr));
        if (lane == 0) { sh.bits[wave][c] = m; sh.trans[wave][c] = tr; }
      }
      if (lane == 0) {
        sh.col_first[wave] = (int)(sh.bits[wave][0] & 1ull);
        sh.col_lastbit[wave] = carry;
        sh.col_count[wave] = cnt;
        sh.col_lastpos[wave] = lastpos;
      }
    }
    __syncthreads();
    // ---- C: where do this column's boundaries go? --------------------------------------------------
    const int ncols = min(YP_RLE_COLS, p.out_w - x0);
    long long off = total, lp = last_pos;
    int lb = prev_lastbit;
    long long g_total = total, g_lp = last_pos;
    int g_lb = prev_lastbit;
    for (int w = 0; w < ncols; ++w) {            // uniform, <= 16 iterations
      const int first = sh.col_first[w] != g_lb ? 1 : 0;
      const long long cs = (long long)(x0 + w) * p.out_h;
      if (w == wave) { off = g_total; lp = g_lp; lb = g_lb; }
      g_total += first + sh.col_count[w];
      if (first) g_lp = cs;
      if (sh.col_lastpos[w] >= 0) g_lp = sh.col_lastpos[w];
      g_lb = sh.col_lastbit[w];
    }
    if (active) {
      if (sh.col_first[wave] != lb) {            // the column's first pixel starts a run
        if (lane == 0 && off < p.cap) counts[off] = (uint32_t)(col_start - lp);
        off += 1;
        lp = col_start;
      }
      for (int c = 0; c < HB; ++c) {
        const u64 tr = sh.trans[wave][c];
        if (tr == 0ull) continue;                // wave-uniform
        if ((tr >> lane) & 1ull) {
          const u64 below = tr & lt_mask;
          const long long pos = col_start + c * 64 + lane;
          const long long prev = below ? col_start + c * 64 + (63 - __clzll((long long)below)) : lp;
          const long long j = off + __popcll(below);
          if (j < p.cap) counts[j] = (uint32_t)(pos - prev);
        }
        off += __popcll(tr);
        lp = col_start + c * 64 + (63 - __clzll((long long)tr));
      }
    }
    // ---- the row-major bitmap, 16 columns of every row per group (train.py:502-506's detection_masks) ----
    if (bitmap) {
      for (int idx = t; idx < p.out_h * YP_RLE_COLS; idx += YP_RLE_THREADS) {
        const int y = idx / YP_RLE_COLS, xc = idx - y * YP_RLE_COLS;
        if (xc < ncols) bitmap[(size_t)y * p.out_w + x0 + xc] = (uint8_t)((sh.bits[xc][y >> 6] >> (y & 63)) & 1ull);
      }
    }
    total = g_total; last_pos = g_lp; prev_lastbit = g_lb;
    __syncthreads();                             // LDS is rewritten by the next group
  }
  if (t == 0) {
    const long long N = (long long)p.out_h * p.out_w;
    if (total < p.cap) counts[total] = (uint32_t)(N - last_pos);   // the run that reaches the end
    p.rle_len[slot] = (int32_t)(total + 1 <= 0x7fffffffll ? total + 1 : 0x7fffffff);
  }
}

hipError_t yp_launch_mask_rle(int batch, int max_dets, int ph, int pw, const float* det_masks, const int32_t* num_det,
                              int out_h, int out_w, float threshold, int order, uint8_t* bitmaps, uint32_t* counts,
                              int32_t* rle_len, int cap, hipStream_t st) {
  RleArgs a;
  a.masks = det_masks; a.num_det = num_det; a.bitmaps = bitmaps; a.counts = counts; a.rle_len = rle_len;
  a.max_dets = max_dets; a.ph = ph; a.pw = pw; a.out_h = out_h; a.out_w = out_w; a.cap = cap; a.order = order;
  a.threshold = threshold;
  a.scale_x = 1.0 / ((double)out_w / (double)pw);
  a.scale_y = 1.0 / ((double)out_h / (double)ph);
  hipLaunchKernelGGL(mask_rle_kernel, dim3(max_dets, batch), dim3(YP_RLE_THREADS), 0, st, a);
  return hipGetLastError();
}

int yp_rle_limits(int* max_src_h, int* max_out_h) {
  *max_src_h = YP_RLE_MAX_SRC_H;
  *max_out_h = YP_RLE_MAX_OUT_H;
  return 0;
}
