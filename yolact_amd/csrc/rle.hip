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
// One 1024-thread workgroup per detection walks the output image in column-major order, a batch of
// up to 64 columns at a time, every wave taking consecutive columns of the batch:
//   S  the source columns the batch needs (a [mask_h][~nc*scale+2] sub-block) are staged in LDS once,
//      by all threads: one global round trip per batch instead of one per column;
//   A  a column's source columns sx, sx+1 and weights are wave-uniform: the wave builds the
//      horizontally interpolated source column (mask_h values) in LDS - exactly OpenCV's row buffer;
//   B  64 output rows at a time: two LDS reads, the vertical interpolation (per-lane row indices and
//      weights live in registers), the threshold, one ballot -> 64 bits of the bitmap.  Run boundaries
//      are the set bits of m ^ (m << 1 | carry): counting and locating them is popcount / clz;
//   C  after a barrier every wave scans the batch's per-column boundary counts (one lane per column)
//      and so knows how many boundaries precede its columns and where the last one was; it writes its
//      run lengths straight to their final places: count j = position of boundary j minus position
//      of boundary j-1.  No second pass, no atomics.
// HBM traffic is the source mask (once) plus the counts (and the optional bitmap): the kernel is
// bound by instruction issue.
#include "common.hpp"

#define YP_RLE_THREADS 1024
#define YP_RLE_WAVES (YP_RLE_THREADS / YP_WAVE)
#define YP_RLE_MAX_COLS 64        // columns per batch (one lane per column in phase C)
#define YP_RLE_CELLS 1024         // 64-row chunks of bitmap a batch holds in LDS: ncols * ceil(out_h/64) <= this
#define YP_RLE_FAST_HB 16         // out_h <= 1024: per-lane vertical coefficients in registers
#define YP_RLE_MAX_SRC_H 512
#define YP_RLE_MAX_OUT_H 4096
#define YP_RLE_SRC_FLOATS 10240   // staged source sub-block (40 KiB at most)

struct RleArgs {
  const float* masks;       // [B,max_dets,ph,pw]
  const int32_t* num_det;   // [B]
  uint8_t* bitmaps;         // [B,max_dets,out_h,out_w] row-major 0/1, or null
  uint32_t* counts;         // [B,max_dets,cap]
  int32_t* rle_len;         // [B,max_dets]
  int max_dets, ph, pw, out_h, out_w, cap, order;
  int ncols;                // columns per batch: multiple of 16, <= 64, ncols * ceil(out_h/64) <= YP_RLE_CELLS
  int ws;                   // staged source columns per batch (0: not staged, phase A reads global memory)
  float threshold;
  double scale_x, scale_y;  // 1. / ((double)dst / src), as cv::resize computes them
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

__device__ __forceinline__ int wave_incl_add(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v += o; }
  return v;
}
__device__ __forceinline__ int wave_incl_max(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(v, d); if (lane >= d) v = max(v, o); }
  return v;
}

// Dynamic LDS, carved by the kernel (sizes from the launch arguments):
//   src   [ph][ws] float            staged source sub-block
//   hcol  [16][ph] float|int        OpenCV's row buffer, one per wave
//   bits  [YP_RLE_CELLS] u64        the batch's bitmap, [column][chunk]
//   trans [YP_RLE_CELLS] u64        its run boundaries (a column's first pixel excluded)
//   col_* [64] int                  per column: first bit, last bit, inner boundary count, last inner boundary position
//   fy/sy [out_h]                   vertical coefficients (only when out_h > 1024)
template <bool FAST>
__global__ __launch_bounds__(YP_RLE_THREADS) void mask_rle_kernel(const RleArgs p) {
  extern __shared__ __align__(16) char rle_lds[];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = blockIdx.x, b = blockIdx.y;
  const size_t slot = (size_t)b * p.max_dets + n;
  if (n >= p.num_det[b]) {                       // not a detection: an empty encoding
    if (t == 0) p.rle_len[slot] = 0;
    return;
  }
  const int HB = (p.out_h + 63) >> 6;
  char* lp_ = rle_lds;
  float* const s_src = reinterpret_cast<float*>(lp_);                 lp_ += (size_t)((p.ph * p.ws * 4 + 15) & ~15);
  float* const s_hcol = reinterpret_cast<float*>(lp_) + wave * p.ph;  lp_ += (size_t)((YP_RLE_WAVES * p.ph * 4 + 15) & ~15);
  u64* const s_bits = reinterpret_cast<u64*>(lp_);                    lp_ += YP_RLE_CELLS * 8;
  u64* const s_trans = reinterpret_cast<u64*>(lp_);                   lp_ += YP_RLE_CELLS * 8;
  int* const s_first = reinterpret_cast<int*>(lp_);                   lp_ += YP_RLE_MAX_COLS * 4;
  int* const s_lastbit = reinterpret_cast<int*>(lp_);                 lp_ += YP_RLE_MAX_COLS * 4;
  int* const s_count = reinterpret_cast<int*>(lp_);                   lp_ += YP_RLE_MAX_COLS * 4;
  int* const s_lastpos = reinterpret_cast<int*>(lp_);                 lp_ += YP_RLE_MAX_COLS * 4;
  float* const s_fy = reinterpret_cast<float*>(lp_);                  lp_ += FAST ? 0 : (size_t)p.out_h * 4;
  short* const s_sy = reinterpret_cast<short*>(lp_);
  int* const s_hcol_i = reinterpret_cast<int*>(s_hcol);

  const float* __restrict__ S = p.masks + slot * (size_t)p.ph * p.pw;
  uint32_t* const counts = p.counts + slot * (size_t)p.cap;
  uint8_t* const bitmap = p.bitmaps ? p.bitmaps + slot * (size_t)p.out_h * p.out_w : nullptr;
  const bool identity = p.ph == p.out_h && p.pw == p.out_w;     // cv::resize copies when the sizes agree
  const bool fixed = p.order == YP_THRESHOLD_THEN_RESIZE;
  const float one = 1.0f, cscale = 2048.0f;                      // INTER_RESIZE_COEF_SCALE
  const int phm1 = p.ph - 1;

  // vertical coefficients of this lane's output rows (weights kept, row indices clipped at use):
  // both clipped row indices packed in one register, the fractional weight in another
  int vr[YP_RLE_FAST_HB];
  float vf[YP_RLE_FAST_HB];
  if (FAST) {
#pragma unroll
    for (int c = 0; c < YP_RLE_FAST_HB; ++c) {
      const int y = c * 64 + lane;
      float f = (float)(((double)y + 0.5) * p.scale_y - 0.5);
      const int s = (int)floorf(f);
      vf[c] = f - (float)s;
      vr[c] = min(max(s, 0), phm1) | (min(max(s + 1, 0), phm1) << 16);
    }
  } else {
    for (int y = t; y < p.out_h; y += YP_RLE_THREADS) {
      float f = (float)(((double)y + 0.5) * p.scale_y - 0.5);
      const int s = (int)floorf(f);
      s_fy[y] = f - (float)s;
      s_sy[y] = (short)s;
    }
    __syncthreads();
  }

  const u64 lt_mask = (1ull << lane) - 1ull;
  const int cpw = p.ncols / YP_RLE_WAVES;        // columns per wave and batch
  int total = 0;              // boundaries before the current batch (identical in every thread)
  int last_pos = 0;           // position of the last boundary so far (0: none yet)
  int prev_lastbit = 0;       // bit of the last pixel of the previous column
  for (int x0 = 0; x0 < p.out_w; x0 += p.ncols) {
    const int nc = min(p.ncols, p.out_w - x0);   // columns in this batch
    // ---- S: stage the source columns this batch reads ----------------------------------------------
    int sx_lo = 0;
    if (p.ws > 0 && !identity) {
      float f_; bool r_;
      linear_coeff_x(x0, p.scale_x, p.pw, sx_lo, f_, r_);
      const int wcols = min(p.ws, p.pw - sx_lo);
      for (int idx = t; idx < p.ph * wcols; idx += YP_RLE_THREADS) {
        const int r = idx / wcols, c = idx - r * wcols;
        s_src[r * p.ws + c] = S[(size_t)r * p.pw + sx_lo + c];
      }
      __syncthreads();
    }
    for (int q = 0; q < cpw; ++q) {
      const int col = __builtin_amdgcn_readfirstlane(wave * cpw + q);   // column of the batch, wave-uniform
      if (col >= nc) break;
      const int x = x0 + col;
      const int col_start = x * p.out_h;
      // ---- A: horizontal pass for this column ------------------------------------------------------
      if (!identity) {
        int sx; float fx; bool right;
        linear_coeff_x(x, p.scale_x, p.pw, sx, fx, right);
        const float a0 = one - fx, a1 = fx;
        const int ia0 = (int)rintf(a0 * cscale), ia1 = (int)rintf(a1 * cscale);   // saturate_cast<short>: cvRound
        const int sx1 = min(sx + 1, p.pw - 1);
        for (int r = lane; r < p.ph; r += YP_WAVE) {
          float s0, s1;
          if (p.ws > 0) { s0 = s_src[r * p.ws + (sx - sx_lo)]; s1 = s_src[r * p.ws + (sx1 - sx_lo)]; }
          else { s0 = S[(size_t)r * p.pw + sx]; s1 = S[(size_t)r * p.pw + sx1]; }
          if (!fixed) {
            s_hcol[r] = right ? s0 * one : s0 * a0 + s1 * a1;
          } else {
            const int b0 = s0 > p.threshold ? 1 : 0, b1 = s1 > p.threshold ? 1 : 0;            // train.py:497
            s_hcol_i[r] = right ? b0 * 2048 : b0 * ia0 + b1 * ia1;
          }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): single wave, LDS in order
        __builtin_amdgcn_wave_barrier();
      }
      // ---- B: vertical pass, threshold, bits and boundaries --------------------------------------
      int cnt = 0, carry = 0, lastpos = -1;
      u64 m0 = 0ull;
      auto chunk = [&](int c, int r0, int r1, float f) {
        float w0, w1;
        if (!fixed) { w0 = one - f; w1 = f; }
        else { w0 = __int_as_float((int)rintf((one - f) * cscale)); w1 = __int_as_float((int)rintf(f * cscale)); }
        const int y = c * 64 + lane;
        bool bit = false;
        if (y < p.out_h) {
          if (identity) {
            bit = S[(size_t)y * p.pw + x] > p.threshold;
          } else if (!fixed) {
            const float v = s_hcol[r0] * w0 + s_hcol[r1] * w1;
            bit = v > p.threshold;                                                             // coco_eval.py:135
          } else {
            const int ib0 = __float_as_int(w0), ib1 = __float_as_int(w1);
            const int v = (((ib0 * (s_hcol_i[r0] >> 4)) >> 16) + ((ib1 * (s_hcol_i[r1] >> 4)) >> 16) + 2) >> 2;
            bit = (v & 0xff) != 0;
          }
        }
        const u64 m = __ballot(bit);
        const int nvalid = min(64, p.out_h - c * 64);
        const u64 vmask = nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1ull);
        // a boundary wherever a pixel differs from the one before it; the column's first pixel is
        // compared with the previous COLUMN's last pixel in phase C, so it is excluded here
        const u64 tr = (m ^ ((m << 1) | (u64)(c == 0 ? (int)(m & 1ull) : carry))) & vmask;
        carry = (int)((m >> (nvalid - 1)) & 1ull);
        cnt += __popcll(tr);
        if (tr != 0ull) lastpos = col_start + c * 64 + (63 - __clzll((long long)tr));
        if (c == 0) m0 = m;
        if (lane == 0) { s_bits[col * HB + c] = m; s_trans[col * HB + c] = tr; }
      };
      if (FAST) {
#pragma unroll
        for (int c = 0; c < YP_RLE_FAST_HB; ++c)
          if (c < HB) chunk(c, vr[c] & 0xffff, vr[c] >> 16, vf[c]);
      } else {
        for (int c = 0; c < HB; ++c) {
          const int y = min(c * 64 + lane, p.out_h - 1);
          const int s = s_sy[y];
          const float f = s_fy[y];
          const int r0 = min(max(s, 0), phm1), r1 = min(max(s + 1, 0), phm1);
          chunk(c, r0, r1, f);
        }
      }
      if (lane == 0) {
        s_first[col] = (int)(m0 & 1ull);
        s_lastbit[col] = carry;
        s_count[col] = cnt;
        s_lastpos[col] = lastpos;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);        // this column's hcol reads are done before the next column rewrites it
      __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- C: one lane per column: where do the batch's boundaries go? --------------------------------
    int c_first = 0, c_cnt = 0, c_cand = -1;
    {
      const bool have = lane < nc;
      const int cl = have ? lane : 0;
      const int fb = s_first[cl], pb = lane == 0 ? prev_lastbit : s_lastbit[max(cl - 1, 0)];
      c_first = have && (fb != pb) ? 1 : 0;                              // the column's first pixel starts a run
      c_cnt = have ? s_count[cl] : 0;
      const int lpz = s_lastpos[cl];
      c_cand = !have ? -1 : (lpz >= 0 ? lpz : (c_first ? (x0 + cl) * p.out_h : -1));   // last boundary inside column cl
    }
    const int tot = c_first + c_cnt;
    const int incl = wave_incl_add(tot, lane);
    const int excl = incl - tot;                                         // boundaries of the batch before column `lane`
    const int inclmax = wave_incl_max(c_cand, lane);
    int exclmax = __shfl_up(inclmax, 1);
    if (lane == 0) exclmax = -1;
    const int lp_before = max(last_pos, exclmax);                        // last boundary before column `lane`
    for (int q = 0; q < cpw; ++q) {
      const int col = __builtin_amdgcn_readfirstlane(wave * cpw + q);
      if (col >= nc) break;
      const int col_start = (x0 + col) * p.out_h;
      int off = total + __builtin_amdgcn_readlane(excl, col);
      int lp = __builtin_amdgcn_readlane(lp_before, col);
      if (__builtin_amdgcn_readlane(c_first, col)) {
        if (lane == 0 && off < p.cap) counts[off] = (uint32_t)(col_start - lp);
        off += 1;
        lp = col_start;
      }
      for (int c = 0; c < HB; ++c) {
        const u64 tr = s_trans[col * HB + c];
        if (tr == 0ull) continue;                // wave-uniform
        if ((tr >> lane) & 1ull) {
          const u64 below = tr & lt_mask;
          const int pos = col_start + c * 64 + lane;
          const int prev = below ? col_start + c * 64 + (63 - __clzll((long long)below)) : lp;
          const int j = off + __popcll(below);
          if (j < p.cap) counts[j] = (uint32_t)(pos - prev);
        }
        off += __popcll(tr);
        lp = col_start + c * 64 + (63 - __clzll((long long)tr));
      }
    }
    // ---- the row-major bitmap: nc bytes of every row per batch (train.py:502-506's detection_masks) ----
    if (bitmap) {
      for (int idx = t; idx < p.out_h * YP_RLE_MAX_COLS; idx += YP_RLE_THREADS) {
        const int y = idx >> 6, xc = idx & 63;
        if (xc < nc) bitmap[(size_t)y * p.out_w + x0 + xc] = (uint8_t)((s_bits[xc * HB + (y >> 6)] >> (y & 63)) & 1ull);
      }
    }
    total += __builtin_amdgcn_readlane(incl, 63);
    last_pos = max(last_pos, __builtin_amdgcn_readlane(inclmax, 63));
    prev_lastbit = s_lastbit[nc - 1];
    __syncthreads();                             // LDS is rewritten by the next batch
  }
  if (t == 0) {
    const int N = p.out_h * p.out_w;
    if (total < p.cap) counts[total] = (uint32_t)(N - last_pos);   // the run that reaches the end
    p.rle_len[slot] = total + 1;
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
  const int HB = (out_h + 63) / 64;
  int nc = YP_RLE_MAX_COLS;
  while (nc > 16 && nc * HB > YP_RLE_CELLS) nc -= 16;
  // staged source sub-block: columns sx(x0) .. sx(x0+nc-1)+1, at most floor(nc*scale)+3 of them
  auto width = [&](int ncols) { return (int)((double)ncols * a.scale_x) + 3; };
  while (nc > 16 && (long long)ph * width(nc) > YP_RLE_SRC_FLOATS) nc -= 16;
  a.ncols = nc;
  a.ws = (long long)ph * width(nc) <= YP_RLE_SRC_FLOATS ? (width(nc) < pw ? width(nc) : pw) : 0;
  const bool fast = HB <= YP_RLE_FAST_HB;
  static bool attr_set = false;
  if (!attr_set) {   // up to ~113 KiB of dynamic LDS at the largest supported sizes
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_rle_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mask_rle_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  size_t lds = (size_t)((ph * a.ws * 4 + 15) & ~15) + (size_t)((YP_RLE_WAVES * ph * 4 + 15) & ~15) + 2 * YP_RLE_CELLS * 8 +
               4 * YP_RLE_MAX_COLS * 4 + (fast ? 0 : (size_t)out_h * 6 + 16);
  if (fast) hipLaunchKernelGGL((mask_rle_kernel<true>), dim3(max_dets, batch), dim3(YP_RLE_THREADS), lds, st, a);
  else hipLaunchKernelGGL((mask_rle_kernel<false>), dim3(max_dets, batch), dim3(YP_RLE_THREADS), lds, st, a);
  return hipGetLastError();
}

int yp_rle_limits(int* max_src_h, int* max_out_h) {
  *max_src_h = YP_RLE_MAX_SRC_H;
  *max_out_h = YP_RLE_MAX_OUT_H;
  return 0;
}
