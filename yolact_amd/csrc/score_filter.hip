// Kernel 1: class softmax + confidence filter + anchor box decode + candidate compaction.
//
// Replaces detection.py:56-62 (softmax, drop background, per-anchor max), :73-76 (the four
// tf.boolean_mask passes), :79 -> utils/utils.py:173-213 (_decode) and, for the NMS that follows,
// detection.py:184-185 (argmax / reduce_max over classes).
//
// HBM-bound: the [B,A,C] logits are read exactly once (4*A*C bytes per image).  One wave owns a
// tile of 64 consecutive anchors: the tile's 64*C floats are contiguous in memory, so the wave
// streams them with coalesced 16-byte loads into LDS and then works lane-per-anchor on LDS rows
// (row stride C is odd for C=81 -> conflict-free ds_read_b32).  Offsets and priors are only
// touched for the few percent of anchors that survive.
#include "common.hpp"

struct ScoreFilterArgs {
  const float* cls;      // [B,A,C]
  const float* off;      // [B,A,4]
  const float* priors;   // [A,4]
  u64* cand;             // [B,nseg,A]
  int32_t* cand_count;   // [B,nseg]
  int32_t* surv_count;   // [B]
  float4* dec_box;       // [B,A]
  int A, C, stride, nseg;
  float conf, conf_lo;
};

// One row of tf.nn.softmax's numerator/denominator in YP-MATH order.  On return row[c] holds
// e_c = exp(x_c - max) for every class and the function yields S = sum_c e_c.
__device__ __forceinline__ float softmax_row_inplace(float* row, int C) {
  float m = row[0];
  for (int c = 1; c < C; ++c) m = fmaxf(m, row[c]);
  float s[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) s[j] = 0.0f;
  for (int c0 = 0; c0 < C; c0 += 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = c0 + j;
      if (c < C) {
        const float e = yp_exp(row[c] - m);
        row[c] = e;
        s[j] = s[j] + e;
      }
    }
  }
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1) {
#pragma unroll
    for (int j = 0; j < w; ++j) s[j] = s[j] + s[j + w];
  }
  return s[0];
}

template <int MODE>
__global__ __launch_bounds__(YP_WAVE) void score_filter_kernel(const ScoreFilterArgs p) {
  extern __shared__ __align__(16) float tile[];
  const int lane = threadIdx.x;
  const int b = blockIdx.y;
  const int a0 = blockIdx.x * YP_WAVE;
  const int na = min(YP_WAVE, p.A - a0);
  const int C = p.C;
  const int nelem = na * C;
  const float* __restrict__ src = p.cls + ((size_t)b * p.A + a0) * C;

  // ---- global -> LDS, coalesced -----------------------------------------------------------
  if (p.stride == C && (((uintptr_t)src) & 15u) == 0) {
    const int nvec = nelem >> 2;
    const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src);
    float4* dst4 = reinterpret_cast<float4*>(tile);
#pragma unroll 8
    for (int v = lane; v < nvec; v += YP_WAVE) dst4[v] = src4[v];
    for (int e = (nvec << 2) + lane; e < nelem; e += YP_WAVE) tile[e] = src[e];
  } else {
    for (int e = lane; e < nelem; e += YP_WAVE) {
      const int a = e / C;
      tile[a * p.stride + (e - a * C)] = src[e];
    }
  }
  __syncthreads();

  // ---- lane per anchor ----------------------------------------------------------------------
  const int a = a0 + lane;
  bool surv = false;
  float pmax = 0.0f, emax = 0.0f, S = 1.0f;
  int arg = 0;
  float* row = tile + lane * p.stride;
  if (lane < na) {
    S = softmax_row_inplace(row, C);
    emax = row[1];
    for (int c = 2; c < C; ++c) emax = fmaxf(emax, row[c]);
    // IEEE division is monotone, so max_c fl(e_c/S) = fl(max_c e_c / S)   (detection.py:62)
    pmax = emax / S;
    surv = pmax > p.conf;                                                     // detection.py:73
  }

  if (MODE == YP_NMS_CROSS_CLASS) {
    if (surv) {
      // tf.argmax over the probabilities (detection.py:184): the FIRST class whose rounded
      // probability equals the maximum.  Only an e_c within 2^-21 of the largest can round to it.
      const float ethr = emax * (1.0f - 0x1p-21f);
      for (int c = 1; c < C; ++c) {
        const float e = row[c];
        if (arg == 0 && e >= ethr) {
          if (e / S == pmax) arg = c;
        }
      }
    }
    const u64 ball = __ballot(surv);
    if (ball) {
      int base = 0;
      const int leader = __ffsll((long long)ball) - 1;
      if (lane == leader) base = atomicAdd(&p.cand_count[b], __popcll(ball));
      base = __shfl(base, leader);
      if (surv) {
        const int pos = base + __popcll(ball & ((1ull << lane) - 1ull));
        p.cand[(size_t)b * p.A + pos] = yp_seg_key(pmax, a, arg);
      }
    }
  } else {
    // per-class candidates: every (anchor, class) with p > conf                (detection.py:233)
    if (__ballot(surv)) {
      const float elo = S * p.conf_lo;
      for (int c = 1; c < C; ++c) {
        const float e = row[c];
        if (surv && e >= elo) {
          const float pc = e / S;
          if (pc > p.conf) {
            const size_t seg = (size_t)b * p.nseg + (c - 1);
            const int pos = atomicAdd(&p.cand_count[seg], 1);
            p.cand[seg * p.A + pos] = yp_seg_key(pc, a, c);
          }
        }
      }
    }
  }

  // ---- decode the survivors (utils/utils.py:193-213) ---------------------------------------
  const u64 sball = __ballot(surv);
  if (sball) {
    if (lane == __ffsll((long long)sball) - 1) atomicAdd(&p.surv_count[b], __popcll(sball));
    if (surv) {
      const float4 pr = reinterpret_cast<const float4*>(p.priors)[a];                       // y0,x0,y1,x1
      const float4 of = reinterpret_cast<const float4*>(p.off)[(size_t)b * p.A + a];        // dx,dy,dw,dh
      const float ph = pr.z - pr.x;
      const float pw = pr.w - pr.y;
      const float cx = pr.y + (pw / 2.0f);
      const float cy = pr.x + (ph / 2.0f);
      const float bx = cx + (of.x * pw) * 0.1f;
      const float by = cy + (of.y * ph) * 0.1f;
      const float bw = pw * yp_exp(of.z * 0.2f);
      const float bh = ph * yp_exp(of.w * 0.2f);
      float4 box;
      box.y = bx - bw / 2.0f;   // xmin
      box.x = by - bh / 2.0f;   // ymin
      box.w = bw / 2.0f + bx;   // xmax
      box.z = bh / 2.0f + by;   // ymax
      p.dec_box[(size_t)b * p.A + a] = box;
    }
  }
}

hipError_t yp_launch_score_filter(const yp_handle* h, int batch, const float* pred_cls, const float* pred_offset,
                                  const float* priors, hipStream_t st) {
  ScoreFilterArgs a;
  a.cls = pred_cls; a.off = pred_offset; a.priors = priors;
  a.cand = h->s.cand; a.cand_count = h->s.cand_count; a.surv_count = h->s.surv_count; a.dec_box = h->s.dec_box;
  a.A = h->cfg.num_anchors; a.C = h->cfg.num_classes; a.stride = h->lds_stride; a.nseg = h->nseg;
  a.conf = h->cfg.conf_thresh;
  a.conf_lo = h->cfg.conf_thresh * (1.0f - 0x1p-20f);
  const dim3 grid((a.A + YP_WAVE - 1) / YP_WAVE, batch);
  const size_t lds = (size_t)YP_WAVE * a.stride * sizeof(float);
  if (h->cfg.nms_mode == YP_NMS_CROSS_CLASS)
    hipLaunchKernelGGL(score_filter_kernel<YP_NMS_CROSS_CLASS>, grid, dim3(YP_WAVE), lds, st, a);
  else
    hipLaunchKernelGGL(score_filter_kernel<YP_NMS_PER_CLASS>, grid, dim3(YP_WAVE), lds, st, a);
  return hipGetLastError();
}
