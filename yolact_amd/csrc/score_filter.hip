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
#include <stdlib.h>
#include "common.hpp"

struct ScoreFilterArgs {
  const float* cls;      // [B,A,C]
  u64* cand;             // cross-class: [B, ntiles*TA] tile-major slots; per-class: [B,nseg,A] lists
  int32_t* cand_count;   // per-class: [B,nseg] list lengths (atomic append)
  int32_t* tile_count;   // [B,ntiles] survivors of every tile (plain store; doubles as the slot count)
  int A, C, stride, nseg, ntiles;
  float conf, conf_lo;
  float pre_thresh;      // ln(1/conf - 1) + 1e-3: x_bg - x_fgmax must be below it to have p_fgmax > conf
  int ablate;            // TEMP debug
};

// ---- cross-lane helpers: DPP moves inside a row of 16 lanes (one VALU op, no LDS round trip) ----
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
#define YP_DPP_ROR(n) (0x120 + (n))   // row_ror:n  - lane j reads lane (j+n) mod 16 of its row
#define YP_DPP_XOR1 0xB1              // quad_perm:[1,0,3,2]
#define YP_DPP_XOR2 0x4E              // quad_perm:[2,3,0,1]

// YP-MATH tree over the 16 strided partial sums held by a row of 16 lanes: level 1 pairs (j, j+8),
// level 2 (j, j+4), level 3 (j, j+2), level 4 (j, j+1).  After level k the values are periodic with
// period 16 >> k, so a rotation by the period pairs exactly those operands (IEEE add commutes) and
// every lane ends up with the identical total.
__device__ __forceinline__ float grp16_sum(float v) {
  v = v + dpp_f<YP_DPP_ROR(8)>(v);
  v = v + dpp_f<YP_DPP_ROR(4)>(v);
  v = v + dpp_f<YP_DPP_ROR(2)>(v);
  v = v + dpp_f<YP_DPP_ROR(1)>(v);
  return v;
}
__device__ __forceinline__ float grp16_max(float v) {
  v = fmaxf(v, dpp_f<YP_DPP_ROR(8)>(v));
  v = fmaxf(v, dpp_f<YP_DPP_ROR(4)>(v));
  v = fmaxf(v, dpp_f<YP_DPP_ROR(2)>(v));
  v = fmaxf(v, dpp_f<YP_DPP_ROR(1)>(v));
  return v;
}
__device__ __forceinline__ int grp16_min(int v) {
  v = min(v, dpp_i<YP_DPP_ROR(8)>(v));
  v = min(v, dpp_i<YP_DPP_ROR(4)>(v));
  v = min(v, dpp_i<YP_DPP_ROR(2)>(v));
  v = min(v, dpp_i<YP_DPP_ROR(1)>(v));
  return v;
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// One wave = one tile of TA consecutive anchors (TA = 16, 32 or 64).
//   phase 0  the tile's TA*C logits, contiguous in HBM, go straight into LDS with LDS-DMA
//            (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR staging, all in flight)
//   phase 1  64/TA lanes per anchor: x_bg, max foreground logit, row max; exact-safe early reject:
//            p_fgmax <= 1/(1+exp(x_bg-x_fgmax)), so an anchor with x_bg-x_fgmax >= ln(1/conf-1)+1e-3
//            cannot pass `class_p_max > conf` (detection.py:73) and needs no exponentials at all
//   phase 2  the remaining anchors, four at a time, 16 lanes per anchor: lane j owns classes
//            j, j+16, ... - exactly the 16 strided partial sums of the YP-MATH softmax - so the
//            denominator is a per-lane sequential sum plus one DPP tree (8,4,2,1)
//   phase 3  survivors' keys go to the image's candidate slots / per-class lists (stores only; the
//            boxes are decoded later, for the few candidates that reach the NMS)
template <int MODE, int CPL, int TA>
__global__ __launch_bounds__(YP_WAVE) void score_filter_kernel(const ScoreFilterArgs p) {
  constexpr int LPA = YP_WAVE / TA;   // lanes per anchor in phase 1
  extern __shared__ __align__(16) float tile[];
  const int lane = threadIdx.x;
  const int b = blockIdx.y;
  const int a0 = blockIdx.x * TA;
  const int na = min(TA, p.A - a0);
  const int C = p.C;
  const int nelem = na * C;
  const float* __restrict__ src = p.cls + ((size_t)b * p.A + a0) * C;
  // small per-wave lists behind the tile
  int* s_list = reinterpret_cast<int*>(tile + TA * p.stride);   // tile-local anchors that passed phase 1
  u64* s_key = reinterpret_cast<u64*>(s_list + YP_WAVE);          // keys waiting to be written out
  constexpr int KCAP = 128;                                       // capacity of s_key

  // ---- phase 0: global -> LDS ------------------------------------------------------------------
  if (p.ablate & 4) {} else if (p.stride == C && (((uintptr_t)src) & 15u) == 0) {
    const int nvec = nelem >> 2;
    const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src);
    for (int v0 = 0; v0 < nvec; v0 += YP_WAVE) {
      if (v0 + lane < nvec)
        __builtin_amdgcn_global_load_lds((glb_void*)(src4 + v0 + lane), (lds_void*)(tile + 4 * v0), 16, 0, 0);
    }
    for (int e = (nvec << 2) + lane; e < nelem; e += YP_WAVE) tile[e] = src[e];
  } else {
    for (int e = lane; e < nelem; e += YP_WAVE) {
      const int a = e / C;
      tile[a * p.stride + (e - a * C)] = src[e];
    }
  }
  __syncthreads();   // (single wave) waits vmcnt(0): every DMA piece has landed

  // ---- phase 1: LPA lanes per anchor ----------------------------------------------------------
  const int al1 = lane / LPA, q = lane % LPA;
  float m;
  bool pre;
  {
    const float* row = tile + min(al1, na - 1) * p.stride;
    const float x0 = row[0];
    float xf = -INFINITY;
    for (int c0 = 1 + q; c0 < C; c0 += 16 * LPA) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = row[min(c0 + LPA * j, C - 1)];
#pragma unroll
      for (int j = 0; j < 16; ++j) xf = fmaxf(xf, v[j]);
    }
    if (LPA >= 2) xf = fmaxf(xf, dpp_f<YP_DPP_XOR1>(xf));
    if (LPA >= 4) xf = fmaxf(xf, dpp_f<YP_DPP_XOR2>(xf));
    m = fmaxf(x0, xf);
    pre = (al1 < na) && (q == 0) && ((x0 - xf) < p.pre_thresh);
  }
  const u64 lt_mask = (1ull << lane) - 1ull;
  const u64 pm = __ballot(pre);
  const int npass = (p.ablate & 2) ? 0 : ((p.ablate & 8) ? min(4, __popcll(pm)) : __popcll(pm));
  if (pre) s_list[__popcll(pm & lt_mask)] = al1;
  __syncthreads();

  // ---- phase 2: 16 lanes per anchor -----------------------------------------------------------
  const int grp = lane >> 4, j = lane & 15;
  int nk = 0;      // survivors so far (wave-uniform)
  int ncand = 0;   // per-class candidates parked in s_key (wave-uniform)
  for (int t0 = 0; t0 < npass; t0 += 4) {
    const int slot = t0 + grp;
    const bool act = slot < npass;
    const int al = act ? s_list[slot] : 0;
    const float ma = __shfl(m, al * LPA);
    const float* r = tile + al * p.stride;
    float e[CPL];
    float s = 0.0f, em = 0.0f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = j + 16 * i;
      const bool ok = c < C;
      const float x = r[ok ? c : 0];
      e[i] = ok ? yp_exp_le0(x - ma) : 0.0f;              // x <= row max
      s = s + e[i];                                        // s_j = e_j + e_{j+16} + ...
      if (i > 0) em = fmaxf(em, e[i]);
      else if (j > 0) em = e[0];                           // class 0 is background (detection.py:59)
    }
    const float S = grp16_sum(s);
    em = grp16_max(em);
    // IEEE division is monotone, so max_c fl(e_c/S) = fl(max_c e_c / S)                (detection.py:62)
    const float pmax = em / S;
    const bool surv = act && (pmax > p.conf);                                        // detection.py:73
    const bool lead = surv && j == 0;
    const u64 bl = __ballot(lead);
    const int pos = nk + __popcll(bl & lt_mask);
    if (MODE == YP_NMS_CROSS_CLASS) {
      // tf.argmax over the probabilities (detection.py:184): the FIRST class whose rounded
      // probability equals the maximum.  Only an e_c within 2^-21 of the largest can round to it.
      int arg = 0x7fffffff;
      if (surv) {
        const float ethr = em * (1.0f - 0x1p-21f);
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const int c = j + 16 * i;
          if (c >= 1 && c < C && e[i] >= ethr) {
            if (e[i] / S == pmax) arg = min(arg, c);
          }
        }
      }
      arg = grp16_min(arg);
      if (lead) s_key[pos] = yp_seg_key(pmax, a0 + al, arg);
    } else {
      // per-class candidates: every (anchor, class) with p > conf                   (detection.py:233)
      // They are parked in LDS and appended to the per-class lists 64 at a time, so that the
      // returning atomics of a whole tile are in flight together instead of one after another.
      const float elo = S * p.conf_lo;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = j + 16 * i;
        bool has = false;
        float pc = 0.0f;
        if (surv && c >= 1 && c < C && e[i] >= elo) {
          pc = e[i] / S;
          has = pc > p.conf;
        }
        const u64 bm = __ballot(has);
        if (bm) {
          if (has) s_key[ncand + __popcll(bm & lt_mask)] = yp_seg_key(pc, a0 + al, c);
          ncand += __popcll(bm);
          if (ncand > KCAP - YP_WAVE) {   // wave-uniform
            __syncthreads();
            for (int l = lane; l < ncand; l += YP_WAVE) {
              const u64 key = s_key[l];
              const size_t seg = (size_t)b * p.nseg + (yp_seg_key_class(key) - 1);
              const int qpos = atomicAdd(&p.cand_count[seg], 1);
              p.cand[seg * p.A + qpos] = key;
            }
            __syncthreads();
            ncand = 0;
          }
        }
      }
    }
    nk += __popcll(bl);
  }
  if (lane == 0) p.tile_count[(size_t)b * p.ntiles + blockIdx.x] = (p.ablate & 1) ? 0 : nk;
  if (nk == 0 || (p.ablate & 1)) return;
  __syncthreads();

  // ---- phase 3: survivors ---------------------------------------------------------------------
  if (MODE == YP_NMS_CROSS_CLASS) {
    if (lane < nk) p.cand[((size_t)b * p.ntiles + blockIdx.x) * TA + lane] = s_key[lane];
  } else {
    for (int l = lane; l < ncand; l += YP_WAVE) {
      const u64 key = s_key[l];
      const size_t seg = (size_t)b * p.nseg + (yp_seg_key_class(key) - 1);
      const int qpos = atomicAdd(&p.cand_count[seg], 1);
      p.cand[seg * p.A + qpos] = key;
    }
  }
}

template <int MODE, int TA>
static void launch_mode(int cpl, int A, int batch, size_t lds, hipStream_t st, const ScoreFilterArgs& a) {
  const dim3 grid((A + TA - 1) / TA, batch);
#define YP_SF_CASE(N) case N: hipLaunchKernelGGL((score_filter_kernel<MODE, N, TA>), grid, dim3(YP_WAVE), lds, st, a); break;
  switch (cpl) {
    YP_SF_CASE(1) YP_SF_CASE(2) YP_SF_CASE(3) YP_SF_CASE(4) YP_SF_CASE(5) YP_SF_CASE(6)
    YP_SF_CASE(8) YP_SF_CASE(12) YP_SF_CASE(16)
  }
#undef YP_SF_CASE
}

hipError_t yp_launch_score_filter(const yp_handle* h, int batch, const float* pred_cls, hipStream_t st) {
  ScoreFilterArgs a;
  a.cls = pred_cls;
  a.cand = h->s.cand; a.cand_count = h->s.cand_count; a.tile_count = h->s.tile_count;
  a.A = h->cfg.num_anchors; a.C = h->cfg.num_classes; a.stride = h->lds_stride; a.nseg = h->nseg;
  a.conf = h->cfg.conf_thresh;
  a.conf_lo = h->cfg.conf_thresh * (1.0f - 0x1p-20f);
  a.pre_thresh = h->pre_thresh;
  a.ablate = getenv("YP_ABLATE") ? atoi(getenv("YP_ABLATE")) : 0;
  const int TA = h->tile_anchors;
  a.ntiles = h->ntiles;
  const size_t lds = (size_t)TA * a.stride * sizeof(float) + YP_WAVE * sizeof(int) + 128 * sizeof(u64);
  int cpl = (a.C + 15) / 16;                       // classes per lane in phase 2
  if (cpl == 7) cpl = 8; else if (cpl > 8 && cpl <= 12) cpl = 12; else if (cpl > 12) cpl = 16;
  const bool cc = h->cfg.nms_mode == YP_NMS_CROSS_CLASS;
  if (TA == 64) { if (cc) launch_mode<YP_NMS_CROSS_CLASS, 64>(cpl, a.A, batch, lds, st, a); else launch_mode<YP_NMS_PER_CLASS, 64>(cpl, a.A, batch, lds, st, a); }
  else if (TA == 32) { if (cc) launch_mode<YP_NMS_CROSS_CLASS, 32>(cpl, a.A, batch, lds, st, a); else launch_mode<YP_NMS_PER_CLASS, 32>(cpl, a.A, batch, lds, st, a); }
  else { if (cc) launch_mode<YP_NMS_CROSS_CLASS, 16>(cpl, a.A, batch, lds, st, a); else launch_mode<YP_NMS_PER_CLASS, 16>(cpl, a.A, batch, lds, st, a); }
  return hipGetLastError();
}
