// Kernel 1: class softmax + confidence filter + candidate compaction.
//
// Replaces detection.py:56-62 (softmax, drop background, per-anchor max), :73-76 (the four
// tf.boolean_mask passes) and, for the NMS that follows, detection.py:184-185 (argmax / reduce_max
// over classes).  The boxes (utils._decode, utils/utils.py:173-213) are decoded later, by the NMS
// kernels, for the few candidates that reach them.
//
// HBM-bound: the [B,A,C] logits are read exactly once (4*A*C bytes per image) with non-temporal
// 16-byte loads.  The batch is cut into `nitems` work items, `parts` per image, each a run of
// consecutive 32-anchor tiles of ONE image; one single-wave workgroup owns an item from start to
// end.  The item count is chosen so that all items are resident at once (one round, no tail).
// What an item emits is parked in LDS and leaves - normally once, at the item's end - for the lists the
// NMS launch reads: YP_NLISTS lists per image (per-class mode: one per class group, so that a group's
// workgroup finds exactly its candidates; cross-class mode: the image's items dealt round-robin), every
// entry a 64-bit key plus the candidate's DECODED BOX (prior and offsets gathered here, utils._decode
// evaluated once), so that the NMS workgroups get everything they need in one round trip and never
// gather from the cold pred_offset array.  The slots come from one agent-scope atomic per (flush, list),
// issued back to back; every list counter sits on a cache line of its own.
#include <stdlib.h>
#include "common.hpp"

struct ScoreFilterArgs {
  const float* cls;      // [B,A,C]
  const float4* off;     // [B,A] pred_offset (dx,dy,dw,dh)
  const float4* priors;  // [A]   (ymin,xmin,ymax,xmax) pixels
  u64* lkeys;            // [B][YP_NLISTS][lcap] keys (yp_seg_key), see common.hpp yp_scratch
  float4* lbox;          // [B][YP_NLISTS][lcap] decoded boxes
  int32_t* lcount;       // [B][YP_NLISTS] x YP_CNT_STRIDE
  u64* okeys;            // [B][ocap] keys that found their list full (per-class mode)
  int32_t* ocount;       // [B] x YP_CNT_STRIDE
  int32_t* seg_surv;     // [nitems] anchors past the confidence filter
  int A, C, stride, ntiles, parts, nitems, ngroups;
  long long lcap, ocap;
  int bufb;              // bytes of the LDS tile buffer
  int fast;              // 1: stride == C and every tile starts 16-byte aligned -> wide-load path
  float conf, conf_lo;
  int32_t* overflow;     // [1] set when a key found both its list and the image's overflow list full
  float pre_thresh;      // ln(1/conf - 1) + 1e-3: x_bg - x_fgmax must be below it to have p_fgmax > conf
  float jen_thr, jen_n, jen_inv_n;   // second reject: (1/conf)(1+1e-3) + 0.1 (NaN = off), C-2, 1/(C-2)
};

// ---- cross-lane helpers: DPP moves inside a row of 16 lanes (one VALU op, no LDS round trip) ----
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
#define YP_DPP_XOR1 0xB1              // quad_perm:[1,0,3,2]

#define YP_DPP_XOR2 0x4E              // quad_perm:[2,3,0,1]
// lane j of a group of 8 (half a DPP row) reads lane j ^ 4: row_shl:4 for the lower four lanes of
// every group (banks 0 and 2 of the row), row_shr:4 for the upper four (banks 1 and 3)
__device__ __forceinline__ int xor4_i(int v) {
  int t = __builtin_amdgcn_update_dpp(0, v, 0x104, 0xF, 0x5, false);
  return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);
}
__device__ __forceinline__ float xor4_f(float v) { return __builtin_bit_cast(float, xor4_i(__builtin_bit_cast(int, v))); }

// YP-MATH tree over the 16 strided partial sums of one anchor, held two per lane by a group of 8
// lanes: lane j has s_j (classes j, j+16, ...) and s_{j+8} (classes j+8, j+24, ...).  Level 1 pairs
// (j, j+8): in the lane.  Level 2 (j, j+4), level 3 (j, j+2), level 4 (j, j+1): after level k the
// values are periodic with period 16 >> k, so exchanging with lane j^4, j^2, j^1 pairs exactly those
// operands (IEEE add commutes) and every lane ends up with the identical total.
__device__ __forceinline__ float grp8_sum(float s_lo, float s_hi) {
  float v = s_lo + s_hi;
  v = v + xor4_f(v);
  v = v + dpp_f<YP_DPP_XOR2>(v);
  v = v + dpp_f<YP_DPP_XOR1>(v);
  return v;
}
// max of three floats in one instruction (no NaN canonicalisation passes; logits are finite)
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// max over a group of 8 lanes of NON-NEGATIVE floats: their bit patterns order like integers
__device__ __forceinline__ float grp8_max_nonneg(float f) {
  int v = __builtin_bit_cast(int, f);
  v = max(v, xor4_i(v));
  v = max(v, dpp_i<YP_DPP_XOR2>(v));
  v = max(v, dpp_i<YP_DPP_XOR1>(v));
  return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float max_nonneg(float a, float b) {
  return __builtin_bit_cast(float, max(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b)));
}
__device__ __forceinline__ int grp8_sum_i(int v) {
  v = v + xor4_i(v);
  v = v + dpp_i<YP_DPP_XOR2>(v);
  v = v + dpp_i<YP_DPP_XOR1>(v);
  return v;
}
__device__ __forceinline__ int grp8_min(int v) {
  v = min(v, xor4_i(v));
  v = min(v, dpp_i<YP_DPP_XOR2>(v));
  v = min(v, dpp_i<YP_DPP_XOR1>(v));
  return v;
}

#ifndef YP_SF_AUX
#define YP_SF_AUX 2        // cache policy of the staging loads: 2 = nt (read once, by nobody else)
#endif
#ifndef YP_SF_MIN_WAVES
#define YP_SF_MIN_WAVES 1   // waves per SIMD the register allocation must allow
#endif
#define YP_SF_KCAP 224   // keys the LDS hand-off buffer holds (with the tile of C = 81 the wave still takes ten
                         // 1280-byte LDS units: 12 waves per CU); an ordinary item (~40 keys) flushes once, at its end

// One wave walks the tiles of its item.  Per tile of TA = 32 consecutive anchors (TA*C contiguous
// floats in HBM):
//   phase 0  the tile that was loaded into VGPRs during the previous iteration (NLD 16-byte loads
//            per lane) is written to the wave's single LDS tile buffer, and the loads of the NEXT
//            tile are issued straight away: they are in flight while this tile is processed, and a
//            wave needs LDS for one tile only (12 waves per CU, each with ~10 KiB in flight)
//   phase 1  two lanes per anchor on LDS rows (row stride C is odd for C = 81 -> conflict-free):
//            x_bg, max foreground logit, row max; exact-safe early reject:
//            p_fgmax <= 1/(1+exp(x_bg-x_fgmax)), so an anchor with x_bg-x_fgmax >= ln(1/conf-1)+1e-3
//            cannot pass `class_p_max > conf` (detection.py:73) and needs no exponentials at all
//   phase 2  the remaining anchors, eight at a time, 8 lanes per anchor: lane j owns classes
//            j, j+8, j+16, ... and keeps two running sums (even and odd terms) - exactly the strided
//            partial sums s_j and s_{j+8} of the YP-MATH softmax - so the denominator is two per-lane
//            sequential sums plus one tree ((j,j+8) in the lane, then j^4, j^2, j^1 by DPP).  Eight
//            anchors per round amortise the trees, the division and the bookkeeping over twice as
//            many anchors as sixteen lanes per anchor did: the kernel is bound by instruction issue.
//   phase 3  keys collect in LDS and leave - key + decoded box - for the image's lists when the buffer
//            fills up and at the item's end
// FAST (compile time, so that no path exists on which staged loads are pending and nobody stores them):
// rows are dense in LDS (stride == C) and the logits 16-byte aligned - the register-staged pipeline;
// otherwise a plain element-wise tile copy.
template <int MODE, int CPL, bool FAST>
__global__ __launch_bounds__(YP_WAVE, YP_SF_MIN_WAVES) void score_filter_kernel(const ScoreFilterArgs p) {
  constexpr int TA = YP_SF_TA;
  constexpr int LPA = YP_WAVE / TA;   // lanes per anchor in phase 1
  constexpr int NLD = (TA * CPL + 31) / 32;   // 1 KiB wave-loads that cover TA rows of up to 8*CPL floats
  typedef float nat4 __attribute__((ext_vector_type(4)));
  extern __shared__ __align__(16) char lds_raw[];
  const int lane = threadIdx.x;
  const int C = p.C;
  float* const tile = reinterpret_cast<float*>(lds_raw);
  int2* const s_list = reinterpret_cast<int2*>(lds_raw + p.bufb);   // (tile-local anchor, row max) of anchors past phase 1
  u64* const s_key = reinterpret_cast<u64*>(s_list + YP_WAVE);      // keys waiting to be written out
  const int grp = lane >> 3, j = lane & 7;
  int cidx[CPL];     // this lane's classes j, j+8, ... (clamped) and whether they exist
  float okf[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    cidx[i] = min(j + 8 * i, C - 1);
    okf[i] = (j + 8 * i) < C ? 1.0f : 0.0f;
  }
  const u64 lt_mask = (1ull << lane) - 1ull;
  const float ginv = 1.0f / (float)max(p.ngroups, 1);
  const int lane16 = lane * 16;
  const long long img_bytes = (long long)p.A * C * 4;
  const long long tile_bytes = (long long)TA * C * 4;

  nat4 stg[NLD];
  // The tile's bytes as NLD non-temporal 16-byte buffer loads per lane (read once, by nobody else) through
  // a descriptor of exactly the tile: lanes past its end get zeros from the range check - they land in LDS
  // nobody reads - and with `on` false (no next tile) the descriptor has no records, so the same
  // instructions fetch nothing.  The loads are thereby unconditional: one register set, no copies, and
  // the compiler's wait for them sits at the LDS stores of the next iteration and nowhere else.
  // The whole per-load offset (lane16 + u KiB) travels as the VGPR offset (the compiler folds what fits
  // into the instruction's 12-bit immediate): the hardware range-checks VGPR offset + immediate only, an
  // SGPR offset is outside the check, and with it loads u >= 1 of a short last tile would read past the
  // end of pred_cls (tests/test_abi.py::test_score_filter_buffer_loads_keep_their_offset_inside_the_range_check pins the emitted form).
  typedef unsigned int raw4 __attribute__((__vector_size__(4 * sizeof(unsigned int))));
  auto issue_loads = [&](int bb, int tt, bool on) {
    const int nbytes = on ? min(TA, p.A - tt * TA) * C * 4 : 0;
    const char* g = reinterpret_cast<const char*>(p.cls) + (bb * img_bytes + tt * tile_bytes);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g), 0, nbytes, 0x00020000);
#pragma unroll
    for (int u = 0; u < NLD; ++u)
      stg[u] = __builtin_bit_cast(nat4, (raw4)__builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16 + u * 1024, 0, YP_SF_AUX));
  };

  for (int item = blockIdx.x; item < p.nitems; item += gridDim.x) {
    const int b = item / p.parts, part = item - b * p.parts;
    const int t_begin = (int)(((long long)part * p.ntiles) / p.parts);
    const int t_end = (int)(((long long)(part + 1) * p.ntiles) / p.parts);
    int nbuf = 0;    // keys parked in s_key
    int nsurv = 0;   // anchors past the confidence filter
    // The parked keys leave for the image's lists.  Slots come from ONE atomic instruction per flush, however many
    // keys are parked: lane i first counts the parked keys of list i (ballots), then lanes 0..7 add the totals of
    // lists 2i and 2i+1 to the two halves of one 64-bit counter word.  A wave's returning agent-scope atomics are
    // served one after the other, ~0.2-0.5 us each (tools/atomic_probe.hip), so what a flush costs is the number
    // of active lanes in its atomic instructions: one instruction per list cost 8 us per launch at config 2 and 30
    // at config 4; one instruction per 64 keys with a lane per list 3 us and ~25; this form at most 8 lanes per
    // flush.  (Packing the counters closer - 4 to 64 bytes apart - changed nothing: the cost is per lane, not per
    // line.)  The keys then go out 64 at a time: every lane takes one key, fetches its anchor's prior and offsets -
    // the first batch's gather is issued BEFORE the atomic, the next batch's before the current one is stored -
    // decodes the box and stores key and box at its list's running base + its rank among the batch's keys of
    // that list.
    struct Batch { bool has; u64 key; float4 pr, of; int lst; };
    auto batch_issue = [&](int l0) -> Batch {   // keys s_key[l0 .. min(l0 + 64, nbuf)); no wait inside
      Batch q;
      q.has = l0 + lane < nbuf;
      q.key = s_key[min(l0 + lane, nbuf - 1)];
      const int anchor = min(yp_seg_key_anchor(q.key), p.A - 1);
      q.pr = p.priors[anchor];
      q.of = p.off[(size_t)b * p.A + anchor];
      if (MODE == YP_NMS_CROSS_CLASS) {
        q.lst = part & (YP_NLISTS - 1);
      } else {
        // (c - 1) % ngroups without the integer division: (x + 0.5) / n is at least 0.5 / n away from an
        // integer for 0 <= x < 256, n <= 16 - far more than the float error - so the truncation is x / n
        const int c1 = yp_seg_key_class(q.key) - 1;
        q.lst = c1 - (int)(((float)c1 + 0.5f) * ginv) * p.ngroups;
      }
      return q;
    };
    auto batch_commit = [&](const Batch& q, int slot) {
      const float4 box = yp_decode(q.pr, q.of);                                           // detection.py:79
      if (q.has) {
        if (slot < p.lcap) {
          const size_t o = ((size_t)b * YP_NLISTS + q.lst) * p.lcap + slot;
          p.lkeys[o] = q.key;
          p.lbox[o] = box;
        } else {
          // the list is full (only a batch far from anything a detector emits gets here): the key alone goes
          // to the image's overflow list, which every class group then also reads
          const int os = p.ocap > 0 ? __hip_atomic_fetch_add(p.ocount + (size_t)b * YP_CNT_STRIDE, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : 0x7fffffff;
          if (os < p.ocap) p.okeys[(size_t)b * p.ocap + os] = q.key;
          else *p.overflow = 1;   // (bounded lists only: the key is dropped)
        }
      }
    };
    auto flush = [&]() {   // wave-uniform: everything parked leaves now, and the wave waits for it
      __builtin_amdgcn_wave_barrier();
      if (nbuf == 0) return;
      int tot = 0;         // lane i < YP_NLISTS: parked keys of list i
      if (MODE == YP_NMS_CROSS_CLASS) {
        if (lane == (part & (YP_NLISTS - 1))) tot = nbuf;
      } else {
        for (int l0 = 0; l0 < nbuf; l0 += YP_WAVE) {
          const int c1 = yp_seg_key_class(s_key[min(l0 + lane, nbuf - 1)]) - 1;
          const int lst = l0 + lane < nbuf ? c1 - (int)(((float)c1 + 0.5f) * ginv) * p.ngroups : -1;
          for (int i = 0; i < p.ngroups; ++i) {
            const u64 mm = __ballot(lst == i);
            if (lane == i) tot += __popcll(mm);
          }
        }
      }
      Batch q = batch_issue(0);
      const int t_lo = __shfl(tot, (2 * lane) & 63), t_hi = __shfl(tot, (2 * lane + 1) & 63);
      u64 got = 0;
      if (lane < YP_NLISTS / 2 && (t_lo | t_hi) != 0)
        got = __hip_atomic_fetch_add(reinterpret_cast<u64*>(p.lcount + (size_t)b * YP_NLISTS * YP_CNT_STRIDE + YP_CNT_IDX(2 * lane)),
                                     (u64)(uint32_t)t_lo | ((u64)(uint32_t)t_hi << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int g_lo = __shfl((int)(uint32_t)got, lane >> 1), g_hi = __shfl((int)(uint32_t)(got >> 32), lane >> 1);
      int run = (lane & 1) ? g_hi : g_lo;   // lane i < YP_NLISTS: the next free slot of list i
      for (int l0 = 0; l0 < nbuf; l0 += YP_WAVE) {
        Batch nx = q;
        if (l0 + YP_WAVE < nbuf) nx = batch_issue(l0 + YP_WAVE);
        int rank = 0, add = 0;
        if (MODE == YP_NMS_CROSS_CLASS) {
          const u64 mm = __ballot(q.has);
          rank = __popcll(mm & lt_mask);
          if (lane == q.lst) add = __popcll(mm);   // (q.lst is the same in every lane)
        } else {
          for (int i = 0; i < p.ngroups; ++i) {
            const u64 mm = __ballot(q.has && q.lst == i);
            if (q.lst == i) rank = __popcll(mm & lt_mask);
            if (lane == i) add = __popcll(mm);
          }
        }
        const int slot = __shfl(run, q.lst) + rank;
        run += add;
        batch_commit(q, slot);
        q = nx;
      }
      nbuf = 0;
    };
    // (Tried: the keys of all tiles but the last issued - atomics, gather - when the last tile's processing starts and
    //  committed when it ends, so that only the last tile's handful waits at the item's end: 55.3 us per launch at
    //  config 2 against 51.4 with the one flush at the end; flushing synchronously before the last tile: 52.7.  For
    //  reference, with the per-batch atomics of that time: 44.7 with no flush at all, 48.4 with one atomic lane per
    //  batch instead of one per list.)
    if (FAST) issue_loads(b, t_begin, t_begin < t_end);

    for (int tix = t_begin; tix < t_end; ++tix) {
      const int a0 = tix * TA;
      const int na = min(TA, p.A - a0);
      // ---- phase 0 ------------------------------------------------------------------------------
      if (FAST) {
        // all NLD registers, unconditionally (the buffer has NLD KiB): a store the compiler cannot see
        // happen would leave it unsure whether the load behind the register has landed, and it would
        // then wait in phase 1 - on the loads just issued for the next tile
        nat4* const t4 = reinterpret_cast<nat4*>(tile);
#pragma unroll
        for (int u = 0; u < NLD; ++u) t4[u * YP_WAVE + lane] = stg[u];
        issue_loads(b, min(tix + 1, t_end - 1), tix + 1 < t_end);
      } else {
        const int nelem = na * C;
        const float* __restrict__ src = p.cls + ((size_t)b * p.A + a0) * C;
        for (int e = lane; e < nelem; e += YP_WAVE) {
          const int a = e / C;
          tile[a * p.stride + (e - a * C)] = src[e];
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();   // single wave: LDS operations execute in order

      // ---- phase 1: LPA lanes per anchor --------------------------------------------------------
      const int al1 = lane / LPA, q = lane % LPA;
      float m;
      bool pre;
      {
        const float* row = tile + min(al1, na - 1) * p.stride;
        const float x0 = row[0];
        float xf = -INFINITY;
        yp_f2 s2 = {0.0f, 0.0f};                         // sum of this lane's foreground logits (for the second reject)
        int c0 = 1 + q;
        for (; c0 + LPA * 15 < C; c0 += 16 * LPA) {     // full chunks: base + immediate offsets
          float v[16];
#pragma unroll
          for (int jj = 0; jj < 16; ++jj) v[jj] = row[c0 + LPA * jj];
          const float t0 = max3f(v[0], v[1], v[2]), t1 = max3f(v[3], v[4], v[5]), t2 = max3f(v[6], v[7], v[8]);
          const float t3 = max3f(v[9], v[10], v[11]), t4 = max3f(v[12], v[13], v[14]);
          xf = max3f(max3f(t0, t1, t2), max3f(t3, t4, v[15]), xf);
#pragma unroll
          for (int jj = 0; jj < 16; jj += 2) s2 = s2 + (yp_f2){v[jj], v[jj + 1]};
        }
        float dup = 0.0f;                               // the tail's clamped reads, all of class C-1, counted once too often
        if (c0 < C) {                                   // ragged tail, clamped
          float v[16];
#pragma unroll
          for (int jj = 0; jj < 16; ++jj) v[jj] = row[min(c0 + LPA * jj, C - 1)];
          const float t0 = max3f(v[0], v[1], v[2]), t1 = max3f(v[3], v[4], v[5]), t2 = max3f(v[6], v[7], v[8]);
          const float t3 = max3f(v[9], v[10], v[11]), t4 = max3f(v[12], v[13], v[14]);
          xf = max3f(max3f(t0, t1, t2), max3f(t3, t4, v[15]), xf);
#pragma unroll
          for (int jj = 0; jj < 16; jj += 2) s2 = s2 + (yp_f2){v[jj], v[jj + 1]};
          dup = (float)(16 - (C - c0 + LPA - 1) / LPA) * v[15];
        }
        float sfg = (s2.x + s2.y) - dup;
        xf = fmaxf(xf, dpp_f<YP_DPP_XOR1>(xf));
        sfg = sfg + dpp_f<YP_DPP_XOR1>(sfg);
        if (LPA >= 4) {
          xf = fmaxf(xf, dpp_f<YP_DPP_XOR2>(xf));
          sfg = sfg + dpp_f<YP_DPP_XOR2>(sfg);
        }
        m = fmaxf(x0, xf);
        pre = (al1 < na) && (q == 0) && ((x0 - xf) < p.pre_thresh);
        // Second exact-safe reject (Jensen): the C-2 foreground classes other than the best one have
        // sum exp(x_c) >= (C-2) exp(mean x_c), so 1/p_fgmax >= R = 1 + exp(x_bg-xf) + (C-2) exp(mean-xf),
        // and R >= 1/conf (plus margins, yp_create) means no class of the anchor can pass.  The margins
        // cover the float sum behind `mean` for |xf| <= 128 (its error grows with the logits' size).
        // jen_thr is NaN when the test is off: the comparison is then false.
        {
          const float mu_f = (sfg - xf) * p.jen_inv_n - xf;                       // mean of the others, minus xf (<= 0)
          const float r = 1.0f + __builtin_amdgcn_exp2f((x0 - xf) * 1.44269504f) + p.jen_n * __builtin_amdgcn_exp2f(mu_f * 1.44269504f);
          pre = pre && !((r >= p.jen_thr) && (fabsf(xf) <= 128.0f));
        }
      }
      const u64 pm = __ballot(pre);
      const int npass = __popcll(pm);
      if (pre) s_list[__popcll(pm & lt_mask)] = make_int2(al1, __float_as_int(m));   // (tile-local anchor, row max)
      __builtin_amdgcn_wave_barrier();

      // ---- phase 2: 16 lanes per anchor ---------------------------------------------------------
      // Straight-line on purpose: the CPL logits of a lane are fetched together, the CPL exponentials
      // are independent chains the scheduler can interleave, invalid class slots are multiplied by 0
      // instead of being branched around.
      for (int t0 = 0; t0 < npass; t0 += 8) {
        const int slot = t0 + grp;
        const bool act = slot < npass;
        const int2 ent = s_list[min(slot, npass - 1)];
        const int al = ent.x;
        const float ma = __int_as_float(ent.y);
        const float* r = tile + al * p.stride;
        float e[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) e[i] = r[cidx[i]];
        // x <= row max; 0 for c >= C.  Two classes per instruction on the packed fp32 pipe (same bits as
        // the scalar form: every half is an IEEE operation)
#pragma unroll
        for (int i = 0; i + 1 < CPL; i += 2) {
          const yp_f2 v = yp_exp2_le0((yp_f2){e[i], e[i + 1]} - (yp_f2){ma, ma}) * (yp_f2){okf[i], okf[i + 1]};
          e[i] = v.x; e[i + 1] = v.y;
        }
        if (CPL & 1) e[CPL - 1] = yp_exp_le0(e[CPL - 1] - ma) * okf[CPL - 1];
        float s_lo = e[0], s_hi = CPL > 1 ? e[1] : 0.0f;                       // s_j = e_j + e_{j+16} + ..., s_{j+8} = e_{j+8} + e_{j+24} + ...
#pragma unroll
        for (int i = 2; i < CPL; ++i) { if (i & 1) s_hi = s_hi + e[i]; else s_lo = s_lo + e[i]; }
        float em = j > 0 ? e[0] : 0.0f;                        // class 0 is background (detection.py:59)
#pragma unroll
        for (int i = 1; i < CPL; ++i) em = max_nonneg(em, e[i]);   // e >= 0: integer max on the bit patterns
        const float S = grp8_sum(s_lo, s_hi);
        em = grp8_max_nonneg(em);
        // YP-MATH softmax: p_c = fl(e_c * rS) with rS = fl(1/S) - one IEEE reciprocal per anchor, then a
        // multiply per class (Eigen's form of tf.nn.softmax).  Multiplication by a positive
        // constant is monotone under rounding, so max_c fl(e_c*rS) = fl(max_c e_c * rS)  (detection.py:62)
        const float rS = 1.0f / S;
        const float pmax = em * rS;
        const bool surv = act && (pmax > p.conf);                                      // detection.py:73
        const bool lead = surv && j == 0;
        const u64 bl = __ballot(lead);
        nsurv += __popcll(bl);
        if (bl == 0) continue;   // (wave-uniform) nobody in this round passed: nothing to emit
        if (MODE == YP_NMS_CROSS_CLASS) {
          // tf.argmax over the probabilities (detection.py:184): the FIRST class whose rounded
          // probability equals the maximum.  fl(e_c*rS) = fl(em*rS) needs e_c >= em (1 - 2^-23) while the
          // products are normal numbers, so only an e_c within 2^-21 of the largest can tie; almost always
          // that is one class, found without any multiply.  (A subnormal pmax - conf = 0 and a foreground
          // maximum ~87 below the row maximum - has no relative precision: every real class is then a
          // candidate for the exact test; e_c of a real class is >= exp(-87) > 0, the padding slots hold 0.)
          const float ethr = pmax >= 0x1p-100f ? em * (1.0f - 0x1p-21f) : 0x1p-149f;
          int cnt = 0, cmin = 0x7fffffff;
#pragma unroll
          for (int i = 0; i < CPL; ++i) {
            const bool hit = (e[i] >= ethr) && (i > 0 || j > 0);
            cnt += hit ? 1 : 0;
            cmin = hit ? min(cmin, j + 8 * i) : cmin;
          }
          cnt = grp8_sum_i(cnt);
          int arg = grp8_min(cmin);
          if (__ballot(surv && cnt > 1)) {   // rare: several classes within 2^-21 -> exact test
            int a2 = 0x7fffffff;
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
              const bool hit = (e[i] >= ethr) && (i > 0 || j > 0);
              if (hit && (e[i] * rS == pmax)) a2 = min(a2, j + 8 * i);
            }
            a2 = grp8_min(a2);
            if (cnt > 1) arg = a2;
          }
          if (lead) s_key[nbuf + __popcll(bl & lt_mask)] = yp_seg_key(pmax, a0 + al, arg);
          nbuf += __popcll(bl);   // <= 8 per round, <= 32 per tile: flushed between tiles when a tile's worth of room is gone
        } else {
          // per-class candidates: every (anchor, class) with p > conf                 (detection.py:233)
          // fl(e_c*rS) > conf implies e_c > conf S (1 - 2^-23) (two roundings of 2^-24 each), and
          // elo = fl(S fl(conf (1 - 2^-20))) lies below that: no candidate is lost by testing e_c >= elo first
          const float elo = S * p.conf_lo;
          int nhit = 0, csel = 0;
          float esel = 0.0f;
#pragma unroll
          for (int i = CPL - 1; i >= 0; --i) {
            const bool hit = surv && (e[i] >= elo) && (i > 0 || j > 0);
            nhit += hit ? 1 : 0;
            csel = hit ? j + 8 * i : csel;
            esel = hit ? e[i] : esel;
          }
          if (__ballot(nhit > 1) == 0) {
            // common: at most one class per lane, <= 64 keys (the buffer holds <= KCAP - 64 here)
            const float pc = esel * rS;
            const bool has = (nhit == 1) && (pc > p.conf);
            const u64 bm = __ballot(has);
            if (has) s_key[nbuf + __popcll(bm & lt_mask)] = yp_seg_key(pc, a0 + al, csel);
            nbuf += __popcll(bm);
          } else {
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
              bool has = false;
              float pc = 0.0f;
              if (surv && (e[i] >= elo) && (i > 0 || j > 0)) {
                pc = e[i] * rS;
                has = pc > p.conf;
              }
              const u64 bm = __ballot(has);
              if (bm) {
                if (nbuf + __popcll(bm) > YP_SF_KCAP) flush();   // wave-uniform, rare
                if (has) s_key[nbuf + __popcll(bm & lt_mask)] = yp_seg_key(pc, a0 + al, j + 8 * i);
                nbuf += __popcll(bm);
              }
            }
          }
          if (nbuf > YP_SF_KCAP - YP_WAVE) flush();   // room for the next round's <= 64 keys
        }
      }
      // ---- phase 3 --------------------------------------------------------------------------------
      if (MODE == YP_NMS_CROSS_CLASS && nbuf > YP_SF_KCAP - TA) flush();
      // every LDS read of this tile has returned before the next iteration overwrites the buffer
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
      __builtin_amdgcn_wave_barrier();
    }   // tile loop
    if (nbuf > 0) flush();
    if (lane == 0) p.seg_surv[item] = nsurv;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
  }   // item loop
}

template <int MODE, bool FAST>
static const void* kernel_for(int cpl) {
#define YP_SF_CASE(N) case N: return reinterpret_cast<const void*>(&score_filter_kernel<MODE, N, FAST>);
  switch (cpl) {
    YP_SF_CASE(2) YP_SF_CASE(4) YP_SF_CASE(6) YP_SF_CASE(8) YP_SF_CASE(10) YP_SF_CASE(11) YP_SF_CASE(12)
    YP_SF_CASE(14) YP_SF_CASE(16) YP_SF_CASE(20) YP_SF_CASE(24) YP_SF_CASE(32)
  }
#undef YP_SF_CASE
  return nullptr;
}

// classes per lane in phase 2 (8 lanes per anchor), rounded up to an instantiated value
static int classes_per_lane(int C) {
  static const int avail[] = {2, 4, 6, 8, 10, 11, 12, 14, 16, 20, 24, 32};
  const int need = (C + 7) / 8;
  for (int v : avail) if (v >= need) return v;
  return 32;
}

// Bytes of a wave's LDS tile: the tile itself, and at least the NLD KiB that phase 0 writes without
// looking (all NLD staged wave-loads, also the clamped ones past the tile's end).
static int tile_lds_bytes(const yp_handle* h) {
  const int nld = (YP_SF_TA * classes_per_lane(h->cfg.num_classes) + 31) / 32;
  return h->sf_bufb > nld * 1024 ? h->sf_bufb : nld * 1024;
}
static size_t lds_bytes(const yp_handle* h) { return (size_t)tile_lds_bytes(h) + YP_WAVE * sizeof(int2) + YP_SF_KCAP * sizeof(u64); }

static const void* kernel_of(const yp_handle* h, bool fast = true) {
  const int cpl = classes_per_lane(h->cfg.num_classes);
  if (h->cfg.nms_mode == YP_NMS_CROSS_CLASS) return fast ? kernel_for<YP_NMS_CROSS_CLASS, true>(cpl) : kernel_for<YP_NMS_CROSS_CLASS, false>(cpl);
  return fast ? kernel_for<YP_NMS_PER_CLASS, true>(cpl) : kernel_for<YP_NMS_PER_CLASS, false>(cpl);
}

// Single-wave workgroups of this handle's kernel that one CU holds at once: the smaller of the
// runtime's occupancy answer and an LDS census (measured on MI355X: LDS is handed out in units of
// 1280 bytes, 128 units per CU; the occupancy API reports more than that for some sizes, and an
// over-subscribed grid would run in two rounds).
int yp_score_filter_waves_per_cu(const yp_handle* h) {
  const size_t lds = lds_bytes(h);
  int per_cu = (int)(128 / ((lds + 1279) / 1280));
  int api = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, kernel_of(h), YP_WAVE, lds) == hipSuccess && api > 0 && api < per_cu) per_cu = api;
  if (h->tune.sf_waves > 0) per_cu = h->tune.sf_waves;   // experiment knob (yp_tune)
  return per_cu < 1 ? 1 : (per_cu > 32 ? 32 : per_cu);
}

hipError_t yp_launch_score_filter(const yp_handle* h, int batch, const float* pred_cls, const float* pred_offset,
                                  const float* priors, hipStream_t st) {
  ScoreFilterArgs a;
  a.cls = pred_cls;
  a.off = reinterpret_cast<const float4*>(pred_offset); a.priors = reinterpret_cast<const float4*>(priors);
  a.lkeys = h->s.lkeys; a.lbox = h->s.lbox; a.lcount = h->s.lcount; a.okeys = h->s.okeys; a.ocount = h->s.ocount;
  a.seg_surv = h->s.seg_surv; a.overflow = h->s.overflow;
  {
    int gcap;
    yp_plan_groups(h, &a.ngroups, &gcap);
    yp_plan_lists(h, batch, &a.lcap, &a.ocap);
  }
  a.A = h->cfg.num_anchors; a.C = h->cfg.num_classes; a.stride = h->lds_stride;
  a.conf = h->cfg.conf_thresh;
  a.conf_lo = h->cfg.conf_thresh * (1.0f - 0x1p-20f);
  a.pre_thresh = h->pre_thresh;
  a.jen_thr = h->jen_thr; a.jen_n = (float)(a.C - 2); a.jen_inv_n = a.C > 2 ? 1.0f / (float)(a.C - 2) : 0.0f;
  a.ntiles = (a.A + YP_SF_TA - 1) / YP_SF_TA;
  {
    int longest;
    yp_plan_items(h, batch, &a.parts, &longest);
  }
  a.nitems = batch * a.parts;
  a.bufb = tile_lds_bytes(h);
  a.fast = (a.stride == a.C) && (((long long)a.A * a.C) % 4 == 0) && ((((uintptr_t)pred_cls) & 15u) == 0);
  const int resident = h->num_cus * h->sf_waves_per_cu;
  const dim3 grid(a.nitems < resident ? a.nitems : resident);
  void* args[] = {&a};
  return hipLaunchKernel(kernel_of(h, a.fast != 0), grid, dim3(YP_WAVE), args, lds_bytes(h), st);
}
