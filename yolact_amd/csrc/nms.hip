// Kernels 2+3: top-k selection + Fast NMS, and the final detection rows.
//
// Replaces detection.py:186-207 (_cc_fast_nms: top_k, gather, pairwise IoU, strict upper triangle,
// column max, threshold, gathers), detection.py:217-248 (the per-class intent of _fast_nms),
// utils._decode for the candidates that get this far (utils/utils.py:173-213), detection.py:92-98
// (_sanitize + normalise), the edge computation of utils.crop (utils/utils.py:116-117) and the
// box/class/score scatters of detection.py:109-114.
//
// These kernels are latency, not bandwidth: a few thousand 8-byte keys per image.  What they are
// built around is the number of DEPENDENT global round trips (each 1-2 us under load) and the fact
// that one CU moves only ~10-25 bytes per clock, so an image's work is spread over several
// workgroups wherever it is more than a few KB:
//   cross-class  cc_nms_kernel: one workgroup per image (top_k <= 256 boxes reach the NMS)
//   per-class    group_nms_kernel: 16 workgroups per image, each owning the classes c with
//                (c-1) % 16 == g; score_filter has already dealt the candidates - key + decoded box - to
//                one list per group, so a group's workgroup loads its list length and its candidates in
//                ONE round trip, keeps them in registers (up to 4 per thread), buckets the keys by class
//                in LDS, ranks and suppresses thread-per-candidate;
//                the group that finishes last (write-through stores, arrival counter, one invalidate) then
//                does the image's cross-class top max_dets selection and writes its output rows
// Both kernels carry the zero-fill of the mask rows as extra workgroups behind the NMS workgroups
// (fill_block): it depends on nothing the NMS computes, HBM is idle while these latency-bound
// workgroups wait on their dependent loads, and riding in the same launch costs no second stream,
// no events and no 1024-thread workgroup waiting for wave slots behind thousands of fill workgroups.
// The k x k IoU matrix (160 KB at k=200) is never materialised: the reference's reduce_max over the
// strict upper triangle followed by `<= threshold` is "no higher-ranked box overlaps more than the
// threshold", evaluated pair by pair.
#include "common.hpp"

#define YP_RADIX_BITS 11
#define YP_RADIX_BINS (1 << YP_RADIX_BITS)
#define YP_STAGE_CAP 4096    // keys the cross-class kernel stages in LDS for its selection
#define YP_GSTAGE_CAP 2048   // the same for the per-class kernel's final selection (its LDS is kept small: the
                             // zero-fill workgroups of the same launch are allotted the same LDS and should fit 2 per CU
                             // beside 2 NMS workgroups)
#define YP_GROUP_THREADS 512
#define YP_GROUP_KPT 4       // candidates (key + box, in registers) a thread of a class group owns at most
#define YP_GROUP_CAP (YP_GROUP_THREADS * YP_GROUP_KPT)   // candidates of one class group that take the thread-per-candidate path
#define YP_GROUP_SORTED 1024 // ... if the top_k's of its classes add up to at most this many (boxes kept in LDS, sorted per class)

struct NmsArgs {
  const u64* lkeys;           // [B][YP_NLISTS][lcap] score_filter's candidate lists (common.hpp, yp_scratch)
  const float4* lbox;         // [B][YP_NLISTS][lcap] the candidates' decoded boxes
  int32_t* lcount;            // [B][YP_NLISTS] x YP_CNT_STRIDE entries appended per list; reset here
  const u64* okeys;           // [B][ocap] keys that found their list full (per-class mode)
  int32_t* ocount;            // [B] x YP_CNT_STRIDE; reset here
  const int32_t* seg_surv;    // [B*parts] anchors past the confidence filter per score_filter item
  int32_t* stat_surv;         // [B]
  const float4* off;          // [B,A] pred_offset (dx,dy,dw,dh): only the slow paths decode from it
  const float4* priors;       // [A]   (ymin,xmin,ymax,xmax) pixels
  u64* pool;                  // [B,ngroups,gcap] per-class NMS survivors, final-key format, dense per group
  float4* pool_box;           // [B,ngroups,gcap] their boxes
  int32_t* pool_count;        // [B,ngroups]
  float4* crop;               // [B,sel_rows]
  int32_t* sel_anchor;        // [B,sel_rows] anchors of the kept detections, for mask_assemble
  float* det_boxes;           // [B,max_dets,4]
  float* det_classes;         // [B,max_dets]
  float* det_scores;          // [B,max_dets]
  int32_t* num_det;           // [B]
  int32_t* kept_idx;          // [B,max_dets] (nullable)
  int A, nseg, top_k, max_dets, sel_rows, parts, ngroups, gcap;
  long long lcap, ocap;
  // the mask zero-fill that rides in the same launch (workgroups with blockIdx.x >= nms_blocks)
  int32_t* arrive;            // [B] class groups of the image that have finished (per-class mode); zero between launches
  int nms_blocks;             // workgroups that do NMS work: B (cross-class) or B*ngroups (per-class)
  float* fill_masks;          // [B,max_dets,P] or null: rows [0, fill_rows) of every image get zeros
  uint8_t* fill_bitmaps;      // [B,max_dets,P] or null
  long long fill_P;           // proto_h * proto_w
  long long fill_first;       // the launch's fill workgroups take the chunks [fill_first, fill_last) between them
  long long fill_last;        // (the whole fill; a range so that a launch can carry a part of it)
  int fill_wgs;               // number of fill workgroups in the launch (fill workgroup i takes chunks first + i, + fill_wgs, ...)
  int fill_rows, batch;

  float nms_thresh, img_h, img_w, proto_h, proto_w;
  int use_crop;
  // cross-class selection from registers: bin = min((score bits - sel_base) >> sel_shift, YP_RADIX_BINS - 1)
  uint32_t sel_base;          // float bits of the confidence threshold (every candidate's score is above it)
  int sel_shift;              // smallest shift that maps scores up to 1.0 into the bins
#ifdef YP_STAMPS
  unsigned long long* dbg;   // diagnostic build: [kernel 0..2][block][16] cycle stamps of thread 0
#endif
};
#ifdef YP_STAMPS
#define STAMP_INIT(kern, blk) unsigned long long* const dbg_ = p.dbg + ((size_t)(kern) * 4096 + (blk)) * 16; if (threadIdx.x == 0) dbg_[15] = 1
#define STAMP(i) do { if (threadIdx.x == 0) dbg_[i] = (unsigned long long)wall_clock64(); } while (0)   /* 100 MHz, comparable across workgroups */
#else
#define STAMP_INIT(kern, blk)
#define STAMP(i)
#endif

struct SelectShared {
  u64 key[YP_SORT_MAX];    // result: selected keys, sorted descending
  u64 tmp[YP_SORT_MAX];    // selected keys before sorting
  int rank[YP_SORT_MAX];
  int* hist;               // YP_RADIX_BINS ints of LDS, set by the owner (may alias buffers that are idle during a select)
  int wtot[16];
  u64 lo, hi;
  int sel, need, cnt, fill;
};

// ---- key sources ---------------------------------------------------------------------------
// for_each visits every key once, spread over the workgroup's threads.  Loads are issued in
// batches of independent loads so that one memory round trip serves a whole batch.
#define YP_SRC_UNROLL 4
struct ListSrc {   // n contiguous keys (global or LDS)
  const u64* list;
  int n;
  template <int THREADS, class F>
  __device__ __forceinline__ void for_each(F f) const {
    for (int i0 = threadIdx.x; i0 < n; i0 += THREADS * YP_SRC_UNROLL) {
      u64 k[YP_SRC_UNROLL];
#pragma unroll
      for (int u = 0; u < YP_SRC_UNROLL; ++u) {
        const int i = i0 + u * THREADS;
        k[u] = list[min(i, n - 1)];
      }
#pragma unroll
      for (int u = 0; u < YP_SRC_UNROLL; ++u)
        if (i0 + u * THREADS < n) f(k[u]);
    }
  }
};
// The keys of one class group of an image on its slow paths: the group's own list, then those of the image's
// overflow keys (the keys that found their list full) that belong to the group - or, with cls > 0, of both
// only the keys of that class.
struct GroupSrc {
  const u64* list;
  int n;
  const u64* ovf;
  int novf;
  int cls;           // > 0: only the keys of this class
  int gmod, grem;    // the group: classes c with (c - 1) % gmod == grem
  // (c - 1) % gmod without the integer division: for 0 <= x < 256 and gmod <= 256, (x + 0.5) / gmod is at
  // least 0.5 / gmod away from an integer, far more than the float error, so the truncation is x / gmod
  __device__ __forceinline__ bool mine(u64 k, float ginv) const {
    const int c = yp_seg_key_class(k);
    if (cls != 0) return c == cls;
    const int q = (int)(((float)(c - 1) + 0.5f) * ginv);
    return (c - 1) - q * gmod == grem;
  }
  template <int THREADS, class F>
  __device__ __forceinline__ void for_each(F f) const {
    const ListSrc own{list, n};
    if (cls != 0) { const int c = cls; own.template for_each<THREADS>([&](u64 k) { if (yp_seg_key_class(k) == c) f(k); }); }
    else own.template for_each<THREADS>(f);
    if (novf > 0) {
      const float ginv = 1.0f / (float)max(gmod, 1);
      const ListSrc o{ovf, novf};
      o.template for_each<THREADS>([&](u64 k) { if (mine(k, ginv)) f(k); });
    }
  }
};
// Cross-class mode: the image's YP_NLISTS lists, 64 threads (one wave) to a list.  The first YP_CC_PRE entries
// per thread - keys and boxes - are loaded before the list lengths are known (slots past a list's end hold
// stale entries, still inside its allocation), so that lengths, keys and boxes share ONE round trip.
#define YP_CC_PRE 4
#define YP_CC_THREADS 1024
struct CcSrc {
  const u64* keys;        // the image's lists
  const float4* boxes;
  long long lcap;
  const int* cnt;         // LDS: entries per list (already cut to lcap)
  u64 pre[YP_CC_PRE];     // this thread's first YP_CC_PRE keys (a member, not a pointer: it stays in registers)
  template <int THREADS, class F>
  __device__ __forceinline__ void for_each(F f) const {
    static_assert(THREADS == YP_NLISTS * YP_WAVE, "one wave per list");
    const int l = threadIdx.x >> 6, q = threadIdx.x & 63;
    const int n = cnt[l];
    const u64* list = keys + (size_t)l * lcap;
#pragma unroll
    for (int u = 0; u < YP_CC_PRE; ++u)
      if (q + u * YP_WAVE < n) f(pre[u]);
    for (int i0 = q + YP_WAVE * YP_CC_PRE; i0 < n; i0 += YP_WAVE * YP_SRC_UNROLL) {
      u64 k[YP_SRC_UNROLL];
#pragma unroll
      for (int u = 0; u < YP_SRC_UNROLL; ++u) k[u] = list[min(i0 + u * YP_WAVE, n - 1)];
#pragma unroll
      for (int u = 0; u < YP_SRC_UNROLL; ++u)
        if (i0 + u * YP_WAVE < n) f(k[u]);
    }
  }
};
#define YP_POOL_PRE 8   // keys per thread of PoolSrc's first batch, loaded before the list lengths are known
#define YP_POOL_PRE_BOX 4   // ... of which this many come with their boxes (enough for max_dets <= 128 with 16 groups)
struct PoolSrc {   // ngroups lists of gcap slots; list g holds its keys at [0, start[g+1] - start[g])
  const u64* base;
  const int* start;   // LDS: exclusive prefix of the list lengths, ngroups + 1 entries
  int ngroups, gcap;
  u64 pre[YP_POOL_PRE];   // this thread's first YP_POOL_PRE keys (a member, not a pointer: it stays in registers)
  __device__ __forceinline__ int shift() const {
    int sft = 0;      // 1 << sft threads share a list
    while ((2 << sft) * ngroups <= (int)blockDim.x && sft < 6) ++sft;
    return sft;
  }
  template <int THREADS, class F>
  __device__ __forceinline__ void for_each(F f) const {
    const int sft = shift(), lpp = 1 << sft;
    const int s = threadIdx.x;
    if (s < (ngroups << sft)) {
      const int g = s >> sft, q = s & (lpp - 1);
      const int n = start[g + 1] - start[g];
      const u64* l = base + (size_t)g * gcap;
#pragma unroll
      for (int u = 0; u < YP_POOL_PRE; ++u)
        if (q + u * lpp < n) f(pre[u]);
      for (int i0 = q + lpp * YP_POOL_PRE; i0 < n; i0 += lpp * YP_SRC_UNROLL) {
        u64 k[YP_SRC_UNROLL];
#pragma unroll
        for (int u = 0; u < YP_SRC_UNROLL; ++u) k[u] = l[min(i0 + u * lpp, n - 1)];
#pragma unroll
        for (int u = 0; u < YP_SRC_UNROLL; ++u)
          if (i0 + u * lpp < n) f(k[u]);
      }
    }
  }
};

// ---- wave-wide scans and reductions on DPP (one VALU op per step; a __shfl is an LDS round trip) ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_id(int identity, int v) { return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xF, false); }
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16 (row_shr 1,2,4,8),
// then row 0 -> 1 and 2 -> 3 (row_bcast:15), then rows 0-1 -> 2-3 (row_bcast:31)
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += dpp_id<0x111, 0xF>(0, v);
  v += dpp_id<0x112, 0xF>(0, v);
  v += dpp_id<0x114, 0xF>(0, v);
  v += dpp_id<0x118, 0xF>(0, v);
  v += dpp_id<0x142, 0xA>(0, v);
  v += dpp_id<0x143, 0xC>(0, v);
  return v;
}
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_incl_scan(v), 63); }
__device__ __forceinline__ uint32_t wave_or(uint32_t x) {
  int v = (int)x;
  v |= dpp_id<0x111, 0xF>(0, v);
  v |= dpp_id<0x112, 0xF>(0, v);
  v |= dpp_id<0x114, 0xF>(0, v);
  v |= dpp_id<0x118, 0xF>(0, v);
  v |= dpp_id<0x142, 0xA>(0, v);
  v |= dpp_id<0x143, 0xC>(0, v);
  return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_and(uint32_t x) {
  int v = (int)x;
  v &= dpp_id<0x111, 0xF>(-1, v);
  v &= dpp_id<0x112, 0xF>(-1, v);
  v &= dpp_id<0x114, 0xF>(-1, v);
  v &= dpp_id<0x118, 0xF>(-1, v);
  v &= dpp_id<0x142, 0xA>(-1, v);
  v &= dpp_id<0x143, 0xC>(-1, v);
  return (uint32_t)__builtin_amdgcn_readlane(v, 63);
}

// Slot in an LDS list for every calling lane: one LDS atomic per wave instead of one per lane (lanes
// adding to one address are served one after the other).  May be called under divergence: the lanes
// that are active together are counted together.
__device__ __forceinline__ int lds_append_slot(int* counter) {
  const u64 act = __ballot(1);
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)act) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(act));
  base = __builtin_amdgcn_readlane(base, leader);   // leader is wave-uniform
  return base + __popcll(act & ((1ull << lane) - 1ull));
}

template <int THREADS>
__device__ __forceinline__ int block_sum(int v, int* wtot) {   // wtot: >= THREADS/64 ints of LDS
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = v;
  __syncthreads();
  int s = 0;
  for (int w = 0; w < THREADS / 64; ++w) s += wtot[w];
  return s;
}

template <int THREADS> __device__ __forceinline__ void hist_locate(int need, SelectShared& sh);
// One histogram pass over the keys with (key & mask) == prefix, digit = (key >> shift) & dmask.
// Finds the bin that holds the need-th largest of them: sh.sel = bin, sh.need = how many are still
// to be taken inside that bin, sh.cnt = keys in that bin.  Four barriers.
template <int THREADS, class Src>
__device__ __forceinline__ void hist_pass(const Src& src, u64 mask, u64 prefix, int shift, u64 dmask, int need, SelectShared& sh) {
  constexpr int BPT = YP_RADIX_BINS / THREADS;   // bins per thread
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < BPT; ++i) sh.hist[t * BPT + i] = 0;
  __syncthreads();
  src.template for_each<THREADS>([&](u64 key) {
    if ((key & mask) == prefix) atomicAdd(&sh.hist[(int)((key >> shift) & dmask)], 1);
  });
  __syncthreads();
  hist_locate<THREADS>(need, sh);
}
// The bin of the filled histogram sh.hist[] that holds the need-th largest key (bins are monotone in the key):
// sh.sel, sh.need, sh.cnt as above.  Two barriers.
template <int THREADS>
__device__ __forceinline__ void hist_locate(int need, SelectShared& sh) {
  constexpr int BPT = YP_RADIX_BINS / THREADS;   // bins per thread
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int loc[BPT];
  int tsum = 0;
#pragma unroll
  for (int i = 0; i < BPT; ++i) { loc[i] = sh.hist[t * BPT + i]; tsum += loc[i]; }
  const int incl = wave_incl_scan(tsum);                       // sum over lanes <= lane of this wave
  const int wtotal = __builtin_amdgcn_readlane(incl, 63);
  if (lane == 0) sh.wtot[wave] = wtotal;
  __syncthreads();
  int above = wtotal - incl;   // keys in higher bins: the lanes above in this wave, then the waves above
  {
    const int4* w4 = reinterpret_cast<const int4*>(sh.wtot);
#pragma unroll
    for (int q = 0; q < THREADS / 256; ++q) {
      const int4 x = w4[q];
      above += (4 * q + 0 > wave ? x.x : 0) + (4 * q + 1 > wave ? x.y : 0) + (4 * q + 2 > wave ? x.z : 0) + (4 * q + 3 > wave ? x.w : 0);
    }
  }
#pragma unroll
  for (int i = BPT - 1; i >= 0; --i) {
    if (above < need && need <= above + loc[i]) { sh.sel = t * BPT + i; sh.need = need - above; sh.cnt = loc[i]; }
    above += loc[i];
  }
  __syncthreads();
}

// The k-th largest of n unique keys: on return every key with (key & mask) >= prefix is one of the
// k largest, and exactly k keys satisfy it.  MSB-first, YP_RADIX_BITS bits per pass.
template <int THREADS, class Src>
__device__ __forceinline__ void radix_select(const Src& src, int k, SelectShared& sh, u64& prefix_out, u64& mask_out) {
  u64 prefix = 0, mask = 0;
  int need = k;
  for (int hi = 64; hi > 0; hi -= YP_RADIX_BITS) {
    const int shift = hi >= YP_RADIX_BITS ? hi - YP_RADIX_BITS : 0;
    const int bits = hi - shift;
    const u64 dmask = (1ull << bits) - 1ull;
    hist_pass<THREADS>(src, mask, prefix, shift, dmask, need, sh);
    prefix |= (u64)sh.sel << shift;
    mask |= dmask << shift;
    need = sh.need;
    const bool done = sh.cnt == need;   // every key left in the selected bin is taken (uniform decision)
    __syncthreads();
    if (done) break;
  }
  prefix_out = prefix;
  mask_out = mask;
}

// Sorts sh.tmp[0..m) (unique keys, m <= 256) descending into sh.key[0..m): the rank of a key is the
// number of larger keys, counted with broadcast LDS reads - two barriers instead of a sorting network.
// sh.rank[] must be zero on entry; it is left zero.
template <int THREADS>
__device__ __forceinline__ void rank_sort_desc(SelectShared& sh, int m) {
  constexpr int G = THREADS / YP_SORT_MAX;
  const int t = threadIdx.x;
  const int j = t % YP_SORT_MAX, g = t / YP_SORT_MAX;
  u64 mine = 0;
  if (j < m) {
    mine = sh.tmp[j];
    int c = 0;
#pragma unroll 8
    for (int i = g; i < m; i += G) c += sh.tmp[i] > mine ? 1 : 0;
    if (G == 1) sh.rank[j] = c;
    else if (c) atomicAdd(&sh.rank[j], c);
  }
  __syncthreads();
  if (t < m) {
    sh.key[sh.rank[t]] = mine;
    sh.rank[t] = 0;
  }
  __syncthreads();
}

// Leaves the k = min(want, n) largest keys of src in sh.key[0..k), descending; returns k.
// All threads of the workgroup must call it.  Keys are unique and non-zero.  `stage` is an optional
// LDS buffer of stage_cap keys.
//   n <= 256: everything is rank-sorted.
//   else: ONE histogram pass over the 11 bits just below the highest bit in which the keys differ
//   (found from their min and max) isolates the bin of the k-th key; if the keys above that bin plus
//   the bin itself are <= 256 (the normal case: the bins are narrow) they are rank-sorted and the
//   first k are the answer.  Otherwise a full MSB-first radix select runs.
template <int THREADS, class Src>
__device__ __forceinline__ int select_topk_sorted(const Src& src, int n, int want, SelectShared& sh, u64* stage, int stage_cap,
                                                  unsigned long long* dbg_ = nullptr) {
#ifdef YP_STAMPS
#define SSTAMP(i) do { if (dbg_ && threadIdx.x == 0) dbg_[i] = (unsigned long long)wall_clock64(); } while (0)
#else
#define SSTAMP(i)
#endif
  const int t = threadIdx.x;
  const int k = min(want, n);
  if (t == 0) { sh.fill = 0; sh.lo = ~0ull; sh.hi = 0ull; }
  if (t < YP_SORT_MAX) sh.rank[t] = 0;
  __syncthreads();
  int m;
  if (n <= YP_SORT_MAX) {
    src.template for_each<THREADS>([&](u64 key) {
      const int q = lds_append_slot(&sh.fill);
      if (q < YP_SORT_MAX) sh.tmp[q] = key;
    });
    m = n;
  } else {
    const bool staged = stage != nullptr && n <= stage_cap;
    u64 lo = ~0ull, hi = 0ull;   // AND / OR of the keys: they differ exactly in the bits of lo ^ hi
    src.template for_each<THREADS>([&](u64 key) {
      lo &= key;
      hi |= key;
      if (staged) {
        const int q = lds_append_slot(&sh.fill);
        if (q < stage_cap) stage[q] = key;
      }
    });
    SSTAMP(8);
    lo = ((u64)wave_and((uint32_t)(lo >> 32)) << 32) | wave_and((uint32_t)lo);
    hi = ((u64)wave_or((uint32_t)(hi >> 32)) << 32) | wave_or((uint32_t)hi);
    if ((t & 63) == 0) { atomicAnd(&sh.lo, lo); atomicOr(&sh.hi, hi); }
    __syncthreads();
    if (t == 0) sh.fill = 0;
    const u64 diff = sh.lo ^ sh.hi;                         // != 0: more than one distinct key
    const int top = 63 - __clzll((long long)diff);          // highest bit in which keys differ
    const int shift = top >= YP_RADIX_BITS ? top - (YP_RADIX_BITS - 1) : 0;
    const u64 dmask = (1ull << (top - shift + 1)) - 1ull;
    const ListSrc ls{stage, n};
    if (staged) hist_pass<THREADS>(ls, 0ull, 0ull, shift, dmask, k, sh);   // bits above `top` are common
    else hist_pass<THREADS>(src, 0ull, 0ull, shift, dmask, k, sh);
    SSTAMP(9);
    const int sel = sh.sel;
    const int superset = (k - sh.need) + sh.cnt;            // keys in bins >= sel
    if (superset <= YP_SORT_MAX) {
      auto take = [&](u64 key) {
        if ((int)((key >> shift) & dmask) >= sel) {
          const int q = lds_append_slot(&sh.fill);
          if (q < YP_SORT_MAX) sh.tmp[q] = key;
        }
      };
      if (staged) ls.template for_each<THREADS>(take); else src.template for_each<THREADS>(take);
      m = superset;
    } else {
      __syncthreads();
      u64 prefix, mask;
      if (staged) radix_select<THREADS>(ls, k, sh, prefix, mask); else radix_select<THREADS>(src, k, sh, prefix, mask);
      auto take = [&](u64 key) {
        if ((key & mask) >= prefix) {
          const int q = lds_append_slot(&sh.fill);
          if (q < YP_SORT_MAX) sh.tmp[q] = key;
        }
      };
      if (staged) ls.template for_each<THREADS>(take); else src.template for_each<THREADS>(take);
      m = k;
    }
  }
  __syncthreads();
  SSTAMP(10);
  rank_sort_desc<THREADS>(sh, m);
  SSTAMP(11);
  return k;
}

// Selection from REGISTERS with fixed score bins - the fast form of select_topk_sorted for a workgroup whose threads
// already hold every key (at most NK each).  The general selection first finds the bit range in which the keys differ
// (a pass and a barrier), stages or re-reads the keys for its histogram, takes the k-th bin and above, rank-sorts and
// leaves it to the caller to find where its own keys went (a binary search per key).  Here the bins are fixed in
// advance: the score bits above the confidence threshold, cut to YP_RADIX_BITS bits (NmsArgs::sel_base / sel_shift) -
// monotone in the key, which is all a selection needs.  The keys go from the registers straight into the histogram;
// the keys of the k-th bin and above are appended to sh.tmp, rank-sorted into sh.key, and every owner is told the
// rank of each of its keys, so that it can drop the key's payload (box, output row) at its place from its own
// registers.  Six barriers.
//   in:  sh.hist[0..YP_RADIX_BINS) == 0, sh.rank[0..YP_SORT_MAX) == 0, sh.fill == 0, published by a barrier
//   out: the return value m >= k: sh.key[0..m) holds the m largest keys, descending (the first k are the answer);
//        rank[u] = position of key[u] in it, or -1.  sh.rank is zero again.
//        -1 (workgroup-uniform) if the bins do not isolate <= YP_SORT_MAX keys around the k-th (a score plateau):
//        nothing has been selected and the caller takes the general path.
// All threads must call it; k >= 1 and at most the number of valid keys.
// (In the per-class kernel - the cut of a group's pool to its best max_dets and the image's final selection, both
//  of which hold their keys in registers - only the instantiation launched WITHOUT fill workgroups uses it.  Measured
//  in the other one: the cut of a group's pool to its best max_dets and the image's final
//  selection, both of which hold their keys in registers: 30.0 -> 28.4 us for the chain on tracked output sets (no
//  fill beside it), nothing measurable under the fill, and +2-4 us at config 4, whose pools are a thousand EQUAL
//  scores just below 1.0 ordered by anchor alone - every key in one bin, a thousand LDS atomics on one address - and
//  whose group kernel, already spilling at its 80-register bound, spilled more with the extra code even when the
//  attempt was skipped.  Two-scale bins (linear above 0.5) and a second histogram level did not change that.  Left
//  out there.)
template <int THREADS, int NK>
__device__ __forceinline__ int select_from_registers(const u64 (&key)[NK], const bool (&valid)[NK], int k, uint32_t sel_base,
                                                     int sel_shift, SelectShared& sh, int (&rank)[NK]) {
  const int t = threadIdx.x;
  int bin[NK];
#pragma unroll
  for (int u = 0; u < NK; ++u) {
    const uint32_t sb = (uint32_t)(key[u] >> 32);
    bin[u] = valid[u] ? (int)min((sb > sel_base ? sb - sel_base : 0u) >> sel_shift, (uint32_t)(YP_RADIX_BINS - 1)) : -1;
    if (bin[u] >= 0) atomicAdd(&sh.hist[bin[u]], 1);
    rank[u] = -1;
  }
  __syncthreads();
  hist_locate<THREADS>(k, sh);
  const int sel = sh.sel;
  const int m = (k - sh.need) + sh.cnt;                       // keys in bins >= sel
  if (m > YP_SORT_MAX) return -1;
  int slot[NK];
#pragma unroll
  for (int u = 0; u < NK; ++u) {
    slot[u] = -1;
    if (bin[u] >= sel) {
      slot[u] = lds_append_slot(&sh.fill);
      sh.tmp[slot[u]] = key[u];
    }
  }
  __syncthreads();
  constexpr int G = THREADS / YP_SORT_MAX >= 1 ? THREADS / YP_SORT_MAX : 1;
  const int j = t % YP_SORT_MAX, g = t / YP_SORT_MAX;
  u64 mine = 0;
  if (j < m && g < G) {
    mine = sh.tmp[j];
    int c = 0;
#pragma unroll 8
    for (int i = g; i < m; i += G) c += sh.tmp[i] > mine ? 1 : 0;
    if (c) atomicAdd(&sh.rank[j], c);
  }
  __syncthreads();
  if (t < m) {
    const int r = sh.rank[t];
    sh.key[r] = mine;
    sh.tmp[t] = (u64)r;                                       // slot -> rank, for the key's owner
    sh.rank[t] = 0;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NK; ++u)
    if (slot[u] >= 0) rank[u] = (int)sh.tmp[slot[u]];
  return m;
}

// iou(a, b) > thr with utils._iou's arithmetic (utils/utils.py:19-21, 40-48, 63-70), without the
// division wherever the answer cannot depend on its rounding: thr*union is within 2^-24 of exact, so
// an intersection outside thr*union*(1 -+ 2^-21) is on one side of the threshold for any correctly
// rounded quotient (the neighbours of thr are >= 2^-24 relative away).  In between - and for a
// non-positive union - the reference's expression is evaluated as written.
// (v_min_f32 / v_max_f32 as such: fminf / fmaxf would first pass every operand through a
// canonicalising v_max x,x - eight extra instructions per pair - for signalling NaNs that decoded
// boxes do not contain; on ordinary numbers the results are identical)
__device__ __forceinline__ float vmin(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float vmax(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ bool yp_iou_gt(const float4 a, const float4 b, float thr) {
  const float area1 = (a.z - a.x) * (a.w - a.y);
  const float area2 = (b.z - b.x) * (b.w - b.y);
  const float ih = vmax(0.0f, vmin(a.z, b.z) - vmax(a.x, b.x));
  const float iw = vmax(0.0f, vmin(a.w, b.w) - vmax(a.y, b.y));
  const float inter = ih * iw;
  const float uni = (area1 + area2) - inter;
  const float pr = thr * uni;
  if (uni > 0.0f) {
    if (inter > pr * (1.0f + 0x1p-21f)) return true;
    if (inter < pr * (1.0f - 0x1p-21f)) return false;
  }
  return (inter == 0.0f ? 0.0f : inter / uni) > thr;
}

// Fast NMS pair sweep: sup[j] = 1 for every j < k that some i < j overlaps by more than thr
// (detection.py:190-202; box i outranks box j).  Column j has j rows; columns f and k-1-f are folded
// into one of k-1 rows, and the rows of a folded column are cut into G parts, so that ~nf*G threads
// have equal work.  A thread keeps its two column boxes in registers and reads ONE row box per test;
// the lanes of a wave share the part, so in the first column's rows they read the same box (an LDS
// broadcast) and in the second's consecutive ones.  ~25 VALU operations per pair: the sweep is what
// bounds a saturated image (80 classes x 200 boxes: 1.6 M pairs).
// (tid, nthr): the sweep is done by the nthr threads tid = 0..nthr-1 - the whole workgroup, or the share of it that
// one class gets when the classes of a group are swept side by side.
template <class T>
__device__ __forceinline__ void pair_sweep(const float4* box, T* sup, int k, float thr, int tid, int nthr) {
  if (k < 2) return;
  const int nf = (k + 1) >> 1;
  const int G = max(1, nthr / nf);
  const int L = (k - 1 + G - 1) / G;
  for (int s = tid; s < nf * G; s += nthr) {   // (one trip unless nf > nthr)
    const int part = s / nf, f = s - part * nf;
    const int jb = k - 1 - f;
    const int rows = f + (jb != f ? jb : 0);               // (odd k: the middle column is folded with nothing)
    const int v0 = part * L, v1 = min(v0 + L, rows);
    if (v0 >= v1) continue;
    const float4 ba = box[f], bb = box[jb];
    bool ha = false, hb = false;
    for (int v = v0; v < v1; ++v) {
      const bool first = v < f;
      const float4 bi = box[first ? v : v - f];
      const float4 bj = first ? ba : bb;
      const bool hit = yp_iou_gt(bi, bj, thr);
      ha |= hit && first;
      hb |= hit && !first;
    }
    if (ha) sup[f] = 1;
    if (hb) sup[jb] = 1;
  }
}

// ---- the sweep of a LARGE class: cell masks first, the IoU arithmetic only where two boxes can meet ---------------
// Pair tests are what a saturated image costs (config 4: 80 classes x 19 900 pairs x 32 images = 51 M per batch at
// ~45 vector instructions each: the class groups were vector-ALU bound, 59 us of their 110).  Almost all of those
// pairs are boxes that do not intersect at all, and a pair with inter == 0 can never exceed the threshold
// (utils/utils.py:68-70, thr > 0).  So every box gets a 64-bit mask of the cells of an 8 x 8 grid over the image
// that it touches; two boxes that intersect share a cell (the cell index is a monotone function of the
// coordinate, and max(xmin_a, xmin_b) lies in both boxes' ranges), so `mask_i & mask_j == 0` proves inter == 0
// exactly.  A wave takes four columns at a time, its lanes 64 rows: one LDS read and ~5 vector instructions per
// 64 pairs and column.  The pairs that pass (a few per cent) are queued per wave in LDS and get the full test
// (yp_iou_gt, the reference's arithmetic) 64 at a time, one pair per lane.
__device__ __forceinline__ u64 yp_cell_mask(const float4 bx /* ymin,xmin,ymax,xmax */, float sx, float sy) {
  // cell = clamp(trunc(coordinate * 8 / size), 0, 7): monotone; NaN goes to 0
  const int cx0 = (int)fminf(fmaxf(bx.y * sx, 0.0f), 7.0f), cx1 = (int)fminf(fmaxf(bx.w * sx, 0.0f), 7.0f);
  const int cy0 = (int)fminf(fmaxf(bx.x * sy, 0.0f), 7.0f), cy1 = (int)fminf(fmaxf(bx.z * sy, 0.0f), 7.0f);
  const uint32_t cols = ((2u << max(cx0, cx1)) - (1u << min(cx0, cx1))) & 0xffu;   // bits cx0..cx1
  const uint32_t rows = ((2u << max(cy0, cy1)) - (1u << min(cy0, cy1))) & 0xffu;   // bits cy0..cy1
  // every row bit becomes a byte holding `cols` (cols < 256: no carries between the bytes)
  const uint32_t rl = rows & 0xfu, rh = rows >> 4;
  const uint32_t lo = cols * ((rl & 1u) | ((rl & 2u) << 7) | ((rl & 4u) << 14) | ((rl & 8u) << 21));
  const uint32_t hi = cols * ((rh & 1u) | ((rh & 2u) << 7) | ((rh & 4u) << 14) | ((rh & 8u) << 21));
  return ((u64)hi << 32) | lo;
}

#define YP_SWEEP_COLS 4     // columns a wave tests per pass over the rows
#define YP_SWEEP_QUEUE 128  // pair-queue entries per wave (a drain leaves < 64, a column adds <= 64)
template <class T>
__device__ __forceinline__ void pair_sweep_cells(const float4* box, const u64* cmask, T* sup, uint32_t* queue /* this wave's */,
                                                 int k, float thr, int wave, int nw, int lane) {
  int qlen = 0;                                                 // wave-uniform
  const u64 lt = (1ull << lane) - 1ull;
  auto drain = [&](int n) {                                     // the first n (<= 64) queued pairs get the full test
    __builtin_amdgcn_wave_barrier();
    const uint32_t pr = queue[min(lane, n - 1)];
    const int i = (int)(pr & 0xffffu), j = (int)(pr >> 16);
    const bool hit = yp_iou_gt(box[i], box[j], thr);           // box i outranks box j (detection.py:190-202)
    if (lane < n && hit) sup[j] = 1;
    const int rest = qlen - n;                                  // < 64
    const uint32_t mv = queue[min(n + lane, YP_SWEEP_QUEUE - 1)];
    __builtin_amdgcn_wave_barrier();
    if (lane < rest) queue[lane] = mv;
    qlen = rest;
  };
  const int w = __builtin_amdgcn_readfirstlane(wave);
  for (int j0 = 1 + w * YP_SWEEP_COLS; j0 < k; j0 += nw * YP_SWEEP_COLS) {     // (column 0 has no rows)
    u64 mj[YP_SWEEP_COLS];
#pragma unroll
    for (int c = 0; c < YP_SWEEP_COLS; ++c) mj[c] = cmask[min(j0 + c, k - 1)];   // broadcasts
    const int jl = min(j0 + YP_SWEEP_COLS - 1, k - 1);         // the pass's last column: rows 0 .. jl-1
    for (int i0 = 0; i0 < jl; i0 += YP_WAVE) {
      const int i = i0 + lane;
      const u64 mi = cmask[min(i, jl - 1)];
#pragma unroll
      for (int c = 0; c < YP_SWEEP_COLS; ++c) {
        const int j = j0 + c;
        const bool pass = (i < j) && (j < k) && ((mi & mj[c]) != 0ull);
        const u64 bm = __ballot(pass);
        if (bm != 0ull) {                                       // (wave-uniform)
          if (pass) queue[qlen + __popcll(bm & lt)] = (uint32_t)i | ((uint32_t)j << 16);
          qlen += __popcll(bm);
          if (qlen >= YP_WAVE) drain(YP_WAVE);
        }
      }
    }
  }
  if (qlen > 0) drain(qlen);
}

// Output row r of image b: sanitize + normalise (detection.py:92-98), class/score/index
// (L110-112), the row's crop edges (utils/utils.py:116-117 with padding 1.0, normalised
// coordinates: SURVEY.md A12) and its anchor, from which mask_assemble finds the coefficient row.
// `bx` is the decoded box as score_filter stored it beside the key.
__device__ __forceinline__ void emit_row(const NmsArgs& p, int b, int r, float score, int cls, int anchor,
                                         const float4 bx /* ymin,xmin,ymax,xmax pixels */) {
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
  p.sel_anchor[(size_t)b * p.sel_rows + r] = anchor;         // mask_assemble reads the coefficient row itself (L206)
}

// Position of `key` in k unique keys sorted descending (it is one of them).
__device__ __forceinline__ int rank_desc(const u64* sorted, int k, u64 key) {
  int lo = 0, hi = k - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sorted[mid] > key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Rows [n, max_dets) of image b: zeros / -1 (detection.py:65-67); num_det (L114).
template <int THREADS>
__device__ __forceinline__ void finish_image(const NmsArgs& p, int b, int n, int survivors) {
  const int t = threadIdx.x;
  for (int r = n + t; r < p.max_dets; r += THREADS) {
    const size_t o = (size_t)b * p.max_dets + r;
    reinterpret_cast<float4*>(p.det_boxes)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    p.det_classes[o] = 0.0f;
    p.det_scores[o] = 0.0f;
    if (p.kept_idx) p.kept_idx[o] = -1;
  }
  if (t == 0) {
    p.num_det[b] = n;
    p.stat_surv[b] = survivors;
  }
}

// keep[j] = no box i < j of the k sorted boxes in sbox has iou(i,j) > thr, i.e.
// max_{i<j} iou(i,j) <= thr (detection.py:190-202), order preserved.  Returns this thread's keep
// flag (threads < k) and its rank among the kept ones in `pos`; `total` = number kept.  ssup[] must
// be zero.  All threads must call it.
template <int THREADS>
__device__ __forceinline__ bool fast_nms_keep(const float4* sbox, int* ssup, int* s_wave_cnt, int k, float thr, int& pos, int& total) {
  const int t = threadIdx.x;
  pair_sweep(sbox, ssup, k, thr, t, THREADS);
  __syncthreads();
  bool keep = false;
  if (t < k) keep = ssup[t] == 0;
  pos = 0;
  if (t < YP_SORT_MAX) {
    const u64 ball = __ballot(keep);
    pos = __popcll(ball & ((1ull << (t & 63)) - 1ull));
    if ((t & 63) == 0) s_wave_cnt[t >> 6] = __popcll(ball);
  }
  __syncthreads();
  total = 0;
#pragma unroll
  for (int w = 0; w < YP_SORT_MAX / YP_WAVE; ++w) {
    if (w < (t >> 6)) pos += s_wave_cnt[w];
    total += s_wave_cnt[w];
  }
  return keep;
}

// ---- zero-fill of the mask rows (extra workgroups of the NMS launches) -----------------------------
// Rows [0, fill_rows) of an image are contiguous in [B,max_dets,P]: one linear stream per image, the
// form in which HBM takes writes fastest.  Chunk `ch` = THREADS * YP_FILL_UNROLL * 16 bytes of the float
// masks of one image (YP_FILL_UNROLL 16-byte stores per thread, issued back to back: the kernel's LDS
// footprint allows only 2-3 workgroups per CU, so a workgroup must keep many stores in flight);
// the chunks of the uint8 bitmaps, if any, follow.
#ifndef YP_FILL_UNROLL
#define YP_FILL_UNROLL 16
#endif
template <int THREADS>
__device__ __forceinline__ void fill_region(char* out, long long len, long long img_stride, long long ch, long long cpi) {
  constexpr long long CH = (long long)THREADS * YP_FILL_UNROLL * 16;
  const long long b = ch / cpi;
  char* const base = out + b * img_stride;
  if ((((uintptr_t)base) & 15u) == 0) {
    // Chunk windows are cut at multiples of 128 bytes of the ADDRESS, not of the image's first row (an
    // image's rows start 16-byte aligned only: 4*P*max_dets is rarely a multiple of 128), so that every
    // wave-instruction writes eight whole cache lines instead of seven and two halves.
    typedef float nat4 __attribute__((ext_vector_type(4)));
    const nat4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    const long long shift = (long long)(((uintptr_t)base) & 127u);
    const long long lo = (ch - b * cpi) * CH - shift;
    const long long len16 = len & ~15ll;
#pragma unroll
    for (int u = 0; u < YP_FILL_UNROLL; ++u) {
      const long long o = lo + ((long long)u * THREADS + threadIdx.x) * 16;
      if (o >= 0 && o < len16) *reinterpret_cast<nat4*>(base + o) = z;
    }
    if (lo + CH >= len16 && lo < len)            // the chunk that holds the region's last bytes
      for (long long o = len16 + threadIdx.x; o < len; o += THREADS) base[o] = 0;
  } else {
    const long long lo = (ch - b * cpi) * CH;
    const long long hi = min(lo + CH, len);
    for (long long o = lo + threadIdx.x; o < hi; o += THREADS) base[o] = 0;
  }
}
template <int THREADS>
__host__ __device__ inline long long fill_chunks_per_image(long long P, int rows, int esize) {   // (+128: the windows' alignment shift)
  return ((long long)rows * P * esize + 128 + (long long)THREADS * YP_FILL_UNROLL * 16 - 1) / ((long long)THREADS * YP_FILL_UNROLL * 16);
}
template <int THREADS>
__device__ __forceinline__ void fill_block(const NmsArgs& p, long long ch) {
#ifdef YP_STAMPS
  if (threadIdx.x == 0) atomicMin(p.dbg + (size_t)2 * 4096 * 16 + 0, (unsigned long long)wall_clock64());
  struct AtEnd { const NmsArgs& p; __device__ ~AtEnd() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (threadIdx.x == 0) atomicMax(p.dbg + (size_t)2 * 4096 * 16 + 1, (unsigned long long)wall_clock64()); } } at_end{p};
#endif
  const long long cpi4 = fill_chunks_per_image<THREADS>(p.fill_P, p.fill_rows, 4);
  const long long n4 = p.fill_masks ? cpi4 * p.batch : 0;
  if (ch < n4) {
    fill_region<THREADS>(reinterpret_cast<char*>(p.fill_masks), (long long)p.fill_rows * p.fill_P * 4,
                         (long long)p.max_dets * p.fill_P * 4, ch, cpi4);
  } else if (p.fill_bitmaps) {
    const long long cpi1 = fill_chunks_per_image<THREADS>(p.fill_P, p.fill_rows, 1);
    if (ch - n4 < cpi1 * p.batch)
      fill_region<THREADS>(reinterpret_cast<char*>(p.fill_bitmaps), (long long)p.fill_rows * p.fill_P,
                           (long long)p.max_dets * p.fill_P, ch - n4, cpi1);
  }
}

// A fill workgroup of an NMS launch takes every fill_wgs-th chunk of the launch's range.  (Waiting for a chunk's
// acknowledgements before the next chunk - to bound the zeros in flight - slows the fill down and does not make the
// loads of the NMS workgroups beside it any faster: the stores are acknowledged by the L2, and what a load waits
// for under the fill is the memory side; 32 KiB chunks with waits: 180 us per launch against 56.)
template <int THREADS>
__device__ __forceinline__ void fill_loop(const NmsArgs& p, int wg) {
  for (long long ch = p.fill_first + wg; ch < p.fill_last; ch += p.fill_wgs) fill_block<THREADS>(p, ch);
}

// Cross-class mode: one workgroup per image (detection.py:181-209).
// One round trip brings the lengths of the image's lists and the first YP_CC_PRE entries per thread (keys and
// boxes, 4096 per image); the top_k selection runs on the keys; then every thread that holds one of the selected
// keys drops its box at the key's rank.  No gather from pred_offset, no coefficient copy.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void cc_nms_kernel(const NmsArgs p) {
  __shared__ SelectShared sh;
  __shared__ int s_hist[YP_RADIX_BINS];
  __shared__ u64 stage[YP_STAGE_CAP];
  __shared__ float4 sbox[YP_SORT_MAX];
  __shared__ int ssup[YP_SORT_MAX];
  __shared__ int s_wave_cnt[YP_SORT_MAX / YP_WAVE];
  __shared__ int s_cnt[YP_NLISTS];
  const int t = threadIdx.x;
  if ((int)blockIdx.x >= p.nms_blocks) {   // workgroup-uniform
    fill_loop<THREADS>(p, (int)blockIdx.x - p.nms_blocks);
    return;
  }
  if (t == 0) sh.hist = s_hist;            // (published by the barriers of block_sum below)
  const int b = blockIdx.x;
  STAMP_INIT(0, b);
  STAMP(0);
  const int l = t >> 6, q = t & 63;        // wave l reads list l
  const u64* const lk = p.lkeys + ((size_t)b * YP_NLISTS) * p.lcap;
  const float4* const lb = p.lbox + ((size_t)b * YP_NLISTS) * p.lcap;
  int32_t* const cnt_g = p.lcount + ((size_t)b * YP_NLISTS) * YP_CNT_STRIDE;
  int my_cnt = 0;
  if (q == 0) my_cnt = cnt_g[YP_CNT_IDX(l)];
  CcSrc src{lk, lb, p.lcap, s_cnt, {}};
  float4 preb[YP_CC_PRE];
#pragma unroll
  for (int u = 0; u < YP_CC_PRE; ++u) {
    const size_t o = (size_t)l * p.lcap + min((long long)(q + u * YP_WAVE), p.lcap - 1);
    src.pre[u] = lk[o];
    preb[u] = lb[o];
  }
  // (while the loads are under way) what the selection from registers expects to be zero
#pragma unroll
  for (int i = 0; i < YP_RADIX_BINS / THREADS; ++i) s_hist[t * (YP_RADIX_BINS / THREADS) + i] = 0;
  if (t < YP_SORT_MAX) { sh.rank[t] = 0; ssup[t] = 0; }
  if (t == 0) sh.fill = 0;
  int part = 0;
  for (int i = t; i < p.parts; i += THREADS) part += p.seg_surv[(size_t)b * p.parts + i];
  if (q == 0) {
    my_cnt = (int)min((long long)my_cnt, p.lcap);   // (an anchor emits one key: the lists cannot overflow)
    s_cnt[l] = my_cnt;
    if (my_cnt != 0) cnt_g[YP_CNT_IDX(l)] = 0;  // ready for the next launch: this workgroup is the counter's only reader
  }
  const int survivors = block_sum<THREADS>(part, sh.wtot);    // (its barriers also publish s_cnt[])
  int n = 0;
#pragma unroll
  for (int i = 0; i < YP_NLISTS; ++i) n += s_cnt[i];          // candidates of the image
  if (n == 0) {                                                // detection.py:81
    finish_image<THREADS>(p, b, 0, survivors);
    return;
  }
  STAMP(1);
  const int k = min(p.top_k, n);
  // ---- selection from registers (the usual case: no list is longer than the first batch) ----------------------
  // Small batches are latency: the general selection below costs ten barriers (min/max pass, histogram, take,
  // rank sort, box placement); select_from_registers six.  The result is the same k keys in the same order
  // (unique keys); if the fixed bins do not isolate <= 256 keys the general path runs instead.
  bool in_regs = true;
#pragma unroll
  for (int i = 0; i < YP_NLISTS; ++i) in_regs = in_regs && s_cnt[i] <= YP_CC_PRE * YP_WAVE;
  int m = -1;
  if (in_regs) {                                                // (workgroup-uniform)
    const int nl = s_cnt[l];
    bool valid[YP_CC_PRE];
    int rk[YP_CC_PRE];
#pragma unroll
    for (int u = 0; u < YP_CC_PRE; ++u) valid[u] = q + u * YP_WAVE < nl;
    m = select_from_registers<THREADS, YP_CC_PRE>(src.pre, valid, k, p.sel_base, p.sel_shift, sh, rk);
    if (m >= 0) {
#pragma unroll
      for (int u = 0; u < YP_CC_PRE; ++u)
        if (rk[u] >= 0 && rk[u] < k) sbox[rk[u]] = preb[u];    // detection.py:187: the selected keys' boxes, by their owners
    }
    __syncthreads();
  }
  STAMP(2);
  if (m < 0) {
#ifdef YP_STAMPS
  select_topk_sorted<THREADS>(src, n, p.top_k, sh, stage, YP_STAGE_CAP, dbg_);
#else
  select_topk_sorted<THREADS>(src, n, p.top_k, sh, stage, YP_STAGE_CAP);       // detection.py:186
#endif
  if (t < YP_SORT_MAX) ssup[t] = 0;
    // the selected keys' boxes (detection.py:187, :79): whoever holds a key >= the k-th largest puts its box at
    // the key's rank
    const u64 kth = sh.key[k - 1];
    const int nl = s_cnt[l];
#pragma unroll
    for (int u = 0; u < YP_CC_PRE; ++u)
      if (q + u * YP_WAVE < nl && src.pre[u] >= kth) sbox[rank_desc(sh.key, k, src.pre[u])] = preb[u];
    for (int i = q + YP_WAVE * YP_CC_PRE; i < nl; i += YP_WAVE) {   // (lists longer than the first batch: rare)
      const u64 key = lk[(size_t)l * p.lcap + i];
      if (key >= kth) sbox[rank_desc(sh.key, k, key)] = lb[(size_t)l * p.lcap + i];
    }
  __syncthreads();
  }
  STAMP(3);
  int pos, total;
  const bool keep = fast_nms_keep<THREADS>(sbox, ssup, s_wave_cnt, k, p.nms_thresh, pos, total);
  STAMP(4);
  const int nk = min(total, p.max_dets);
  if (keep && pos < nk) {
    const u64 key = sh.key[t];
    emit_row(p, b, pos, yp_key_score(key), yp_seg_key_class(key), min(yp_seg_key_anchor(key), p.A - 1), sbox[t]);   // detection.py:204-207
  }
  STAMP(5);
  finish_image<THREADS>(p, b, nk, survivors);
  STAMP(6);
}

// ---- per-class mode -------------------------------------------------------------------------
// Stores of data that ANOTHER workgroup of the same launch reads (the groups' pooled survivors and
// their counts, read by the image's last-arriving group): write-through (global_store ... sc1), so
// that the hand-off needs no agent-scope release - a release writes back every dirty line of the
// XCD's L2, and with the mask zero-fill running in the same launch that is megabytes per fence
// (measured: 208 us per launch with release fences against ~50 with write-through stores).
// tests/test_abi.py disassembles the built library and checks that these stores carry sc1 and that the
// arrival is an atomic followed by an invalidate: the C++ memory model promises nothing for this pattern.
// (Tried in round 3: the class groups and the images' selections as TWO launches, each with a part of the fill -
// plain stores, no protocol.  57.7 us against 55.4 at config 2 and 100 against 91 at config 4: the selections'
// ~16 us under the fill's load no longer overlap the groups, and a launch boundary behind a fill costs the L2's
// write-back.  A head start of 2.5 us for the selections' loads - they return in 2 us then, against 10-16 once
// the zero stream runs - changed nothing in total.)
__device__ __forceinline__ void store_wt(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_wt(int32_t* p, int32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_wt(float4* p, const float4 v) {
  u64* const q = reinterpret_cast<u64*>(p);
  store_wt(q, ((u64)__builtin_bit_cast(uint32_t, v.y) << 32) | __builtin_bit_cast(uint32_t, v.x));
  store_wt(q + 1, ((u64)__builtin_bit_cast(uint32_t, v.w) << 32) | __builtin_bit_cast(uint32_t, v.z));
}

#define YP_POOL_CROWDED 0x40000000   // flag in a group's pool_count: the pool was cut from a crowd
struct GroupShared {
  // One 24 KiB region with several lives.
  //   thread-per-candidate path:  bk (keys bucketed by class, while they are ranked), then box (per class the
  //                               top_k boxes in rank order, while they are swept), then stage|hist for the cut
  //                               of the pooled survivors to the group's best max_dets;
  //   class-by-class slow path:   box[0..256) = the class's sorted boxes, hist behind them;
  //   the image's final selection (last group): stage | hist.
  union {
    u64 bk[YP_GROUP_CAP];
    struct {
      u64 stage[YP_GSTAGE_CAP];     // a select's staged keys
      int hist[YP_RADIX_BINS];      // radix histogram of a select
    };
    struct {
      float4 box[YP_GROUP_SORTED];
      u64 cmask[YP_GROUP_SORTED];   // the cells of an 8 x 8 grid over the image that a box touches (sweeps of large classes)
    };
  };
  unsigned char psup[YP_GROUP_SORTED];   // 1 = suppressed by a higher-ranked box of its class (sorted positions)
  SelectShared sh;
  int cls_cnt[256], cls_off[256], cls_fill[256], cls_soff[256];
  int ssup[YP_SORT_MAX];
  int s_wave_cnt[YP_SORT_MAX / YP_WAVE];
  int start[32];                // final selection: exclusive prefix of the groups' pool lengths
  int nsorted, npool, last, m_raw, novf_raw;
};
static_assert(YP_GROUP_CAP * sizeof(u64) <= YP_GSTAGE_CAP * sizeof(u64) && YP_GROUP_CAP <= YP_RADIX_BINS, "bk lies before hist, whose words hold the candidates' ranks");
static_assert(YP_GROUP_SORTED * sizeof(float4) == YP_GSTAGE_CAP * sizeof(u64) && YP_GROUP_SORTED * sizeof(u64) <= YP_RADIX_BINS * sizeof(int),
              "hist lies behind box[] (and behind the slow path's box[0..256)); cmask[] lies over hist");
static_assert((YP_GROUP_THREADS / YP_WAVE) * YP_SWEEP_QUEUE * sizeof(uint32_t) <= 2 * YP_SORT_MAX * sizeof(u64),
              "the waves' pair queues fit SelectShared's key | tmp (idle during the sweeps)");
static_assert(YP_GROUP_SORTED * sizeof(u64) <= YP_GSTAGE_CAP * sizeof(u64), "the pooled keys of the cut fit the stage");

// Top_k + Fast NMS of ONE class with the whole workgroup (any number of candidates): the slow path, for
// groups the thread-per-candidate code cannot take.  Appends the survivors to pool[base...] and
// returns their number.  All threads must call it.
template <int THREADS>
__device__ __forceinline__ int class_nms_blockwide(const NmsArgs& p, int b, int cls, const GroupSrc& src, GroupShared& gs, u64* pool,
                                                   float4* pbox, int base) {
  const int t = threadIdx.x;
  SelectShared& sh = gs.sh;
  float4* sbox = gs.box;
  __syncthreads();
  if (t == 0) sh.fill = 0;
  if (t < YP_SORT_MAX) { gs.ssup[t] = 0; sh.rank[t] = 0; }
  __syncthreads();
  src.template for_each<THREADS>([&](u64 key) {
    const int q = lds_append_slot(&sh.fill);
    if (q < YP_SORT_MAX) sh.tmp[q] = key;
  });
  __syncthreads();
  const int n = sh.fill;
  if (n == 0) return 0;
  int k;
  if (n <= YP_SORT_MAX) {
    rank_sort_desc<THREADS>(sh, n);
    k = min(p.top_k, n);
  } else {
    __syncthreads();   // everybody has read sh.fill
    k = select_topk_sorted<THREADS>(src, n, p.top_k, sh, nullptr, 0);             // detection.py:217
  }
  float4 mybox = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < k) {                                                                          // detection.py:219
    const int a = min(yp_seg_key_anchor(sh.key[t]), p.A - 1);   // (clamp: never index out of bounds)
    mybox = yp_decode(p.priors[a], p.off[(size_t)b * p.A + a]);
    sbox[t] = mybox;
  }
  __syncthreads();
  int pos, total;
  const bool keep = fast_nms_keep<THREADS>(sbox, gs.ssup, gs.s_wave_cnt, k, p.nms_thresh, pos, total);
  if (keep) {                                                                           // detection.py:234-239
    const u64 key = sh.key[t];
    store_wt(&pool[base + pos], yp_fin_key(yp_key_score(key), yp_seg_key_anchor(key), cls));
    store_wt(&pbox[base + pos], mybox);
  }
  return total;
}

// Slow path of a class group: class by class, each with the whole workgroup, streaming the group's list (and the
// image's overflow keys) from memory for every class and decoding the boxes from the inputs.  Taken when the
// group has more candidates than the thread-per-candidate path holds, or when the image has overflow keys.
template <int THREADS>
__device__ __forceinline__ void group_nms_slow(const NmsArgs& p, int b, int g, int m, int novf, GroupShared& gs, u64* pool, float4* pbox) {
  const int t = threadIdx.x;
  const size_t li = (size_t)b * YP_NLISTS + g;
  int base = 0;
  for (int c = g + 1; c <= p.nseg; c += p.ngroups) {
    const GroupSrc csrc{p.lkeys + li * p.lcap, m, p.okeys + (size_t)b * p.ocap, novf, c, p.ngroups, g};
    base += class_nms_blockwide<THREADS>(p, b, c, csrc, gs, pool, pbox, base);
  }
  const int want = min(p.max_dets, YP_SORT_MAX);
  if (base > want) {
    // cut the pooled list to the group's best max_dets (see group_nms_body, step 6): read back from L2, where the
    // write-through stores of all waves are once they have drained
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const ListSrc ls{pool, base};
    const int kept = select_topk_sorted<THREADS>(ls, base, want, gs.sh, gs.stage, YP_GSTAGE_CAP);
    // the boxes of the kept keys, moved to the keys' new places: read all first, then write
    float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
    int dst = -1;
    const u64 kth = gs.sh.key[kept - 1];
    // (base <= nclasses * top_k can exceed the thread count: the old slots are visited in rounds of THREADS.
    //  Every new place is a rank < kept <= 256 <= THREADS, i.e. inside the FIRST round's range, whose reads all
    //  complete before its writes; later rounds read slots that no round writes.)
    for (int i0 = 0; i0 < base; i0 += THREADS) {
      const int i = i0 + t;
      dst = -1;
      if (i < base) {
        const u64 key = pool[i];
        if (key >= kth) { dst = rank_desc(gs.sh.key, kept, key); bx = pbox[i]; }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (dst >= 0) { store_wt(&pool[dst], gs.sh.key[dst]); store_wt(&pbox[dst], bx); }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    base = kept;
  }
  if (t == 0) store_wt(&p.pool_count[(size_t)b * p.ngroups + g], base | YP_POOL_CROWDED);   // (an oversized group: see group_nms_body)
}

// Workgroup (g, b): the classes c of image b with (c-1) % ngroups == g (detection.py:217-239).
template <int THREADS, bool ALONE>
__device__ __forceinline__ void group_nms_body(const NmsArgs& p, int b, int g, GroupShared& gs) {
  constexpr int KPT = YP_GROUP_KPT;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const size_t li = (size_t)b * YP_NLISTS + g;
  const u64* const lk = p.lkeys + li * p.lcap;
  const float4* const lb = p.lbox + li * p.lcap;
  u64* const pool = p.pool + ((size_t)b * p.ngroups + g) * p.gcap;
  float4* const pbox = p.pool_box + ((size_t)b * p.ngroups + g) * p.gcap;
  STAMP_INIT(0, b * p.ngroups + g);
  STAMP(0);
  // 1. ONE round trip: the list's length, the image's overflow count and the first THREADS candidates - key
  //    and box - loaded before the length is known (slots past the list's end hold stale entries, still inside
  //    the allocation)
  //    ONE thread reads the counters and resets the list's for the next launch (this workgroup is that counter's
  //    only reader; a reset by thread 0 while other waves still load the counter themselves would hand them a 0);
  //    everybody else gets the values through LDS.
  if (t == 0) {
    int32_t* const my_count = p.lcount + (size_t)b * YP_NLISTS * YP_CNT_STRIDE + YP_CNT_IDX(g);
    const int c = *my_count;
    const int o = p.ocount[(size_t)b * YP_CNT_STRIDE];
    gs.m_raw = c;
    gs.novf_raw = o;
    if (c != 0) *my_count = 0;
  }
  u64 key[KPT];
  float4 box[KPT];
  {
    const long long i = min((long long)t, p.lcap - 1);
    key[0] = lk[i];
    box[0] = lb[i];
  }
  if (t < 256) { gs.cls_cnt[t] = 0; gs.cls_fill[t] = 0; reinterpret_cast<int*>(gs.psup)[t] = 0; }
  if (t == 0) { gs.npool = 0; gs.sh.hist = gs.hist; }
  __syncthreads();
  const int m = (int)min((long long)gs.m_raw, p.lcap);            // (workgroup-uniform)
  const int novf = (int)min((long long)gs.novf_raw, p.ocap);
  STAMP(1);
  if (m == 0 && novf == 0) {
    if (t == 0) store_wt(&p.pool_count[(size_t)b * p.ngroups + g], 0);
    return;
  }
  if (m > YP_GROUP_CAP || novf > 0) {
    group_nms_slow<THREADS>(p, b, g, m, novf, gs, pool, pbox);
    return;
  }
  //    (longer lists: the remaining candidates, now that the length is known)
#pragma unroll
  for (int u = 1; u < KPT; ++u) {
    key[u] = 0ull;
    box[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u * THREADS < m) {                                        // (workgroup-uniform)
      const int i = min(t + u * THREADS, m - 1);
      key[u] = lk[i];
      box[u] = lb[i];
    }
  }
  // 2. class histogram, class offsets (exclusive prefix over the 256 possible class ids) of the candidates and of
  //    the candidates inside top_k
  int cls[KPT];
#pragma unroll
  for (int u = 0; u < KPT; ++u) {
    cls[u] = -1;
    if (t + u * THREADS < m) {
      cls[u] = yp_seg_key_class(key[u]);
      atomicAdd(&gs.cls_cnt[cls[u]], 1);
    }
  }
  __syncthreads();
  int cv = 0, sv = 0, incl = 0, sincl = 0;
  if (t < 256) {
    cv = gs.cls_cnt[t];
    sv = min(cv, p.top_k);
    incl = wave_incl_scan(cv);
    sincl = wave_incl_scan(sv);
    if (lane == 63) { gs.sh.wtot[wave] = incl; gs.sh.wtot[4 + wave] = sincl; }
  }
  __syncthreads();
  if (t < 256) {
    int base = 0, sbase = 0;
    for (int w = 0; w < wave; ++w) { base += gs.sh.wtot[w]; sbase += gs.sh.wtot[4 + w]; }
    gs.cls_off[t] = base + incl - cv;
    gs.cls_soff[t] = sbase + sincl - sv;
    if (t == 255) gs.nsorted = sbase + sincl;
  }
  __syncthreads();
  STAMP(2);
  if (gs.nsorted > YP_GROUP_SORTED) {                             // (workgroup-uniform; more classes x top_k than the LDS boxes hold)
    group_nms_slow<THREADS>(p, b, g, m, novf, gs, pool, pbox);
    return;
  }
  // 3. bucket the keys by class (the order inside a bucket is arbitrary; the ranks below do not depend on it)
  int bpos[KPT];
#pragma unroll
  for (int u = 0; u < KPT; ++u) {
    bpos[u] = 0;
    if (cls[u] >= 0) {
      bpos[u] = gs.cls_off[cls[u]] + atomicAdd(&gs.cls_fill[cls[u]], 1);
      gs.bk[bpos[u]] = key[u];
    }
  }
  __syncthreads();
  STAMP(3);
  // 4. the rank of every candidate inside its class (= number of larger keys: tf.math.top_k order,
  //    detection.py:217), computed per BUCKET position - neighbouring threads then work on the same class, so the
  //    loop bounds agree across a wave and the reads of bk[] are broadcasts (per list position, with a wave's
  //    lanes in different classes, this step took 24 us of a saturated group's time) - and handed to the
  //    candidate's owner through LDS (the histogram's words, idle here).
  for (int i = t; i < m; i += THREADS) {
    const u64 mine = gs.bk[i];
    const int c = yp_seg_key_class(mine);
    const int lo = gs.cls_off[c], cnt = gs.cls_cnt[c];
    int r = 0, q = 0;
    for (; q + 8 <= cnt; q += 8) {                                // eight independent reads per round trip to LDS
      u64 kk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) kk[u] = gs.bk[lo + q + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) r += kk[u] > mine ? 1 : 0;
    }
    for (; q < cnt; ++q) r += gs.bk[lo + q] > mine ? 1 : 0;
    gs.hist[i] = r;
  }
  __syncthreads();                                                // bk has been read: box[] takes its place
  //    sidx = the candidate's place among its class's sorted top_k, or -1 (detection.py:219)
  int sidx[KPT];
#pragma unroll
  for (int u = 0; u < KPT; ++u) {
    sidx[u] = -1;
    if (cls[u] >= 0) {
      const int r = gs.hist[bpos[u]];
      if (r < p.top_k) sidx[u] = gs.cls_soff[cls[u]] + r;
    }
  }
  __syncthreads();                                                // the ranks have been read: cmask[] takes their place
#pragma unroll
  for (int u = 0; u < KPT; ++u)
    if (sidx[u] >= 0) {
      gs.box[sidx[u]] = box[u];
      gs.cmask[sidx[u]] = yp_cell_mask(box[u], 8.0f / p.img_w, 8.0f / p.img_h);
    }
  __syncthreads();
  STAMP(4);
  // 5. Fast NMS inside every class: a candidate survives unless a higher-ranked box of its class
  //    overlaps it by more than the threshold (detection.py:221-234; every candidate has p > conf).
  //    Small classes (the usual case: a handful of candidates each) are swept side by side, every class by its
  //    own share of the threads; large ones one after the other by everybody (no barrier in between: the
  //    classes' boxes and flags are disjoint).
  {
    const int ncls = (p.nseg - (g + 1)) / p.ngroups + 1;          // classes g+1, g+1+ngroups, ... <= nseg
    int kmax = 0;
    for (int c = g + 1; c <= p.nseg; c += p.ngroups) kmax = max(kmax, min(gs.cls_cnt[c], p.top_k));
    const int tpc = THREADS / ncls;                               // (ncls <= 256 / ngroups... <= THREADS)
    // (the cell-mask sweep for the small classes as well: 28.1 us per launch against 26.7 without a fill beside it)
    if (tpc >= 1 && ((kmax + 1) >> 1) <= tpc && kmax <= 64) {
      const int ci = t / tpc, tid = t - ci * tpc;
      if (ci < ncls) {
        const int c = g + 1 + ci * p.ngroups;
        const int lo = gs.cls_soff[c];
        pair_sweep(gs.box + lo, gs.psup + lo, min(gs.cls_cnt[c], p.top_k), p.nms_thresh, tid, tpc);
      }
    } else {
      for (int c = g + 1; c <= p.nseg; c += p.ngroups) {
        const int lo = gs.cls_soff[c];
        const int kc = min(gs.cls_cnt[c], p.top_k);
        pair_sweep_cells(gs.box + lo, gs.cmask + lo, gs.psup + lo, reinterpret_cast<uint32_t*>(gs.sh.key) + wave * YP_SWEEP_QUEUE, kc,
                         p.nms_thresh, wave, THREADS / YP_WAVE, lane);
      }
    }
  }
  __syncthreads();
  STAMP(5);
  // 6. the group's survivors, dense, for the cross-class selection (detection.py:234-239): key and box go from
  //    the owning thread's registers straight to the pool.  Only a group's best max_dets can be among the
  //    image's best max_dets: a longer list is cut to them here, by all groups at once, so that the image's
  //    selection always fits its LDS stage and its first batch of loads.
  int pslot[KPT];
#pragma unroll
  for (int u = 0; u < KPT; ++u) {
    pslot[u] = -1;
    if (sidx[u] >= 0 && gs.psup[sidx[u]] == 0) {
      pslot[u] = lds_append_slot(&gs.npool);
      key[u] = yp_fin_key(yp_key_score(key[u]), yp_seg_key_anchor(key[u]), cls[u]);
    }
  }
  __syncthreads();
  int np = gs.npool;
  // A pool cut from a crowd (config 4: a thousand survivors, most of their scores EQUAL floats just below 1.0 and
  // ordered by anchor alone) cannot be cut by score bins - every key lands in one bin, a thousand LDS atomics on one
  // address: such a group goes straight to the general selection and tells the image's selection so with its count.
  const int crowded = ALONE && np > 2 * YP_SORT_MAX ? YP_POOL_CROWDED : 0;
  const int want = min(p.max_dets, YP_SORT_MAX);
  if (np > want) {                                                // (workgroup-uniform)
    int m = -1;
    if (ALONE && !crowded) {
      // (box[] has been swept: hist[] - which shares its memory - and the selection's scratch are free)
#pragma unroll
      for (int i = 0; i < YP_RADIX_BINS / THREADS; ++i) gs.hist[t * (YP_RADIX_BINS / THREADS) + i] = 0;
      if (t < YP_SORT_MAX) gs.sh.rank[t] = 0;
      if (t == 0) gs.sh.fill = 0;
      __syncthreads();
      bool valid[KPT];
      int rk[KPT];
#pragma unroll
      for (int u = 0; u < KPT; ++u) valid[u] = pslot[u] >= 0;
      m = select_from_registers<THREADS, KPT>(key, valid, want, p.sel_base, p.sel_shift, gs.sh, rk);
      if (m >= 0) {
#pragma unroll
        for (int u = 0; u < KPT; ++u)
          if (pslot[u] >= 0) pslot[u] = rk[u] >= 0 && rk[u] < want ? rk[u] : -1;
        np = want;
      } else {
        __syncthreads();
      }
    }
    if (m < 0) {
#pragma unroll
      for (int u = 0; u < KPT; ++u)
        if (pslot[u] >= 0) gs.stage[pslot[u]] = key[u];           // (box[] has been swept: the stage takes its place)
      __syncthreads();
      const ListSrc ls{gs.stage, np};
      np = select_topk_sorted<THREADS>(ls, np, want, gs.sh, nullptr, 0);
      const u64 kth = gs.sh.key[np - 1];
#pragma unroll
      for (int u = 0; u < KPT; ++u)
        if (pslot[u] >= 0) pslot[u] = key[u] >= kth ? rank_desc(gs.sh.key, np, key[u]) : -1;
    }
  }
#pragma unroll
  for (int u = 0; u < KPT; ++u)
    if (pslot[u] >= 0) { store_wt(&pool[pslot[u]], key[u]); store_wt(&pbox[pslot[u]], box[u]); }
  if (t == 0) store_wt(&p.pool_count[(size_t)b * p.ngroups + g], np | crowded);
  STAMP(6);
}

// Per-class mode: cross-class top max_dets over the pooled survivors (detection.py:242-248), run by
// the workgroup of the image's class group that finished last.
template <int THREADS, bool ALONE>
__device__ __forceinline__ void final_select_body(const NmsArgs& p, int b, GroupShared& gs) {
  SelectShared& sh = gs.sh;
  int* const start = gs.start;
  const int t = threadIdx.x;
  STAMP_INIT(1, b);
  STAMP(0);
  // ONE round trip: the groups' list lengths, the first batch of their keys (and boxes), the survivor counts
  PoolSrc src{p.pool + (size_t)b * p.ngroups * p.gcap, start, p.ngroups, p.gcap, {}};
  const int sft = src.shift(), lpp = 1 << sft;
  const int sl = min(t, (p.ngroups << sft) - 1);
  const int pg = sl >> sft, pq = sl & (lpp - 1);
  const u64* const pl = src.base + (size_t)pg * p.gcap;
  const float4* const plb = p.pool_box + ((size_t)b * p.ngroups + pg) * p.gcap;
  // (only slots below `want` can hold anything: a group pools at most its best `want` survivors)
  const int want = min(min(p.max_dets, YP_SORT_MAX), p.gcap);
  float4 preb[YP_POOL_PRE_BOX];
#pragma unroll
  for (int u = 0; u < YP_POOL_PRE; ++u) {
    const int i = min(pq + u * lpp, want - 1);
    src.pre[u] = pl[i];
    if (u < YP_POOL_PRE_BOX) preb[u] = plb[i];
  }
#ifdef YP_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(12);
#endif
  int cnt = 0;
  if (t < p.ngroups) cnt = p.pool_count[(size_t)b * p.ngroups + t];
  int part = 0;
  for (int i = t; i < p.parts; i += THREADS) part += p.seg_surv[(size_t)b * p.parts + i];
  bool any_crowded = true;
  if (ALONE) {
    // (while the loads are under way) what the selection from registers expects to be zero
    if (t == 0) { sh.hist = gs.hist; sh.fill = 0; }
#pragma unroll
    for (int i = 0; i < YP_RADIX_BINS / THREADS; ++i) gs.hist[t * (YP_RADIX_BINS / THREADS) + i] = 0;
    if (t < YP_SORT_MAX) sh.rank[t] = 0;
    any_crowded = __syncthreads_or((cnt & YP_POOL_CROWDED) != 0) != 0;
  }
  cnt &= ~YP_POOL_CROWDED;
#ifdef YP_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(13);
#endif
  if (t < 64) {                                   // ngroups <= 16: exclusive prefix of the lengths in the first wave
    const int incl = wave_incl_scan(cnt);
    if (t < p.ngroups) start[t + 1] = incl;
    if (t == 0) start[0] = 0;
  }
  const int survivors = block_sum<THREADS>(part, sh.wtot);   // (its barriers also publish start[])
  const int n = start[p.ngroups];
  STAMP(1);
  int k = 0;
  bool done = false;
  if (ALONE && n > 0 && !any_crowded) {
    // every pooled key is in a register if no list is longer than the first batch (16 groups: always)
    bool in_regs = true;
    for (int g = 0; g < p.ngroups; ++g) in_regs = in_regs && (start[g + 1] - start[g]) <= lpp * YP_POOL_PRE;
    if (in_regs) {                                 // (workgroup-uniform)
      k = min(min(p.max_dets, YP_SORT_MAX), n);
      const int ng = start[pg + 1] - start[pg];
      bool valid[YP_POOL_PRE];
      int rk[YP_POOL_PRE];
#pragma unroll
      for (int u = 0; u < YP_POOL_PRE; ++u) valid[u] = t < (p.ngroups << sft) && pq + u * lpp < ng;
      const int m = select_from_registers<THREADS, YP_POOL_PRE>(src.pre, valid, k, p.sel_base, p.sel_shift, sh, rk);
      if (m >= 0) {
        STAMP(2);
        // every thread that holds one of the selected keys writes that detection's row, at the rank it was told
#pragma unroll
        for (int u = 0; u < YP_POOL_PRE; ++u)
          if (rk[u] >= 0 && rk[u] < k) {
            const u64 key = src.pre[u];
            const float4 bx = u < YP_POOL_PRE_BOX ? preb[u < YP_POOL_PRE_BOX ? u : 0] : plb[pq + u * lpp];
            emit_row(p, b, rk[u], yp_key_score(key), yp_fin_key_class(key), min(yp_fin_key_anchor(key), p.A - 1), bx);
          }
        done = true;
      } else {
        __syncthreads();
      }
    }
  }
  if (n > 0 && !done) {
#ifdef YP_STAMPS
    k = select_topk_sorted<THREADS>(src, n, min(p.max_dets, YP_SORT_MAX), sh, gs.stage, YP_GSTAGE_CAP, dbg_);
#else
    k = select_topk_sorted<THREADS>(src, n, min(p.max_dets, YP_SORT_MAX), sh, gs.stage, YP_GSTAGE_CAP);
#endif
    STAMP(2);
    // every thread that holds one of the selected keys writes that detection's row, at the key's rank
    if (t < (p.ngroups << sft)) {
      const u64 kth = sh.key[k - 1];
      const int ng = start[pg + 1] - start[pg];
#pragma unroll
      for (int u = 0; u < YP_POOL_PRE; ++u) {
        const int i = pq + u * lpp;
        if (i < ng && src.pre[u] >= kth) {
          const u64 key = src.pre[u];
          const float4 bx = u < YP_POOL_PRE_BOX ? preb[u < YP_POOL_PRE_BOX ? u : 0] : plb[i];
          emit_row(p, b, rank_desc(sh.key, k, key), yp_key_score(key), yp_fin_key_class(key), min(yp_fin_key_anchor(key), p.A - 1), bx);
        }
      }
      for (int i = pq + lpp * YP_POOL_PRE; i < ng; i += lpp) {   // (never with 16 groups: a group pools <= 256 keys)
        const u64 key = pl[i];
        if (key >= kth)
          emit_row(p, b, rank_desc(sh.key, k, key), yp_key_score(key), yp_fin_key_class(key), min(yp_fin_key_anchor(key), p.A - 1), plb[i]);
      }
    }
  }
  STAMP(3);
  finish_image<THREADS>(p, b, k, survivors);
  STAMP(4);
}

// Per-class mode, one launch: workgroup b*ngroups + g does class group g of image b; the group of an
// image that arrives last does the image's final selection; workgroups beyond nms_blocks zero-fill.
// Hand-off of the groups' pooled survivors (written by other workgroups, possibly on other XCDs) to
// the last arriver: the bytes are stored write-through (store_wt), every storing wave drains its
// stores, the workgroup meets at a barrier, one lane adds to the image's counter at agent scope; the
// workgroup whose add returns ngroups-1 acquires at agent scope (invalidates its CU's L1) and meets at
// a barrier before any of its waves loads the lists.
#ifndef YP_GROUP_MIN_WAVES
#define YP_GROUP_MIN_WAVES 6   // waves per SIMD the register allocation must allow: 6 = three 512-thread workgroups per CU
                               // (two NMS + one fill; measured per launch at config 2: 66 us at 4, 60 at 6, 61 at 8)
#endif
#ifndef YP_GROUP_MIN_WAVES_ALONE
#define YP_GROUP_MIN_WAVES_ALONE 4   // ... of the instantiation that is launched without fill workgroups (tracked output sets):
                                     // nothing has to fit beside the NMS workgroups then, and at 128 registers the kernel does
                                     // not spill (28.6 us against 30.3 per launch; under a fill it is 63.7 against 52.9)
#endif
template <int THREADS, int MIN_WAVES>
__global__ __launch_bounds__(THREADS, MIN_WAVES) void group_nms_kernel(const NmsArgs p) {
  if ((int)blockIdx.x >= p.nms_blocks) {   // workgroup-uniform
    fill_loop<THREADS>(p, (int)blockIdx.x - p.nms_blocks);
    return;
  }
  __shared__ GroupShared gs;
#ifdef YP_EXP_NMS_NOP   // experiment build (timing only, wrong results): how long does the fill take with no NMS beside it?
  if (threadIdx.x == 0 && (int)blockIdx.x % p.ngroups == 0) p.num_det[(int)blockIdx.x / p.ngroups] = 0;
  if (threadIdx.x == 0) p.lcount[((size_t)blockIdx.x / p.ngroups * YP_NLISTS) * YP_CNT_STRIDE + YP_CNT_IDX(blockIdx.x % p.ngroups)] = 0;
  return;
#endif
  const int b = (int)blockIdx.x / p.ngroups, g = (int)blockIdx.x - b * p.ngroups;
  // ALONE: the instantiation launched without fill workgroups (tracked output sets).  Its time is the latency chain
  // itself, and it has registers to spare: it takes the selections from registers (select_from_registers) that cost
  // the other instantiation - spilling at its 80-register bound, and hidden under the fill anyway - more than they save.
  constexpr bool ALONE = MIN_WAVES == YP_GROUP_MIN_WAVES_ALONE && YP_GROUP_MIN_WAVES_ALONE != YP_GROUP_MIN_WAVES;
  group_nms_body<THREADS, ALONE>(p, b, g, gs);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pool stores have completed
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(&p.arrive[b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == p.ngroups - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      p.arrive[b] = 0;                               // ready for the next launch (kernel boundary in between)
      p.ocount[(size_t)b * YP_CNT_STRIDE] = 0;       // every group of the image has read it
    }
    gs.last = last;
  }
  __syncthreads();
  if (!gs.last) return;
  final_select_body<THREADS, ALONE>(p, b, gs);
}

// The zero-fill alone (callers with no NMS launch to carry it: yp_assemble_masks; YP_TUNE_PREFILL=0).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void fill_kernel(const NmsArgs p) { fill_loop<THREADS>(p, (int)blockIdx.x); }

hipError_t yp_launch_fill(float* masks, uint8_t* bitmaps, long long P, int fill_rows, int max_dets, int batch, hipStream_t st) {
  constexpr int T = YP_GROUP_THREADS;
  NmsArgs a = {};
  a.fill_masks = masks; a.fill_bitmaps = masks ? bitmaps : nullptr;
  a.fill_P = P; a.fill_rows = fill_rows; a.max_dets = max_dets; a.batch = batch; a.fill_first = 0;
  long long nfill = 0;
  if (a.fill_masks) nfill += fill_chunks_per_image<T>(P, fill_rows, 4) * batch;
  if (a.fill_bitmaps) nfill += fill_chunks_per_image<T>(P, fill_rows, 1) * batch;
  if (nfill == 0) return hipSuccess;
  a.fill_first = 0; a.fill_last = nfill;
  a.fill_wgs = (int)(nfill < 4096 ? nfill : 4096);   // (alone in its launch: nothing queues behind the zeros)
  hipLaunchKernelGGL((fill_kernel<T>), dim3((unsigned)a.fill_wgs), dim3(T), 0, st, a);
  return hipGetLastError();
}

static NmsArgs make_args(const yp_handle* h, int batch, const float* pred_offset, const float* priors,
                         float* det_boxes, float* det_classes, float* det_scores, int32_t* num_det, int32_t* kept_idx) {
  NmsArgs a;
  a.lkeys = h->s.lkeys; a.lbox = h->s.lbox; a.lcount = h->s.lcount; a.okeys = h->s.okeys; a.ocount = h->s.ocount;
  a.seg_surv = h->s.seg_surv; a.stat_surv = h->s.stat_surv;
  a.off = reinterpret_cast<const float4*>(pred_offset); a.priors = reinterpret_cast<const float4*>(priors);
  a.pool = h->s.pool; a.pool_box = h->s.pool_box; a.pool_count = h->s.pool_count; a.crop = h->s.crop;
  a.sel_anchor = h->s.sel_anchor;
  a.det_boxes = det_boxes; a.det_classes = det_classes; a.det_scores = det_scores; a.num_det = num_det;
  a.kept_idx = kept_idx;
  a.A = h->cfg.num_anchors; a.nseg = h->nseg; a.top_k = h->cfg.top_k; a.max_dets = h->cfg.max_dets;
  a.sel_rows = h->sel_rows;
  a.nms_thresh = h->cfg.nms_thresh;
  {
    const float conf = h->cfg.conf_thresh > 0.0f ? h->cfg.conf_thresh : 0.0f;
    const uint32_t cb = __builtin_bit_cast(uint32_t, conf);
    a.sel_base = cb < 0x3F800000u ? cb : 0u;
    a.sel_shift = 0;
    while (((0x3F800000u - a.sel_base) >> a.sel_shift) > (uint32_t)(YP_RADIX_BINS - 1)) ++a.sel_shift;
  }
  a.img_h = (float)h->cfg.img_h; a.img_w = (float)h->cfg.img_w;
  a.proto_h = (float)h->cfg.proto_h; a.proto_w = (float)h->cfg.proto_w;
  a.use_crop = h->cfg.use_cropped_mask;
  {
    int longest;
    yp_plan_items(h, batch, &a.parts, &longest);
  }
  yp_plan_groups(h, &a.ngroups, &a.gcap);
  yp_plan_lists(h, batch, &a.lcap, &a.ocap);
#ifdef YP_STAMPS
  a.dbg = (unsigned long long*)h->dbg;
#endif
  return a;
}

#ifndef YP_FILL_WGS_PER_CU
#define YP_FILL_WGS_PER_CU 6   // fill workgroups of an NMS launch per compute unit (each walks its share of the chunks)
#endif

// NMS (both modes; in per-class mode including the final selection) plus, when fill_masks is given,
// the zero-fill of rows [0, fill_rows) of every image's masks (and bitmaps) as extra workgroups.
hipError_t yp_launch_nms(const yp_handle* h, int batch, const float* pred_offset, const float* priors, float* det_boxes,
                         float* det_classes, float* det_scores, int32_t* num_det, int32_t* kept_idx,
                         float* fill_masks, uint8_t* fill_bitmaps, int fill_rows, hipStream_t st, float4* crop_out) {
  NmsArgs a = make_args(h, batch, pred_offset, priors, det_boxes, det_classes, det_scores, num_det, kept_idx);
  if (crop_out) a.crop = crop_out;   // a tracked output set keeps the crop edges of its last call (yp_run_kept)
  a.arrive = h->s.arrive;
  a.fill_masks = fill_masks; a.fill_bitmaps = fill_masks ? fill_bitmaps : nullptr;
  a.fill_P = (long long)h->cfg.proto_h * h->cfg.proto_w; a.fill_rows = fill_rows; a.batch = batch;
  const long long max_wgs = YP_FILL_WGS_PER_CU > 0 ? (long long)h->num_cus * YP_FILL_WGS_PER_CU : 0x7fffffffll - (long long)batch * YP_NLISTS;   // (0: a workgroup per chunk)
  if (h->cfg.nms_mode == YP_NMS_CROSS_CLASS) {
    constexpr int T = YP_CC_THREADS;
    a.nms_blocks = batch;
    long long nfill = 0;
    if (a.fill_masks) nfill += fill_chunks_per_image<T>(a.fill_P, fill_rows, 4) * batch;
    if (a.fill_bitmaps) nfill += fill_chunks_per_image<T>(a.fill_P, fill_rows, 1) * batch;
    a.fill_first = 0; a.fill_last = nfill; a.fill_wgs = (int)(nfill < max_wgs ? nfill : max_wgs);
    hipLaunchKernelGGL((cc_nms_kernel<T>), dim3((unsigned)(a.nms_blocks + a.fill_wgs)), dim3(T), 0, st, a);
  } else {
    constexpr int T = YP_GROUP_THREADS;
    a.nms_blocks = batch * a.ngroups;
    long long nfill = 0;
    if (a.fill_masks) nfill += fill_chunks_per_image<T>(a.fill_P, fill_rows, 4) * batch;
    if (a.fill_bitmaps) nfill += fill_chunks_per_image<T>(a.fill_P, fill_rows, 1) * batch;
    a.fill_first = 0; a.fill_last = nfill; a.fill_wgs = (int)(nfill < max_wgs ? nfill : max_wgs);
    if (a.fill_wgs > 0) hipLaunchKernelGGL((group_nms_kernel<T, YP_GROUP_MIN_WAVES>), dim3((unsigned)(a.nms_blocks + a.fill_wgs)), dim3(T), 0, st, a);
    else hipLaunchKernelGGL((group_nms_kernel<T, YP_GROUP_MIN_WAVES_ALONE>), dim3((unsigned)a.nms_blocks), dim3(T), 0, st, a);
  }
  return hipGetLastError();
}
