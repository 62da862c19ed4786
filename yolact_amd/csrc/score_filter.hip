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
  int A, C, stride, nseg, ntiles, batch;
  int bufb;              // bytes of one LDS tile buffer (TA*stride*4 rounded up to 16)
  int nbuf;              // LDS tile buffers per wave: 2 or 3 (tiles in flight ahead of the current one: nbuf-1)
  int ndma;              // full 1 KiB LDS-DMA wave-instructions per tile: floor(TA*C*4 / 1024)
  int rem_lanes;         // lanes of the last, partial LDS-DMA instruction: (TA*C*4 % 1024) / 16
  int fast;              // 1: stride == C and every tile starts 16-byte aligned -> LDS-DMA path
#ifdef YP_SF_STAMPS
  unsigned long long* dbg;   // diagnostic build only: per-wave phase cycle sums
  int abl;                   // diagnostic build only: 1 = DMA only (no survivors), 2 = DMA + phase 1 only
  int stagger;               // diagnostic build only: random start delay in [0, stagger) x 64 cycles
#endif
  float conf, conf_lo;
  float pre_thresh;      // ln(1/conf - 1) + 1e-3: x_bg - x_fgmax must be below it to have p_fgmax > conf
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
// max of three floats in one instruction (no NaN canonicalisation passes; logits are finite)
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// max over a row of 16 lanes of NON-NEGATIVE floats: their bit patterns order like integers
__device__ __forceinline__ float grp16_max_nonneg(float f) {
  int v = __builtin_bit_cast(int, f);
  v = max(v, dpp_i<YP_DPP_ROR(8)>(v));
  v = max(v, dpp_i<YP_DPP_ROR(4)>(v));
  v = max(v, dpp_i<YP_DPP_ROR(2)>(v));
  v = max(v, dpp_i<YP_DPP_ROR(1)>(v));
  return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float max_nonneg(float a, float b) {
  return __builtin_bit_cast(float, max(__builtin_bit_cast(int, a), __builtin_bit_cast(int, b)));
}
__device__ __forceinline__ int grp16_sum_i(int v) {
  v = v + dpp_i<YP_DPP_ROR(8)>(v);
  v = v + dpp_i<YP_DPP_ROR(4)>(v);
  v = v + dpp_i<YP_DPP_ROR(2)>(v);
  v = v + dpp_i<YP_DPP_ROR(1)>(v);
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

// s_waitcnt vmcnt(n) with a run-time (wave-uniform) n: "wait until at most n vector-memory
// operations of this wave are outstanding".  Any smaller count is a stricter, still correct wait;
// counts above 31 are clamped to 31.
__device__ __forceinline__ void wait_vmcnt_le(int n) {
#define YP_VMC(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n < 31 ? n : 31) {
    YP_VMC(1) YP_VMC(2) YP_VMC(3) YP_VMC(4) YP_VMC(5) YP_VMC(6) YP_VMC(7) YP_VMC(8) YP_VMC(9) YP_VMC(10)
    YP_VMC(11) YP_VMC(12) YP_VMC(13) YP_VMC(14) YP_VMC(15) YP_VMC(16) YP_VMC(17) YP_VMC(18) YP_VMC(19) YP_VMC(20)
    YP_VMC(21) YP_VMC(22) YP_VMC(23) YP_VMC(24) YP_VMC(25) YP_VMC(26) YP_VMC(27) YP_VMC(28) YP_VMC(29) YP_VMC(30)
    YP_VMC(31)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef YP_VMC
}

// One PERSISTENT wave walks tiles of TA consecutive anchors (TA = 16, 32 or 64): tile blockIdx.x,
// blockIdx.x + gridDim.x, ...  Two LDS buffers: while tile i is being processed, the LDS-DMA of
// tile i+1 is in flight, so every resident wave always has HBM reads outstanding.
//   phase 0  the tile's TA*C logits, contiguous in HBM, go straight into LDS with LDS-DMA
//            (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR staging); always exactly
//            `ndma` full-wave instructions per tile (addresses past the tile are clamped, the
//            excess lands in the buffer's padding)
//   phase 1  64/TA lanes per anchor: x_bg, max foreground logit, row max; exact-safe early reject:
//            p_fgmax <= 1/(1+exp(x_bg-x_fgmax)), so an anchor with x_bg-x_fgmax >= ln(1/conf-1)+1e-3
//            cannot pass `class_p_max > conf` (detection.py:73) and needs no exponentials at all
//   phase 2  the remaining anchors, four at a time, 16 lanes per anchor: lane j owns classes
//            j, j+16, ... - exactly the 16 strided partial sums of the YP-MATH softmax - so the
//            denominator is a per-lane sequential sum plus one DPP tree (8,4,2,1)
//   phase 3  survivors' keys go to the image's candidate slots / per-class lists (stores only; the
//            boxes are decoded later, for the few candidates that reach the NMS).  The stores are
//            not waited for: the wait for tile i+1's DMA is a COUNTED vmcnt that leaves exactly the
//            store instructions issued after that DMA in flight.
template <int MODE, int CPL, int TA, bool STG>
__global__ __launch_bounds__(YP_WAVE) void score_filter_kernel(const ScoreFilterArgs p) {
  constexpr int LPA = YP_WAVE / TA;   // lanes per anchor in phase 1
  // STG: the tile in flight is held in VGPRs (NLD 16-byte loads per lane) instead of a second LDS
  // buffer, so a wave needs one LDS tile only and twice as many waves fit a CU.
  constexpr int NLD = STG ? (TA * CPL) / 16 : 1;
  typedef float nat4 __attribute__((ext_vector_type(4)));
  nat4 stg[NLD];
  constexpr int KCAP = 128;           // capacity of the key list
  extern __shared__ __align__(16) char lds_raw[];
  const int lane = threadIdx.x;
  const int C = p.C;
#ifdef YP_SF_STAMPS
  const int ablm = p.abl % 100;          // diagnostic build: ablation mode; +100 = write no outputs at all
  const bool nostore = p.abl >= 100;
#endif
  int2* s_list = reinterpret_cast<int2*>(lds_raw + p.nbuf * p.bufb);  // (tile-local anchor, row max) of anchors past phase 1
  u64* s_key = reinterpret_cast<u64*>(s_list + YP_WAVE);          // keys waiting to be written out
  const int grp = lane >> 4, j = lane & 15;
  int cidx[CPL];     // this lane's classes j, j+16, ... (clamped) and whether they exist
  float okf[CPL];
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    cidx[i] = min(j + 16 * i, C - 1);
    okf[i] = (j + 16 * i) < C ? 1.0f : 0.0f;
  }
  const u64 lt_mask = (1ull << lane) - 1ull;
  const int total_tiles = p.batch * p.ntiles;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void*)lds_raw;
  const uint32_t lane16 = (uint32_t)lane * 16u;
  const long long img_bytes = (long long)p.A * C * 4;
  const long long tile_bytes = (long long)TA * C * 4;
  const long long all_bytes = (long long)p.batch * img_bytes;

  // LDS-DMA of one whole tile: p.ndma full-wave instructions of 1 KiB plus one partial instruction
  // (p.rem_lanes lanes) - exactly the tile's bytes, so the buffers need no padding.  Inline asm, so
  // that the compiler's own s_waitcnt bookkeeping does not drain the prefetch of the NEXT tiles
  // before every LDS read of the current one (it cannot tell the buffers apart); the DMA is waited
  // for by hand with the counted wait at the top of the tile loop.  Scalar base + lane*16 +
  // immediate; the immediate offset applies to the LDS address as well, so four 1 KiB pieces share
  // one M0.
  auto issue_dma = [&](int bb, int tt, uint32_t buf_off) {
    const long long off = bb * img_bytes + tt * tile_bytes;
    const char* gbase = reinterpret_cast<const char*>(p.cls) + off;
    uint32_t dst = lds_base + buf_off;
    int left = p.ndma;
    if (off + tile_bytes > all_bytes) {
      // only the last tile of the batch: never read past the caller's allocation (clamped lanes)
      const char* const last = reinterpret_cast<const char*>(p.cls) + all_bytes - 16;
      for (; left >= 0; --left, gbase += 1024, dst += 1024) {
        const char* g = gbase + lane16;
        g = g < last ? g : last;
        if (left > 0 || lane < p.rem_lanes)
          asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" ::"v"(g), "s"(dst) : "memory");
      }
      return;
    }
    for (; left >= 4; left -= 4, gbase += 4096, dst += 4096)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, %1 nt\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:1024 nt\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:2048 nt\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:3072 nt" ::"v"(lane16), "s"(gbase), "s"(dst) : "memory");
    if (left == 3)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, %1 nt\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:1024 nt\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:2048 nt" ::"v"(lane16), "s"(gbase), "s"(dst) : "memory");
    else if (left == 2)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, %1 nt\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:1024 nt" ::"v"(lane16), "s"(gbase), "s"(dst) : "memory");
    else if (left == 1)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, %1 nt" ::"v"(lane16), "s"(gbase), "s"(dst) : "memory");
    if (lane < p.rem_lanes) {
      gbase += left * 1024;
      dst += left * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, %1 nt" ::"v"(lane16), "s"(gbase), "s"(dst) : "memory");
    }
  };
  const int nvm = p.ndma + (p.rem_lanes > 0 ? 1 : 0);   // vector-memory instructions per tile DMA
  // STG: the tile's bytes as NLD non-temporal 16-byte loads per lane (addresses past the tile are
  // clamped to its last 16 bytes; those registers land in LDS space nobody reads)
  auto issue_loads = [&](int bb, int tt) {
    const int nbytes = min(TA, p.A - tt * TA) * C * 4;
    const char* g = reinterpret_cast<const char*>(p.cls) + (bb * img_bytes + tt * tile_bytes);
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
      const int o = min(u * 1024 + (int)lane16, nbytes - 16);
      stg[u] = __builtin_nontemporal_load(reinterpret_cast<const nat4*>(g + o));
    }
  };

  int flat = blockIdx.x;
  if (flat >= total_tiles) return;
  const int G = gridDim.x;
  int b = flat / p.ntiles, tix = flat - b * p.ntiles;       // current tile
  int nb = b, ntix = tix;                                    // next tile of this wave
  auto advance = [&](int& bb, int& tt) {
    tt += G;
    while (tt >= p.ntiles) { tt -= p.ntiles; ++bb; }
  };
  advance(nb, ntix);
  // Prefetch queue: the DMAs of the current tile and of the next nbuf-2 tiles are in flight.
  // y1 / y2 = vector-memory instructions issued after the oldest / second pending DMA: the counted
  // wait for the oldest one leaves exactly those in flight.
  int pf_flat = flat, pb = b, ptix = tix, pbuf = 0, cbuf = 0;
  int npend = 0, y1 = 0, y2 = 0;
  auto note_vmem = [&](int k) { y1 += npend >= 1 ? k : 0; y2 += npend >= 2 ? k : 0; };
  auto prefetch = [&]() {
    if (pf_flat >= total_tiles) return;
    issue_dma(pb, ptix, (uint32_t)(pbuf * p.bufb));
    note_vmem(nvm);
    ++npend;
    pf_flat += G;
    advance(pb, ptix);
    pbuf = pbuf + 1 == p.nbuf ? 0 : pbuf + 1;
  };
  if (STG) { if (p.fast) issue_loads(b, tix); }
  else if (p.fast) for (int d = 1; d < p.nbuf; ++d) prefetch();
  int pend_flat = -1, pend_nk = 0;   // cross-class mode: tile whose outputs are still to be stored
#ifdef YP_SF_STAMPS
  long long dd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long Tstart = clock64();
#define STAMP(v) const long long v = clock64()
#else
#define STAMP(v)
#endif
  for (; flat < total_tiles; flat += G, cbuf = (cbuf + 1 == p.nbuf ? 0 : cbuf + 1), b = nb, tix = ntix, advance(nb, ntix)) {
    const int a0 = tix * TA;
    const int na = min(TA, p.A - a0);
    float* tile = reinterpret_cast<float*>(lds_raw + cbuf * p.bufb);

    // ---- phase 0 ------------------------------------------------------------------------------
    STAMP(T0);
    if (STG && p.fast) {
      nat4* t4 = reinterpret_cast<nat4*>(tile);
#pragma unroll
      for (int u = 0; u < NLD; ++u)
        if (u * 1024 + (int)lane16 < p.bufb) t4[u * YP_WAVE + lane] = stg[u];
      if (flat + G < total_tiles) issue_loads(nb, ntix);      // in flight while this tile is processed
      asm volatile("" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    } else if (p.fast) {
      wait_vmcnt_le(y1);                       // this tile's DMA has landed; everything younger stays in flight
      y1 = y2; y2 = 0; --npend;
#ifdef YP_SF_STAMPS
      dd[1] += clock64() - T0;
#endif
#ifdef YP_SF_STAMPS
      const long long Ti0 = clock64();
#endif
      prefetch();                              // into the buffer the previous tile has just left
#ifdef YP_SF_STAMPS
      dd[4] += clock64() - Ti0;
#endif
#ifdef YP_SF_STAMPS
      if (nostore) pend_flat = -1;
#endif
      if (MODE == YP_NMS_CROSS_CLASS && pend_flat >= 0) {   // previous tile's outputs (s_key is rewritten only in phase 2)
        if (lane == 0) p.tile_count[pend_flat] = pend_nk;
#ifdef YP_SF_STAMPS
        if (lane < (ablm == 19 ? 1 : max(pend_nk, 1))) p.cand[(size_t)pend_flat * TA + lane] = s_key[lane];
#else
        if (lane < max(pend_nk, 1)) p.cand[(size_t)pend_flat * TA + lane] = s_key[lane];
#endif
        note_vmem(2);
      }
    } else {
      const int nelem = na * C;
      const float* __restrict__ src = p.cls + ((size_t)b * p.A + a0) * C;
      for (int e = lane; e < nelem; e += YP_WAVE) {
        const int a = e / C;
        tile[a * p.stride + (e - a * C)] = src[e];
      }
      __syncthreads();
    }

    STAMP(T2);
    // ---- phase 1: LPA lanes per anchor --------------------------------------------------------
    const int al1 = lane / LPA, q = lane % LPA;
    float m = 0.0f;
    bool pre = false;
#ifdef YP_SF_STAMPS
    if (ablm != 1)
#endif
    {
      const float* row = tile + min(al1, na - 1) * p.stride;
      const float x0 = row[0];
      float xf = -INFINITY;
      int c0 = 1 + q;
      for (; c0 + LPA * 15 < C; c0 += 16 * LPA) {     // full chunks: base + immediate offsets
        float v[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) v[jj] = row[c0 + LPA * jj];
        const float t0 = max3f(v[0], v[1], v[2]), t1 = max3f(v[3], v[4], v[5]), t2 = max3f(v[6], v[7], v[8]);
        const float t3 = max3f(v[9], v[10], v[11]), t4 = max3f(v[12], v[13], v[14]);
        xf = max3f(max3f(t0, t1, t2), max3f(t3, t4, v[15]), xf);
      }
      if (c0 < C) {                                   // ragged tail, clamped
        float v[16];
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) v[jj] = row[min(c0 + LPA * jj, C - 1)];
        const float t0 = max3f(v[0], v[1], v[2]), t1 = max3f(v[3], v[4], v[5]), t2 = max3f(v[6], v[7], v[8]);
        const float t3 = max3f(v[9], v[10], v[11]), t4 = max3f(v[12], v[13], v[14]);
        xf = max3f(max3f(t0, t1, t2), max3f(t3, t4, v[15]), xf);
      }
      if (LPA >= 2) xf = fmaxf(xf, dpp_f<YP_DPP_XOR1>(xf));
      if (LPA >= 4) xf = fmaxf(xf, dpp_f<YP_DPP_XOR2>(xf));
      m = fmaxf(x0, xf);
      pre = (al1 < na) && (q == 0) && ((x0 - xf) < p.pre_thresh);
    }
#ifdef YP_SF_STAMPS
    if (ablm >= 2 && (ablm < 12 || ablm == 15 || ablm == 16)) pre = false;
    if (ablm == 15 || ablm == 16) {   // dense VALU (8 independent FMA chains), random (15) or fixed (16) amount per tile
      float a0_ = m, a1_ = m + 1.f, a2_ = m + 2.f, a3_ = m + 3.f, a4_ = m + 4.f, a5_ = m + 5.f, a6_ = m + 6.f, a7_ = m + 7.f;
      const int nblk = ablm == 16 ? 10 : (int)((((uint32_t)flat * 0x9E3779B1u) >> 20) % 21u);
      for (int bq = 0; bq < nblk; ++bq) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          a0_ = __builtin_fmaf(a0_, 1.0001f, 0.5f); a1_ = __builtin_fmaf(a1_, 1.0001f, 0.5f);
          a2_ = __builtin_fmaf(a2_, 1.0001f, 0.5f); a3_ = __builtin_fmaf(a3_, 1.0001f, 0.5f);
          a4_ = __builtin_fmaf(a4_, 1.0001f, 0.5f); a5_ = __builtin_fmaf(a5_, 1.0001f, 0.5f);
          a6_ = __builtin_fmaf(a6_, 1.0001f, 0.5f); a7_ = __builtin_fmaf(a7_, 1.0001f, 0.5f);
        }
      }
      if (a0_ + a1_ + a2_ + a3_ + a4_ + a5_ + a6_ + a7_ == 123.456f) s_list[lane] = make_int2(0, 0);
    }
    if (ablm == 11) {   // random amount of dependent FMA work per tile (mean 750): desynchronises the waves
      float acc = m;
      const int nf = (int)((((uint32_t)flat * 0x9E3779B1u) >> 20) % 1501u);
      for (int i = 0; i < nf; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f);
      if (acc == 123.456f) s_list[lane] = make_int2(0, 0);
    }
    if (ablm >= 6 && ablm <= 9) {   // phase 2 replaced by ~3k cycles of one instruction kind
      float acc = m;
      if (ablm == 6) { for (int i = 0; i < 750; ++i) acc = __builtin_fmaf(acc, 1.0001f, 0.5f); }
      if (ablm == 7) { for (int i = 0; i < 375; ++i) acc = acc + dpp_f<YP_DPP_ROR(1)>(acc); }
      if (ablm == 8) { for (int i = 0; i < 190; ++i) acc = __builtin_amdgcn_rcpf(acc) + 1.0f; }
      if (ablm == 9) { for (int i = 0; i < 300; ++i) acc = acc + (__ballot(acc > (float)i) ? 1.0f : 2.0f); }
      if (acc == 123.456f) s_list[lane] = make_int2(0, 0);
    }
    if (ablm == 4 || ablm == 5) {   // phase 2 replaced by an idle delay: fixed, or random per tile
      const int nd = ablm == 4 ? 6 : (int)((((uint32_t)flat * 0x9E3779B1u) >> 20) % 13u);
      for (int i = 0; i < nd; ++i) __builtin_amdgcn_s_sleep(8);
    }
#endif
    const u64 pm = __ballot(pre);
    const int npass = __popcll(pm);
    if (pre) s_list[__popcll(pm & lt_mask)] = make_int2(al1, __float_as_int(m));   // (tile-local anchor, row max)
    __builtin_amdgcn_wave_barrier();   // single wave: LDS operations execute in order

    STAMP(T3);
    // ---- phase 2: 16 lanes per anchor ---------------------------------------------------------
    // Straight-line on purpose: the CPL logits of a lane are fetched together, the CPL exponentials
    // are independent chains the scheduler can interleave, invalid class slots are multiplied by 0
    // instead of being branched around.
    int nk = 0;      // survivors so far (wave-uniform)
    int ncand = 0;   // per-class candidates parked in s_key (wave-uniform)
    for (int t0 = 0; t0 < npass; t0 += 4) {
      const int slot = t0 + grp;
      const bool act = slot < npass;
      const int2 ent = s_list[min(slot, npass - 1)];
      const int al = ent.x;
      const float ma = __int_as_float(ent.y);
      const float* r = tile + al * p.stride;
      float e[CPL];
#pragma unroll
      for (int i = 0; i < CPL; ++i) e[i] = r[cidx[i]];
#ifdef YP_SF_STAMPS
      if (ablm == 13) {   // no LDS reads of the tile in phase 2: synthetic logits
#pragma unroll
        for (int i = 0; i < CPL; ++i) e[i] = ma - 0.37f * (float)(cidx[i] + al);
      }
      if (ablm == 14) {   // LDS reads only, no arithmetic
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) sacc += e[i];
        if (sacc == 123.456f) s_key[lane] = 1;
        continue;
      }
#endif
#pragma unroll
      for (int i = 0; i < CPL; ++i) e[i] = yp_exp_le0(e[i] - ma) * okf[i];   // x <= row max; 0 for c >= C
      float s = e[0];                                                        // s_j = e_j + e_{j+16} + ...
#pragma unroll
      for (int i = 1; i < CPL; ++i) s = s + e[i];
      float em = j > 0 ? e[0] : 0.0f;                        // class 0 is background (detection.py:59)
#pragma unroll
      for (int i = 1; i < CPL; ++i) em = max_nonneg(em, e[i]);   // e >= 0: integer max on the bit patterns
#ifdef YP_SF_STAMPS
      if (ablm == 17) { if (s + em == 123.456f) s_key[lane] = 1; continue; }   // exponentials only
#endif
      const float S = grp16_sum(s);
      em = grp16_max_nonneg(em);
      // IEEE division is monotone, so max_c fl(e_c/S) = fl(max_c e_c / S)              (detection.py:62)
      const float pmax = em / S;
#ifdef YP_SF_STAMPS
      if (ablm == 18) { if (pmax == 123.456f) s_key[lane] = 1; continue; }    // + DPP reductions + division
#endif
      const bool surv = act && (pmax > p.conf);                                      // detection.py:73
      const bool lead = surv && j == 0;
      const u64 bl = __ballot(lead);
      if (MODE == YP_NMS_CROSS_CLASS) {
        // tf.argmax over the probabilities (detection.py:184): the FIRST class whose rounded
        // probability equals the maximum.  Only an e_c within 2^-21 of the largest can round to it;
        // almost always that is one class, found without any division.
        const float ethr = em * (1.0f - 0x1p-21f);
        int cnt = 0, cmin = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
          const bool hit = (e[i] >= ethr) && (i > 0 || j > 0);
          cnt += hit ? 1 : 0;
          cmin = hit ? min(cmin, j + 16 * i) : cmin;
        }
        cnt = grp16_sum_i(cnt);
        int arg = grp16_min(cmin);
#ifdef YP_SF_STAMPS
        if (ablm == 20) { cnt = 1; arg = 1; }
#endif
        if (__ballot(surv && cnt > 1)) {   // rare: several classes within 2^-21 -> exact test
          int a2 = 0x7fffffff;
#pragma unroll
          for (int i = 0; i < CPL; ++i) {
            const bool hit = (e[i] >= ethr) && (i > 0 || j > 0);
            if (hit && (e[i] / S == pmax)) a2 = min(a2, j + 16 * i);
          }
          a2 = grp16_min(a2);
          if (cnt > 1) arg = a2;
        }
        if (lead) s_key[nk + __popcll(bl & lt_mask)] = yp_seg_key(pmax, a0 + al, arg);
      } else {
        // per-class candidates: every (anchor, class) with p > conf                 (detection.py:233)
        // They are parked in LDS and appended to the per-class lists 64 at a time, so that the
        // returning atomics of a whole tile are in flight together instead of one after another.
        const float elo = S * p.conf_lo;
        int nhit = 0, csel = 0;
        float esel = 0.0f;
#pragma unroll
        for (int i = CPL - 1; i >= 0; --i) {
          const bool hit = surv && (e[i] >= elo) && (i > 0 || j > 0);
          nhit += hit ? 1 : 0;
          csel = hit ? j + 16 * i : csel;
          esel = hit ? e[i] : esel;
        }
        if (__ballot(nhit > 1) == 0) {
          // common: at most one class per lane
          const float pc = esel / S;
          const bool has = (nhit == 1) && (pc > p.conf);
          const u64 bm = __ballot(has);
          if (has) s_key[ncand + __popcll(bm & lt_mask)] = yp_seg_key(pc, a0 + al, csel);
          ncand += __popcll(bm);
        } else {
#pragma unroll
          for (int i = 0; i < CPL; ++i) {
            bool has = false;
            float pc = 0.0f;
            if (surv && (e[i] >= elo) && (i > 0 || j > 0)) {
              pc = e[i] / S;
              has = pc > p.conf;
            }
            const u64 bm = __ballot(has);
            if (bm) {
              if (ncand + __popcll(bm) > KCAP) {   // wave-uniform, rare: flush first
                __builtin_amdgcn_wave_barrier();
                for (int l0 = 0; l0 < ncand; l0 += YP_WAVE) {
                  const int l = l0 + lane;
                  if (l < ncand) {
                    const u64 key = s_key[l];
                    const size_t seg = (size_t)b * p.nseg + (yp_seg_key_class(key) - 1);
                    const int qpos = atomicAdd(&p.cand_count[seg], 1);
                    p.cand[seg * p.A + qpos] = key;
                  }
                }
                y1 = y2 = -1000;   // unknown mix of atomics and stores in flight: the next waits drain everything
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): s_key is rewritten below
                ncand = 0;
              }
              if (has) s_key[ncand + __popcll(bm & lt_mask)] = yp_seg_key(pc, a0 + al, j + 16 * i);
              ncand += __popcll(bm);
            }
          }
        }
        if (ncand > KCAP - YP_WAVE) {   // keep room for the next round's fast path (<= 64 keys)
          __builtin_amdgcn_wave_barrier();
          for (int l0 = 0; l0 < ncand; l0 += YP_WAVE) {
            const int l = l0 + lane;
            if (l < ncand) {
              const u64 key = s_key[l];
              const size_t seg = (size_t)b * p.nseg + (yp_seg_key_class(key) - 1);
              const int qpos = atomicAdd(&p.cand_count[seg], 1);
              p.cand[seg * p.A + qpos] = key;
            }
          }
          y1 = y2 = -1000;
          __builtin_amdgcn_s_waitcnt(0xC07F);
          ncand = 0;
        }
      }
      nk += __popcll(bl);
    }

    STAMP(T4);
    // ---- phase 3: survivors ---------------------------------------------------------------------
    __builtin_amdgcn_wave_barrier();
    if (MODE == YP_NMS_CROSS_CLASS && p.fast && !STG) {
      // deferred: the two stores are issued right AFTER the next iteration's DMA issue, so that they
      // never sit in front of it in this wave's in-order vector-memory queue
      pend_flat = flat;
      pend_nk = nk;
    } else {
      if (lane == 0) p.tile_count[flat] = nk;
      note_vmem(1);
      if (MODE == YP_NMS_CROSS_CLASS) {
        // always issued (lane 0 writes an ignored slot when nk == 0) so that the count is exact
        if (lane < max(nk, 1)) p.cand[(size_t)flat * TA + lane] = s_key[lane];
        note_vmem(1);
      } else {
        for (int l0 = 0; l0 < ncand; l0 += YP_WAVE) {
          const int l = l0 + lane;
          if (l < ncand) {
            const u64 key = s_key[l];
            const size_t seg = (size_t)b * p.nseg + (yp_seg_key_class(key) - 1);
            const int qpos = atomicAdd(&p.cand_count[seg], 1);   // the compiler drains vmcnt for qpos
            p.cand[seg * p.A + qpos] = key;
          }
        }
        // the returning atomics above forced a full drain before their stores: every pending DMA has
        // landed, and only those stores (one instruction per started batch of 64) can be in flight
        if (ncand > 0) {
          y1 = y2 = 0;
          note_vmem((ncand + YP_WAVE - 1) / YP_WAVE);
        }
      }
    }
    // LDS reads of this tile must have returned before the DMA issued at the top of the next
    // iteration overwrites the other buffer's successor; lgkmcnt(0) only.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#ifdef YP_SF_STAMPS
    { const long long T5 = clock64(); long long T1x = T0; if (p.fast) T1x = T2; (void)T1x;
      dd[0] += T2 - T0; dd[2] += T3 - T2; dd[3] += T4 - T3; (void)T5; dd[5] += 1; dd[6] += npass; }
#endif
  }   // tile loop
#ifdef YP_SF_STAMPS
  if (nostore) pend_flat = -1;
#endif
  if (MODE == YP_NMS_CROSS_CLASS && pend_flat >= 0) {
    if (lane == 0) p.tile_count[pend_flat] = pend_nk;
    if (lane < pend_nk) p.cand[(size_t)pend_flat * TA + lane] = s_key[lane];
  }
#ifdef YP_SF_STAMPS
  dd[7] = clock64() - Tstart;
  if (p.dbg && lane == 0) { unsigned long long* o = p.dbg + (size_t)blockIdx.x * 8; for (int i = 0; i < 8; ++i) o[i] = dd[i]; }
#endif
}

template <int MODE, int TA, bool STG>
static void launch_mode(int cpl, int nblocks, size_t lds, hipStream_t st, const ScoreFilterArgs& a) {
  const dim3 grid(nblocks);
#define YP_SF_CASE(N) case N: hipLaunchKernelGGL((score_filter_kernel<MODE, N, TA, STG>), grid, dim3(YP_WAVE), lds, st, a); break;
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
  const int TA = h->tile_anchors;
  a.ntiles = h->ntiles; a.batch = batch;
  a.ndma = h->sf_ndma; a.rem_lanes = h->sf_rem; a.bufb = h->sf_bufb; a.nbuf = h->sf_nbuf;
#ifdef YP_SF_STAMPS
  a.dbg = (unsigned long long*)h->dbg;
  a.abl = getenv("YP_SF_ABL") ? atoi(getenv("YP_SF_ABL")) : 0;
  a.stagger = getenv("YP_SF_STAGGER") ? atoi(getenv("YP_SF_STAGGER")) : 0;
#endif
  const bool fast_ok = (a.stride == a.C) && (((long long)a.A * a.C) % 4 == 0) && ((((uintptr_t)pred_cls) & 15u) == 0) &&
                       ((long long)batch * a.A * a.C / 4 < 0x7fffffffLL);
  a.fast = fast_ok ? 1 : 0;
  const size_t lds = (size_t)a.nbuf * a.bufb + YP_WAVE * sizeof(int2) + 128 * sizeof(u64);
  int cpl = (a.C + 15) / 16;                       // classes per lane in phase 2
  if (cpl == 7) cpl = 8; else if (cpl > 8 && cpl <= 12) cpl = 12; else if (cpl > 12) cpl = 16;
  // persistent grid: as many single-wave workgroups as the LDS lets the chip hold
  const int total_tiles = batch * h->ntiles;
  // (measured on MI355X with a residency census: LDS is handed out in units of 1280 bytes, 128 units
  //  per CU - the occupancy API reports more than that for some sizes; an over-subscribed persistent
  //  grid would run in two rounds)
  int per_cu = (int)(128 / ((lds + 1279) / 1280));
  per_cu = per_cu < 1 ? 1 : (per_cu > 16 ? 16 : per_cu);
  const int wpc = getenv("YP_SF_WAVES_PER_CU") ? atoi(getenv("YP_SF_WAVES_PER_CU")) : per_cu;
  const int nblocks = total_tiles < h->num_cus * wpc ? total_tiles : h->num_cus * wpc;
  const bool cc = h->cfg.nms_mode == YP_NMS_CROSS_CLASS;
  if (h->sf_stg) { if (cc) launch_mode<YP_NMS_CROSS_CLASS, 32, true>(cpl, nblocks, lds, st, a); else launch_mode<YP_NMS_PER_CLASS, 32, true>(cpl, nblocks, lds, st, a); }
  else if (TA == 64) { if (cc) launch_mode<YP_NMS_CROSS_CLASS, 64, false>(cpl, nblocks, lds, st, a); else launch_mode<YP_NMS_PER_CLASS, 64, false>(cpl, nblocks, lds, st, a); }
  else if (TA == 32) { if (cc) launch_mode<YP_NMS_CROSS_CLASS, 32, false>(cpl, nblocks, lds, st, a); else launch_mode<YP_NMS_PER_CLASS, 32, false>(cpl, nblocks, lds, st, a); }
  else { if (cc) launch_mode<YP_NMS_CROSS_CLASS, 16, false>(cpl, nblocks, lds, st, a); else launch_mode<YP_NMS_PER_CLASS, 16, false>(cpl, nblocks, lds, st, a); }
  return hipGetLastError();
}
