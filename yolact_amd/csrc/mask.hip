// Kernel 4: mask assembly  sigmoid(proto[P,32] @ coef[N,32]^T)  with the crop fused in.
//
// Replaces detection.py:89-90 (matmul + sigmoid), utils.crop's ten broadcast passes
// (utils/utils.py:119-133), detection.py:105-107 (clip, transpose to [N,ph,pw]) and the mask
// scatter of detection.py:113.
//
// HBM-bound by the prototypes it reads (4*P*32 bytes per image, once) - the mask rows are zeros already,
// written in bulk by the fill workgroups that ride in the NMS launch (nms.hip fill_block) or by a fill
// launch of their own - and, beside that, by the vector ALU: every detection a wave visits costs a 32-term
// fma chain and a sigmoid on all 64 lanes, whatever share of them lies inside the box.  So a wave owns an
// 8 x 8 TILE of proto pixels, not 64 consecutive ones: a crop box of 10 x 11 pixels meets ~5 such tiles with
// ~45 % of their lanes inside, where it meets ~14 runs of 64 pixels with ~12 % inside (config 2: 2.0 instead
// of 4.5 visited detections per wave).  A lane keeps its pixel's 32 prototype values in VGPRs; the
// coefficients and crop edges of a detection are wave-uniform and come in through scalar loads, the
// coefficient row straight from pred_mask_coef by the detection's anchor.  The dot product is a K-ordered
// fma chain (YP-MATH).  Output is written [N][P] (pixel-contiguous), which is the reference's final layout.
#include <stdlib.h>
#include "common.hpp"

struct MaskArgs {
  const float* proto;      // [B,P,32]
  const float* coef;       // anchor != null: pred_mask_coef [B,A,32], row (b, anchor[b][n]) belongs to detection n;
                           // anchor == null: [B,sel_rows,32], row (b, n)                       (detection.py:206)
  const int32_t* anchor;   // [B,sel_rows] anchors of the kept detections (written by the NMS launch), or null
  int A;
  const float4* crop;      // [B,sel_rows]  (x1,x2,y1,y2)
  const int32_t* num_det;  // [B]
  float* masks;            // [B,max_dets,P]; every row this kernel can touch holds zeros already
  uint8_t* bitmaps;        // [B,max_dets,P] or null (zeros already)
  int P, pw, ph, max_dets, sel_rows, tiles_x;
  float mask_threshold;
  // tracked output sets (yp_run_kept; TRACK kernels only): what the call before wrote into these mask buffers
  const float4* crop_prev;   // [B,sel_rows]
  const int32_t* num_prev;   // [B]
  int32_t* num_new;          // [B] this call's count, for the next call
};

#define YP_MASK_BLOCK 256
#define YP_MASK_TILE 16  // a workgroup owns a 16 x 16 tile of proto pixels: four waves, an 8 x 8 quadrant each
#ifndef YP_MASK_WAVES_PER_EU
#define YP_MASK_WAVES_PER_EU 1   // no bound: 66 VGPRs, 7 waves per SIMD.  Bound to 8 (64 VGPRs) the kernel spills 4: 28.5 us against 22.9
#endif
#ifndef YP_MASK_PRE
#define YP_MASK_PRE 1    // detections per wave whose coefficients are prefetched before the LDS transpose
#endif                   // (1 / 2 / 3 / 4 / 6: 22.9 / 22.9 / 23.1 / 23.1 / 23.2 us at config 2, 41.3 / 41.6 / 41.8 / 42.4 / 43.5 at 13 % box coverage)
#define YP_MASK_ROW4 3   // float4 per pixel row in LDS: 2 of data + 1 of padding (48-byte stride: conflict-free b128 reads)

// One workgroup = one 16 x 16 tile of one image's prototypes; wave w owns the quadrant (w >> 1, w & 1), lane l
// its pixel (l >> 3, l & 7).
//   1. the tile's 256 x 32 prototype values - sixteen runs of 2 KiB in HBM - are fetched with coalesced
//      NON-TEMPORAL 16-byte loads (read once, by nobody else: they must not displace the output lines
//      that live in L2 / Infinity Cache) and transposed through LDS, so that every lane ends up with its
//      pixel's 32 values in VGPRs;
//   2. the detections whose crop box can touch the wave's quadrant (one bit each, found with ctz) get the
//      K-ordered fma chain + sigmoid + crop and are stored over the zeros.
// Workgroups of one image are placed on one XCD (the dispatcher deals consecutive workgroup ids
// round-robin over the 8 XCDs), so the partial cache lines of neighbouring tiles merge in a single L2.
// TRACK (yp_run_kept): nobody zero-filled the rows.  They hold what the call before wrote - values inside ITS crop
// boxes, zeros elsewhere - so a wave also looks at that call's crop rows and stores zeros on the pixels of those boxes
// that this call's box of the same row does not cover.
template <bool NT_STORE, int NG, bool TRACK>   // NG: 64-detection groups the kernel looks at (sel_rows <= 64 * NG)
__global__ __launch_bounds__(YP_MASK_BLOCK, YP_MASK_WAVES_PER_EU) void mask_assemble_kernel(const MaskArgs p, int blocks_per_image, int batch, int by_xcd) {
  __shared__ float4 sproto[YP_MASK_BLOCK * YP_MASK_ROW4];
  typedef float nat4 __attribute__((ext_vector_type(4)));
  const int t = threadIdx.x;
  // by_xcd (batch >= 8, grid padded to a multiple of 8 images): image b on the XCD that takes the
  // workgroup ids congruent to b % 8.  Smaller batches would leave XCDs idle that way and take the
  // plain order instead.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = by_xcd ? (slot / blocks_per_image) * 8 + xcd : (int)blockIdx.x / blocks_per_image;
  if (b >= batch) return;
  const int tile = by_xcd ? slot % blocks_per_image : (int)blockIdx.x % blocks_per_image;
  const int ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
  const int row0 = ty * YP_MASK_TILE, col0 = tx * YP_MASK_TILE;      // the tile's first pixel
  // The detections of the image: the count arrives by a scalar load that nothing below waits for until
  // the crop rows are tested (their loads are issued without it: rows >= N hold stale values, masked).
  const int N = __builtin_amdgcn_readfirstlane(min(p.num_det[b], p.sel_rows));   // wave-uniform
  // The tile's prototypes: thread t's float4 number i is quad (t & 7) of the tile's pixel
  // (2i + (t >> 7), (t & 127) >> 3) - every load instruction of the workgroup covers two whole tile rows
  // (2 x 2 KiB).  Pixels past the image's edge are read from the edge pixel instead (and never used).
  // First in the queue: the crop edges and anchors of the image's detections (lane l: detection g*64 + l) - vector
  // loads return in order, so the relevance test below waits for these only, not for the prototypes behind them.
  const int lane = t & 63;
  float4 cbv[NG];
  int anch[NG];      // lane l: anchor of detection g*64 + l (same round trip as its crop edges)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    cbv[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    anch[g] = 0;
    if (g * 64 < p.sel_rows) {
      const int nc = min(g * 64 + lane, p.sel_rows - 1);
      cbv[g] = p.crop[(size_t)b * p.sel_rows + nc];
      if (p.anchor) anch[g] = p.anchor[(size_t)b * p.sel_rows + nc];
    }
  }
  float4 cbp[NG];    // TRACK: the same rows of the call before
  int Np = 0;
  if (TRACK) {
    Np = __builtin_amdgcn_readfirstlane(min(p.num_prev[b], p.sel_rows));
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      cbp[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g * 64 < p.sel_rows) cbp[g] = p.crop_prev[(size_t)b * p.sel_rows + min(g * 64 + lane, p.sel_rows - 1)];
    }
    if (tile == 0 && t == 0) p.num_new[b] = N;
  }
  nat4 x[YP_MASK_DIM / 4];
  {
    const int pcol = min(col0 + ((t & 127) >> 3), p.pw - 1);
    const nat4* src = reinterpret_cast<const nat4*>(p.proto + (size_t)b * p.P * YP_MASK_DIM) + (t & 7);
#pragma unroll
    for (int i = 0; i < YP_MASK_DIM / 4; ++i) {
      const int prow = min(row0 + 2 * i + (t >> 7), p.ph - 1);
      x[i] = __builtin_nontemporal_load(src + ((size_t)prow * p.pw + pcol) * (YP_MASK_DIM / 4));
    }
  }

  // Which detections can touch this wave's quadrant at all?  Lane l tests detection g*64+l against the
  // quadrant's rows and columns (a superset test: the per-pixel compares below decide the values); the
  // ballots give a bit per detection.  Everything wave-uniform is kept in SGPRs on purpose (readfirstlane
  // where the compiler cannot prove it), so that walking the bits is scalar work.
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int qr0 = row0 + (wv >> 1) * 8, qc0 = col0 + (wv & 1) * 8;   // the quadrant's first pixel
  const int r = qr0 + (lane >> 3), c = qc0 + (lane & 7);             // this lane's pixel
  const bool live = r < p.ph && c < p.pw;
  u64 rel[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) rel[g] = 0ull;
  if (qr0 < p.ph && qc0 < p.pw) {
    const float fr0 = (float)qr0, fr1 = (float)min(qr0 + 7, p.ph - 1), fc0 = (float)qc0, fc1 = (float)min(qc0 + 7, p.pw - 1);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g * 64 < p.sel_rows) {
        const int n = g * 64 + lane;
        const float4 cb = cbv[g];
        const bool hit = (n < N) && (fc1 >= cb.x) && (fc0 < cb.y) && (fr1 >= cb.z) && (fr0 < cb.w);
        const u64 m = __ballot(hit);
        rel[g] = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                 (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
      }
    }
  }
  u64 relp[NG];      // TRACK: the detections of the call before whose box can touch the quadrant
#pragma unroll
  for (int g = 0; g < NG; ++g) relp[g] = 0ull;
  if (TRACK && qr0 < p.ph && qc0 < p.pw) {
    const float fr0 = (float)qr0, fr1 = (float)min(qr0 + 7, p.ph - 1), fc0 = (float)qc0, fc1 = (float)min(qc0 + 7, p.pw - 1);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g * 64 < p.sel_rows) {
        const float4 cb = cbp[g];
        const bool hit = (g * 64 + lane < Np) && (fc1 >= cb.x) && (fc0 < cb.y) && (fr1 >= cb.z) && (fr0 < cb.w);
        const u64 m = __ballot(hit);
        relp[g] = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                  (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
      }
    }
  }
  // A workgroup none of whose waves is touched by any detection has nothing to write.
  u64 any_rel = 0ull;
#pragma unroll
  for (int g = 0; g < NG; ++g) any_rel |= rel[g] | relp[g];
  if (__syncthreads_or(any_rel != 0ull) == 0) return;

  // The first YP_MASK_PRE detections the wave visits: their wave-uniform data - 32 coefficients and 4 crop edges
  // each - are fetched into one VGPR per detection now (lane k < 32: coefficient k of the detection's row, by its
  // anchor, straight from pred_mask_coef, detection.py:206; lanes 32..35: its crop edges), so that the round trip
  // passes during the LDS transpose below instead of after it.  (Later visits fetch theirs by scalar loads when
  // their turn comes: vector loads for them would wait behind the stores of the visits before - one counter for
  // both on gfx9.)
  int npre[YP_MASK_PRE];                        // wave-uniform; -1: none
  float prev[YP_MASK_PRE];
#pragma unroll
  for (int u = 0; u < YP_MASK_PRE; ++u) {
    int n = -1;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (n < 0 && rel[g] != 0ull) { n = g * 64 + __builtin_ctzll(rel[g]); rel[g] &= rel[g] - 1ull; }
    npre[u] = n;
    prev[u] = 0.0f;
    if (n >= 0) {
      int crow = n;
      if (p.anchor) {
#pragma unroll
        for (int g = 0; g < NG; ++g) if ((n >> 6) == g) crow = __builtin_amdgcn_readlane(anch[g], n & 63);
      }
      const float* const src = lane < YP_MASK_DIM
          ? p.coef + ((size_t)b * (p.anchor ? (size_t)p.A : (size_t)p.sel_rows) + crow) * YP_MASK_DIM + lane
          : reinterpret_cast<const float*>(p.crop + (size_t)b * p.sel_rows + n) + (lane & 3);
      if (lane < YP_MASK_DIM + 4) prev[u] = *src;
    }
  }

  float pr[YP_MASK_DIM];
  {
    // The transpose goes through LDS two quads at a time (four passes over a 12 KiB buffer instead of one
    // pass over 36 KiB): LDS is what limits residency here, and with it the waves that hide each
    // other's latencies.  The LDS row of a pixel is the id of the thread that owns it.
    // Thread t loaded the pixels (2i + (t >> 7), (t & 127) >> 3) of the tile: owner = quadrant << 6 | row in the
    // quadrant << 3 | column in it.  Everything but the row's part is the same for all eight loads.
    const int lcol = (t & 127) >> 3;
    const int own0 = ((lcol >> 3) << 6) | (lcol & 7);
#pragma unroll
    for (int q = 0; q < YP_MASK_DIM / 8; ++q) {
      if (q > 0) __syncthreads();            // the previous pass has been read
      if (((t & 7) >> 1) == q) {
#pragma unroll
        for (int i = 0; i < YP_MASK_DIM / 4; ++i) {
          const int lrow = 2 * i + (t >> 7);
          const int own = own0 | ((lrow >> 3) << 7) | ((lrow & 7) << 3);
          sproto[own * YP_MASK_ROW4 + (t & 1)] = make_float4(x[i].x, x[i].y, x[i].z, x[i].w);
        }
      }
      __syncthreads();
      const float4 lo = sproto[t * YP_MASK_ROW4 + 0], hi = sproto[t * YP_MASK_ROW4 + 1];
      pr[8 * q + 0] = lo.x; pr[8 * q + 1] = lo.y; pr[8 * q + 2] = lo.z; pr[8 * q + 3] = lo.w;
      pr[8 * q + 4] = hi.x; pr[8 * q + 5] = hi.y; pr[8 * q + 6] = hi.z; pr[8 * q + 7] = hi.w;
    }
  }
  const float frow = live ? (float)r : -1.0f;                  // tf.range(..., dtype=float32), utils/utils.py:119-120
  const float fcol = live ? (float)c : -1.0f;

  // Row addresses are a wave-uniform base (SGPRs) plus a 32-bit lane offset (global_store saddr form)
  float* orow = p.masks + ((size_t)b * p.max_dets) * p.P;
  uint8_t* brow = p.bitmaps ? p.bitmaps + ((size_t)b * p.max_dets) * p.P : nullptr;
  typedef const __attribute__((address_space(4))) float cfloat;
  cfloat* const crop_c = (cfloat*)(p.crop + (size_t)b * p.sel_rows);
  cfloat* const coef_c = (cfloat*)(p.coef + (size_t)b * (p.anchor ? (size_t)p.A : (size_t)p.sel_rows) * YP_MASK_DIM);

  // The CU's scalar unit is shared by all its waves, and a loop over every detection with a test and
  // a branch per row is bound by it.  So only the rows whose bit is set (found with ctz) are visited at all.
  const int pix = live ? r * p.pw + c : 0;
  const uint32_t voff = (uint32_t)pix * 4u;
  const uint32_t rstride = (uint32_t)p.P * 4u;     // yp_create bounds sel_rows * P * 4 below 2^32
  const bool plain = !NT_STORE && brow == nullptr;
  // One visited detection: crop test per pixel, K-ordered fma chain, sigmoid, store inside the box.
  auto visit = [&](int n, const float4 cb, auto coef_at) {
    // mask_left & mask_right & mask_bottom & mask_top, float compares (utils/utils.py:125-130)
    const bool in = live && (fcol >= cb.x) && (fcol < cb.y) && (frow >= cb.z) && (frow < cb.w);
    if (__ballot(in) == 0ull) return;          // the superset test let it through: the zeros stand
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < YP_MASK_DIM; ++k) acc = __builtin_fmaf(pr[k], coef_at(k), acc);   // detection.py:89
    const float sg = yp_sigmoid(acc);                                                    // detection.py:90
    const float v = yp_clip(in ? sg : 0.0f, 0.0f, 1.0f);                                  // utils.py:133, L105
    // only the pixels inside the box are written (the others hold their zeros): in a tile that is most of
    // the lanes, and the neighbouring tiles write the rest of the cache line
    if (in) {
      if (plain) {
        const uint32_t vo = voff + (uint32_t)n * rstride;
        asm volatile("global_store_dword %0, %1, %2" ::"v"(vo), "v"(v), "s"(orow) : "memory");
      } else {
        if (NT_STORE) __builtin_nontemporal_store(v, orow + (size_t)n * p.P + pix); else orow[(size_t)n * p.P + pix] = v;
        if (brow) brow[(size_t)n * p.P + pix] = v > p.mask_threshold ? 1 : 0;                // train.py:497
      }
    }
  };
  if (TRACK) {
    // the pixels the call before wrote that this call's box of the same row does not cover: back to zero
    cfloat* const cprev_c = (cfloat*)(p.crop_prev + (size_t)b * p.sel_rows);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      u64 w = relp[g];                          // wave-uniform
      while (w != 0ull) {
        const int n = g * 64 + __builtin_ctzll(w);
        w &= w - 1ull;
        float4 cp;
        cp.x = cprev_c[4 * n + 0]; cp.y = cprev_c[4 * n + 1]; cp.z = cprev_c[4 * n + 2]; cp.w = cprev_c[4 * n + 3];
        const bool was = live && (fcol >= cp.x) && (fcol < cp.y) && (frow >= cp.z) && (frow < cp.w);
        bool now = false;
        if (n < N) {                            // (wave-uniform) the row is in use again
          float4 cn;
          cn.x = crop_c[4 * n + 0]; cn.y = crop_c[4 * n + 1]; cn.z = crop_c[4 * n + 2]; cn.w = crop_c[4 * n + 3];
          now = live && (fcol >= cn.x) && (fcol < cn.y) && (frow >= cn.z) && (frow < cn.w);
        }
        if (was && !now) {
          orow[(size_t)n * p.P + pix] = 0.0f;
          if (brow) brow[(size_t)n * p.P + pix] = 0;
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < YP_MASK_PRE; ++u)
    if (npre[u] >= 0) {
      const int fv = __builtin_bit_cast(int, prev[u]);
      float4 cb;
      cb.x = __builtin_bit_cast(float, __builtin_amdgcn_readlane(fv, YP_MASK_DIM + 0));
      cb.y = __builtin_bit_cast(float, __builtin_amdgcn_readlane(fv, YP_MASK_DIM + 1));
      cb.z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(fv, YP_MASK_DIM + 2));
      cb.w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(fv, YP_MASK_DIM + 3));
      visit(npre[u], cb, [&](int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(fv, k)); });
    }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    u64 w = rel[g];                            // wave-uniform: what the prefetched visits left
    while (w != 0ull) {
      const int n = g * 64 + __builtin_ctzll(w);
      w &= w - 1ull;
      // Crop edges and coefficients of detection n are wave-uniform and were written by the previous
      // kernel (the coefficients by the network): read them through the constant address space so they
      // arrive by scalar loads (SGPRs, scalar cache) and never wait on this wave's own outstanding vector
      // stores.
      float4 cb;
      cb.x = crop_c[4 * n + 0]; cb.y = crop_c[4 * n + 1]; cb.z = crop_c[4 * n + 2]; cb.w = crop_c[4 * n + 3];
      const int crow = p.anchor ? __builtin_amdgcn_readlane(anch[g], n & 63) : n;
      cfloat* const cf = coef_c + (size_t)crow * YP_MASK_DIM;
      visit(n, cb, [&](int k) { return cf[k]; });
    }
  }
}

// Are the zeros written by extra workgroups of the NMS launch (1, the default: the fill depends on nothing the
// NMS computes, HBM is idle while those latency-bound workgroups wait on their loads, and riding in the same
// launch costs no second stream and no events) or by a fill launch of its own just before mask_assemble
// (0: YP_TUNE_PREFILL=0)?  Either way it is the same linear fill (nms.hip fill_block) - the form in which HBM
// takes writes fastest - over the same rows.
int yp_mask_prefill(const yp_handle* h, int batch) {
  (void)batch;
  return h->tune.prefill >= 0 ? h->tune.prefill != 0 : 1;
}

// Rows [0, yp_mask_fill_rows) of every image are zero-filled before mask_assemble runs: every row a
// detection can land in (and, with zero_pad_masks, the tf.zeros rows of detection.py:68 as well).
int yp_mask_fill_rows(const yp_handle* h) {
  const yp_config& c = h->cfg;
  if (c.zero_pad_masks) return c.max_dets;
  int rows = h->sel_rows;                                                   // min(max_dets, 256)
  if (c.nms_mode == YP_NMS_CROSS_CLASS && c.top_k < rows) rows = c.top_k;   // N <= k (detection.py:186)
  return rows;
}

static hipError_t launch_mask_kernel(const MaskArgs& a, int batch, int ph, bool nt, hipStream_t st) {
  const int tiles_y = (ph + YP_MASK_TILE - 1) / YP_MASK_TILE;
  const int bpi = a.tiles_x * tiles_y;
  const int by_xcd = batch >= 8 ? 1 : 0;
  const dim3 grid(by_xcd ? 8 * ((batch + 7) / 8) * bpi : batch * bpi);
  // Default-policy stores: the zeros the fill has just written usually still sit (dirty) in the
  // Infinity Cache, and these stores overwrite them there.  (tune.mask_nt: non-temporal instead.)
  const int ng = (a.sel_rows + 63) / 64;
#define YP_MASK_LAUNCH(NT, NG, TR) hipLaunchKernelGGL((mask_assemble_kernel<NT, NG, TR>), grid, dim3(YP_MASK_BLOCK), 0, st, a, bpi, batch, by_xcd)
  if (a.crop_prev) { if (ng <= 1) YP_MASK_LAUNCH(false, 1, true); else if (ng <= 2) YP_MASK_LAUNCH(false, 2, true); else YP_MASK_LAUNCH(false, 4, true); }
  else if (nt) { if (ng <= 1) YP_MASK_LAUNCH(true, 1, false); else if (ng <= 2) YP_MASK_LAUNCH(true, 2, false); else YP_MASK_LAUNCH(true, 4, false); }
  else { if (ng <= 1) YP_MASK_LAUNCH(false, 1, false); else if (ng <= 2) YP_MASK_LAUNCH(false, 2, false); else YP_MASK_LAUNCH(false, 4, false); }
#undef YP_MASK_LAUNCH
  return hipGetLastError();
}

hipError_t yp_launch_mask_assemble(const yp_handle* h, int batch, const float* proto, const float* pred_mask_coef,
                                   const int32_t* num_det, float* det_masks, uint8_t* det_bitmaps, hipStream_t st,
                                   const yp_track* track) {
  if (!track && !yp_mask_prefill(h, batch)) {   // the zeros did not ride in the NMS launch
    const hipError_t e = yp_launch_fill(det_masks, det_bitmaps, (long long)h->cfg.proto_h * h->cfg.proto_w, yp_mask_fill_rows(h),
                                        h->cfg.max_dets, batch, st);
    if (e != hipSuccess) return e;
  }
  MaskArgs a;
  a.proto = proto; a.coef = pred_mask_coef; a.anchor = h->s.sel_anchor; a.A = h->cfg.num_anchors;
  a.crop = h->s.crop; a.num_det = num_det;
  a.masks = det_masks; a.bitmaps = det_bitmaps;
  a.P = h->cfg.proto_h * h->cfg.proto_w; a.pw = h->cfg.proto_w; a.ph = h->cfg.proto_h;
  a.tiles_x = (a.pw + YP_MASK_TILE - 1) / YP_MASK_TILE;
  a.max_dets = h->cfg.max_dets; a.sel_rows = h->sel_rows;
  a.mask_threshold = h->cfg.mask_threshold;
  a.crop_prev = nullptr; a.num_prev = nullptr; a.num_new = nullptr;
  if (track) { a.crop = track->crop_new; a.crop_prev = track->crop_prev; a.num_prev = track->num_prev; a.num_new = track->num_new; }
  return launch_mask_kernel(a, batch, a.ph, h->tune.mask_nt != 0, st);
}

// ---- the same kernel for the training loss (the gradients: loss_grad.hip) ----------
// loss/loss_yolact.py:239-253: sigmoid(proto @ pos_coef^T) cropped by the positives' prior boxes, which
// arrive in pixels and are normalised by the image size (L247-252) before utils.crop (L253).
__global__ __launch_bounds__(256) void crop_edges_kernel(const float4* boxes /* [total] ymin,xmin,ymax,xmax pixels */,
                                                         float4* crop /* [total] x1,x2,y1,y2 proto pixels */, int total,
                                                         float img_h, float img_w, float proto_h, float proto_w, int use_crop) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const float4 bx = boxes[i];
  float4 cb;
  if (use_crop) {
    const float ny1 = bx.x / img_h, nx1 = bx.y / img_w, ny2 = bx.z / img_h, nx2 = bx.w / img_w;   // L247-252
    yp_sanitize(nx1, nx2, proto_w, true, 1.0f, cb.x, cb.y);                                        // utils/utils.py:116
    yp_sanitize(ny1, ny2, proto_h, true, 1.0f, cb.z, cb.w);                                        // utils/utils.py:117
  } else {
    cb.x = 0.0f; cb.y = proto_w; cb.z = 0.0f; cb.w = proto_h;
  }
  crop[i] = cb;
}

hipError_t yp_launch_crop_edges(int total, int ph, int pw, int img_h, int img_w, int use_crop, const float* boxes, float* crop_ws,
                                hipStream_t st) {
  hipLaunchKernelGGL(crop_edges_kernel, dim3((total + 255) / 256), dim3(256), 0, st, reinterpret_cast<const float4*>(boxes),
                     reinterpret_cast<float4*>(crop_ws), total, (float)img_h, (float)img_w, (float)ph, (float)pw, use_crop);
  return hipGetLastError();
}

hipError_t yp_launch_loss_masks(int batch, int rows, int ph, int pw, int img_h, int img_w, int use_crop, const float* proto,
                                const float* coef, const float* boxes, const int32_t* num_rows, float* crop_ws, float* out_masks,
                                hipStream_t st) {
  hipError_t e = yp_launch_crop_edges(batch * rows, ph, pw, img_h, img_w, use_crop, boxes, crop_ws, st);
  if (e != hipSuccess) return e;
  e = yp_launch_fill(out_masks, nullptr, (long long)ph * pw, rows, rows, batch, st);   // rows >= num_rows[b] stay zeros (L273 pads nothing; the caller slices)
  if (e != hipSuccess) return e;
  MaskArgs a;
  a.proto = proto; a.coef = coef; a.anchor = nullptr; a.A = 0;
  a.crop = reinterpret_cast<const float4*>(crop_ws); a.num_det = num_rows;
  a.masks = out_masks; a.bitmaps = nullptr;
  a.P = ph * pw; a.pw = pw; a.ph = ph; a.tiles_x = (pw + YP_MASK_TILE - 1) / YP_MASK_TILE;
  a.max_dets = rows; a.sel_rows = rows;
  a.mask_threshold = -1.0f;
  a.crop_prev = nullptr; a.num_prev = nullptr; a.num_new = nullptr;
  return launch_mask_kernel(a, batch, ph, false, st);
}
