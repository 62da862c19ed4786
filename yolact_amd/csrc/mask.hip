// Kernel 4: mask assembly  sigmoid(proto[P,32] @ coef[N,32]^T)  with the crop fused in.
//
// Replaces detection.py:89-90 (matmul + sigmoid), utils.crop's ten broadcast passes
// (utils/utils.py:119-133), detection.py:105-107 (clip, transpose to [N,ph,pw]) and the mask
// scatter of detection.py:113.
//
// HBM-bound by the N*4*P bytes of masks it writes per image (K=32 gives ~12 flop/byte, below the
// ~25 flop/byte balance of 157 TFLOP/s fp32 over ~6.3 TB/s).  A lane owns one proto pixel and
// keeps its 32 prototype values in VGPRs for the whole kernel, so the prototypes are read from
// HBM exactly once; the coefficients and crop edges of a detection are wave-uniform and come in
// through scalar loads.  The dot product is a K-ordered fma chain (YP-MATH) and is only evaluated
// by waves whose 64 pixels can lie inside the detection's crop box; everything else is zeros, written
// in bulk: by the fill workgroups that ride in the NMS launch (nms.hip fill_block; the default), or by
// pass Z below.
// Output is written [N][P] (pixel-contiguous), which is the reference's final layout.
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
  float* masks;            // [B,max_dets,P]
  uint8_t* bitmaps;        // [B,max_dets,P] or null
  int P, pw, max_dets, sel_rows, zero_pad;
  int prezeroed;           // 1: the NMS launch's fill workgroups have already zero-filled every row this kernel would
  float mask_threshold;
};

#define YP_MASK_BLOCK 256
#define YP_MASK_ROW4 3   // float4 per pixel row in LDS: 2 of data + 1 of padding (48-byte stride: conflict-free b128 reads)

// One workgroup = 256 consecutive proto pixels of one image; a lane owns one pixel.
//   1. the block's 256 x 32 prototype values (32 KiB, contiguous in HBM) are fetched with fully
//      coalesced NON-TEMPORAL 16-byte loads (read once, by nobody else: they must not displace the
//      output lines that live in L2 / Infinity Cache) and transposed through LDS, so that every lane
//      ends up with its pixel's 32 values in VGPRs;
//   2. the rows are zeros first - either zero-filled beforehand by the NMS launch's fill workgroups
//      or by this kernel's pass Z (16-byte stores, 1 KiB per wave-instruction);
//      pass R: the detections whose crop box can touch the wave's pixels (one bit each, found with
//      ctz) get the K-ordered fma chain + sigmoid + crop and are stored over the zeros.
// Workgroups of one image are placed on one XCD (the dispatcher deals consecutive workgroup ids
// round-robin over the 8 XCDs), so the partial cache lines at workgroup borders (mask rows are
// only 16-byte aligned: P*4 bytes is not a multiple of 128) merge in a single L2.
template <bool NT_STORE, int NG>   // NG: 64-detection groups the kernel looks at (sel_rows <= 64 * NG)
__global__ __launch_bounds__(YP_MASK_BLOCK) void mask_assemble_kernel(const MaskArgs p, int blocks_per_image, int batch, int by_xcd) {
  __shared__ float4 sproto[YP_MASK_BLOCK * YP_MASK_ROW4];
  typedef float nat4 __attribute__((ext_vector_type(4)));
  const int t = threadIdx.x;
  // by_xcd (batch >= 8, grid padded to a multiple of 8 images): image b on the XCD that takes the
  // workgroup ids congruent to b % 8.  Smaller batches would leave XCDs idle that way and take the
  // plain order instead.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = by_xcd ? (slot / blocks_per_image) * 8 + xcd : (int)blockIdx.x / blocks_per_image;
  if (b >= batch) return;
  const int base_pix = (by_xcd ? slot % blocks_per_image : (int)blockIdx.x % blocks_per_image) * YP_MASK_BLOCK;
  const int npx = min(YP_MASK_BLOCK, p.P - base_pix);
  const int pix = base_pix + t;
  const bool live = t < npx;
  const int N = __builtin_amdgcn_readfirstlane(min(p.num_det[b], p.sel_rows));   // wave-uniform
  // The block's 256 x 32 prototype values (32 KiB, contiguous in HBM): fully coalesced NON-TEMPORAL
  // 16-byte loads (read once, by nobody else), issued first so that they are in flight while the
  // detection rows are looked at.  Thread t's float4 number i is quad (t & 7) of pixel i*32 + (t >> 3).
  const nat4* src = reinterpret_cast<const nat4*>(p.proto + ((size_t)b * p.P + base_pix) * YP_MASK_DIM);
  nat4 x[YP_MASK_DIM / 4];
#pragma unroll
  for (int i = 0; i < YP_MASK_DIM / 4; ++i) {
    const int idx = i * YP_MASK_BLOCK + t;
    x[i] = __builtin_nontemporal_load(src + min(idx, npx * (YP_MASK_DIM / 4) - 1));
  }

  // Which detections can touch this wave's 64 pixels at all?  Lane l tests detection g*64+l against
  // a bounding range of the wave's pixels (a superset test: the per-pixel compares below decide the
  // values); the ballots give a bit per detection.  Rows of all other detections are plain zeros.  Everything wave-uniform is kept in SGPRs on purpose (readfirstlane where
  // the compiler cannot prove it), so that walking the bits is scalar work.
  const int lane = t & 63;
  const int w0 = base_pix + __builtin_amdgcn_readfirstlane(t >> 6) * 64, w1 = min(w0 + 63, p.P - 1);
  u64 rel[NG];
  int anch[NG];      // lane l: anchor of detection g*64 + l (same round trip as its crop edges)
#pragma unroll
  for (int g = 0; g < NG; ++g) { rel[g] = 0ull; anch[g] = 0; }
  if (w0 < p.P) {
    const int r0 = w0 / p.pw, r1 = w1 / p.pw;
    int c0 = w0 - r0 * p.pw, c1 = w1 - r1 * p.pw;
    if (r1 > r0) { c0 = 0; c1 = p.pw - 1; }
    const float fr0 = (float)r0, fr1 = (float)r1, fc0 = (float)c0, fc1 = (float)c1;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g * 64 < N) {
        const int n = g * 64 + lane;
        const float4 cb = p.crop[(size_t)b * p.sel_rows + min(n, N - 1)];
        if (p.anchor) anch[g] = p.anchor[(size_t)b * p.sel_rows + min(n, N - 1)];
        const bool hit = (n < N) && (fc1 >= cb.x) && (fc0 < cb.y) && (fr1 >= cb.z) && (fr0 < cb.w);
        const u64 m = __ballot(hit);
        rel[g] = ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) |
                 (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m);
      }
    }
  }
  // Rows already zero-filled: a workgroup none of whose waves is touched by any detection has
  // nothing to write.
  u64 any_rel = 0ull;
#pragma unroll
  for (int g = 0; g < NG; ++g) any_rel |= rel[g];
  if (p.prezeroed && __syncthreads_or(any_rel != 0ull) == 0) return;

  float pr[YP_MASK_DIM];
  {
    // The transpose goes through LDS two quads at a time (four passes over a 12 KiB buffer instead of one
    // pass over 36 KiB): LDS is what limits residency here, and with it the waves that hide each
    // other's latencies.
#pragma unroll
    for (int q = 0; q < YP_MASK_DIM / 8; ++q) {
      if (q > 0) __syncthreads();            // the previous pass has been read
      if (((t & 7) >> 1) == q) {
#pragma unroll
        for (int i = 0; i < YP_MASK_DIM / 4; ++i)
          sproto[(i * (YP_MASK_BLOCK / 8) + (t >> 3)) * YP_MASK_ROW4 + (t & 1)] = make_float4(x[i].x, x[i].y, x[i].z, x[i].w);
      }
      __syncthreads();
      const float4 lo = sproto[t * YP_MASK_ROW4 + 0], hi = sproto[t * YP_MASK_ROW4 + 1];
      pr[8 * q + 0] = lo.x; pr[8 * q + 1] = lo.y; pr[8 * q + 2] = lo.z; pr[8 * q + 3] = lo.w;
      pr[8 * q + 4] = hi.x; pr[8 * q + 5] = hi.y; pr[8 * q + 6] = hi.z; pr[8 * q + 7] = hi.w;
    }
  }
  const int r = pix / p.pw;
  const float frow = live ? (float)r : -1.0f;                  // tf.range(..., dtype=float32), utils/utils.py:119-120
  const float fcol = live ? (float)(pix - r * p.pw) : -1.0f;

  // Row addresses are a wave-uniform base (SGPRs) plus a 32-bit lane offset (global_store saddr form)
  float* orow = p.masks + ((size_t)b * p.max_dets) * p.P + base_pix;
  uint8_t* brow = p.bitmaps ? p.bitmaps + ((size_t)b * p.max_dets) * p.P + base_pix : nullptr;
  typedef const __attribute__((address_space(4))) float cfloat;
  cfloat* const crop_c = (cfloat*)(p.crop + (size_t)b * p.sel_rows);
  cfloat* const coef_c = (cfloat*)(p.coef + (size_t)b * (p.anchor ? (size_t)p.A : (size_t)p.sel_rows) * YP_MASK_DIM);

  // The CU's scalar unit is shared by all its waves, and a loop over every detection with a test and
  // a branch per row is bound by it (measured: ~36 of 62 us with no memory traffic at all).  So only
  // the rows whose bit is set (found with ctz) are visited at all.
  const uint32_t voff = (uint32_t)t * 4u;
  const uint32_t rstride = (uint32_t)p.P * 4u;     // yp_create bounds sel_rows * P * 4 below 2^32
  const bool plain = !NT_STORE && brow == nullptr;
  //   pass Z  (only when the rows were not zero-filled beforehand) every row gets zeros: one store +
  //           one vector add per row, nothing scalar but the loop
  //   pass R  only the rows whose bit is set are computed and stored again, after the zeros have
  //           been acknowledged (vmcnt(0), plus a barrier where another wave wrote them)
  const int lane_ = t & 63;
  const bool wide = plain && (p.P & 3) == 0 && ((uintptr_t)p.masks & 15u) == 0;
  if (p.prezeroed) {
    // nothing: the rows are zero already and only the detections that touch this wave are rewritten
  } else if (wide) {
    // the workgroup's 256 pixels of a row are 1 KiB: one 16-byte store per lane zeroes a whole row
    // segment (eight full lines), wave w taking rows w, w+4, ...  The rows are then rewritten by the
    // waves that own their pixels, so these stores must have completed first.
    const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
    if (lane_ < (npx >> 2)) {
      uint32_t vo = (uint32_t)lane_ * 16u + (uint32_t)wv * rstride;
      const nat4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
      const int nz = p.zero_pad ? p.max_dets : N;   // tf.zeros rows of detection.py:68 included
      for (int n = wv; n < nz; n += YP_MASK_BLOCK / 64, vo += (YP_MASK_BLOCK / 64) * rstride)
        asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(vo), "v"(zero4), "s"(orow) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  } else if (live) {
    if (plain) {
      uint32_t vo = voff;
      const float zero = 0.0f;
      for (int n = 0; n < N; ++n, vo += rstride)
        asm volatile("global_store_dword %0, %1, %2" ::"v"(vo), "v"(zero), "s"(orow) : "memory");
    } else {
      for (int n = 0; n < N; ++n) {
        if (NT_STORE) __builtin_nontemporal_store(0.0f, orow + (size_t)n * p.P + t); else orow[(size_t)n * p.P + t] = 0.0f;
        if (brow) brow[(size_t)n * p.P + t] = 0.0f > p.mask_threshold ? 1 : 0;               // train.py:497
      }
    }
  }
  // per-wave zero pass: this wave rewrites its own addresses below; its zeros are acknowledged first
  if (!p.prezeroed && !wide) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    u64 w = g * 64 < N ? rel[g] : 0ull;        // wave-uniform
    while (w != 0ull) {
      const int n = g * 64 + __builtin_ctzll(w);
      w &= w - 1ull;
      // Crop edges and coefficients of detection n are wave-uniform and were written by the previous
      // kernel: read them through the constant address space so they arrive by scalar loads (SGPRs,
      // scalar cache) and never wait on this wave's own outstanding vector stores.
      float4 cb;
      cb.x = crop_c[4 * n + 0]; cb.y = crop_c[4 * n + 1]; cb.z = crop_c[4 * n + 2]; cb.w = crop_c[4 * n + 3];
      // the detection's coefficient row: by its anchor, straight from pred_mask_coef (a wave-uniform index: the
      // lane that tested detection n holds the anchor)
      const int crow = p.anchor ? __builtin_amdgcn_readlane(anch[g], n & 63) : n;
      cfloat* const cf = coef_c + (size_t)crow * YP_MASK_DIM;
      // mask_left & mask_right & mask_bottom & mask_top, float compares (utils/utils.py:125-130)
      const bool in = live && (fcol >= cb.x) && (fcol < cb.y) && (frow >= cb.z) && (frow < cb.w);
      if (__ballot(in) == 0ull) continue;      // the superset test let it through: the zeros stand
      float acc = 0.0f;
#pragma unroll
      for (int k = 0; k < YP_MASK_DIM; ++k) acc = __builtin_fmaf(pr[k], cf[k], acc);       // detection.py:89
      const float sg = yp_sigmoid(acc);                                                    // detection.py:90
      const float v = yp_clip(in ? sg : 0.0f, 0.0f, 1.0f);                                  // utils.py:133, L105
      if (live) {
        if (plain) {
          const uint32_t vo = voff + (uint32_t)n * rstride;
          asm volatile("global_store_dword %0, %1, %2" ::"v"(vo), "v"(v), "s"(orow) : "memory");
        } else {
          if (NT_STORE) __builtin_nontemporal_store(v, orow + (size_t)n * p.P + t); else orow[(size_t)n * p.P + t] = v;
          if (brow) brow[(size_t)n * p.P + t] = v > p.mask_threshold ? 1 : 0;                // train.py:497
        }
      }
    }
  }
  if (p.zero_pad && live && !wide && !p.prezeroed) {   // tf.zeros rows of detection.py:68 (the wide zero pass has written them already)
    float* const out_img = p.masks + (size_t)b * p.max_dets * p.P;
    uint8_t* const bit_img = p.bitmaps ? p.bitmaps + (size_t)b * p.max_dets * p.P : nullptr;
    for (int n = N; n < p.max_dets; ++n) {
      const size_t o = (size_t)n * p.P + pix;
      out_img[o] = 0.0f;
      if (bit_img) bit_img[o] = 0;
    }
  }
}

// Does this batch take the separate zero-fill (1: extra workgroups of the NMS launch, nms.hip
// fill_block) or mask_assemble's fused pass Z (0)?  The linear fill is the form in which HBM takes
// writes fastest, it does not depend on the NMS, and the NMS workgroups it shares a launch with are
// latency-bound and leave HBM idle; the fused pass (1 KiB pieces at a row stride) only keeps up while
// the whole output set lives in the 256 MiB Infinity Cache because the caller rewrites one buffer.
// The separate fill is the default; the fused pass stays selectable (YP_TUNE_PREFILL=0).
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

hipError_t yp_launch_mask_assemble(const yp_handle* h, int batch, const float* proto, const float* pred_mask_coef,
                                   const int32_t* num_det, float* det_masks, uint8_t* det_bitmaps, hipStream_t st) {
  MaskArgs a;
  a.proto = proto; a.coef = pred_mask_coef; a.anchor = h->s.sel_anchor; a.A = h->cfg.num_anchors;
  a.crop = h->s.crop; a.num_det = num_det;
  a.masks = det_masks; a.bitmaps = det_bitmaps;
  a.P = h->cfg.proto_h * h->cfg.proto_w; a.pw = h->cfg.proto_w;
  a.max_dets = h->cfg.max_dets; a.sel_rows = h->sel_rows; a.zero_pad = h->cfg.zero_pad_masks;
  a.prezeroed = yp_mask_prefill(h, batch) ? 1 : 0;
  a.mask_threshold = h->cfg.mask_threshold;
  const int bpi = (a.P + YP_MASK_BLOCK - 1) / YP_MASK_BLOCK;
  const int by_xcd = batch >= 8 ? 1 : 0;
  const dim3 grid(by_xcd ? 8 * ((batch + 7) / 8) * bpi : batch * bpi);
  // Default-policy stores: the zeros the fill has just written usually still sit (dirty) in the
  // Infinity Cache, and these stores overwrite them there.  (tune.mask_nt: non-temporal instead.)
  const int ng = (a.sel_rows + 63) / 64;
#define YP_MASK_LAUNCH(NT, NG) hipLaunchKernelGGL((mask_assemble_kernel<NT, NG>), grid, dim3(YP_MASK_BLOCK), 0, st, a, bpi, batch, by_xcd)
  if (h->tune.mask_nt) { if (ng <= 1) YP_MASK_LAUNCH(true, 1); else if (ng <= 2) YP_MASK_LAUNCH(true, 2); else YP_MASK_LAUNCH(true, 4); }
  else { if (ng <= 1) YP_MASK_LAUNCH(false, 1); else if (ng <= 2) YP_MASK_LAUNCH(false, 2); else YP_MASK_LAUNCH(false, 4); }
  return hipGetLastError();
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
  const hipError_t e = yp_launch_crop_edges(batch * rows, ph, pw, img_h, img_w, use_crop, boxes, crop_ws, st);
  if (e != hipSuccess) return e;
  MaskArgs a;
  a.proto = proto; a.coef = coef; a.anchor = nullptr; a.A = 0;
  a.crop = reinterpret_cast<const float4*>(crop_ws); a.num_det = num_rows;
  a.masks = out_masks; a.bitmaps = nullptr;
  a.P = ph * pw; a.pw = pw; a.max_dets = rows; a.sel_rows = rows; a.zero_pad = 1; a.prezeroed = 0;
  a.mask_threshold = -1.0f;
  const int bpi = (a.P + YP_MASK_BLOCK - 1) / YP_MASK_BLOCK;
  const int by_xcd = batch >= 8 ? 1 : 0;
  const dim3 grid(by_xcd ? 8 * ((batch + 7) / 8) * bpi : batch * bpi);
  hipLaunchKernelGGL((mask_assemble_kernel<false, 4>), grid, dim3(YP_MASK_BLOCK), 0, st, a, bpi, batch, by_xcd);
  return hipGetLastError();
}
