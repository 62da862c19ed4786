// Kernel 4: mask assembly  sigmoid(proto[P,32] @ coef[N,32]^T)  with the crop fused in.
//
// Replaces detection.py:89-90 (matmul + sigmoid), utils.crop's ten broadcast passes
// (utils/utils.py:119-133), detection.py:105-107 (clip, transpose to [N,ph,pw]) and the mask
// scatter of detection.py:113.
//
// HBM-bound by the N*4*P bytes of masks it writes per image (K=32 gives ~12 flop/byte, below the
// ~25 flop/byte balance of 157 TFLOP/s fp32 over ~6.3 TB/s).  A lane owns PPL consecutive proto
// pixels and keeps their 32 prototype values in VGPRs for the whole kernel, so the prototypes are
// read from HBM exactly once; the coefficients and crop edges of detection n are wave-uniform and
// come in through scalar loads.  The dot product is a K-ordered fma chain (YP-MATH) and is skipped
// for waves whose 64*PPL pixels lie entirely outside detection n's crop box - those only store
// zeros.  Output is written [N][P] (pixel-contiguous), which is the reference's final layout.
#include <stdlib.h>
#include "common.hpp"

struct MaskArgs {
  const float* proto;      // [B,P,32]
  const float* coef;       // [B,sel_rows,32]
  const float4* crop;      // [B,sel_rows]  (x1,x2,y1,y2)
  const int32_t* num_det;  // [B]
  float* masks;            // [B,max_dets,P]
  uint8_t* bitmaps;        // [B,max_dets,P] or null
  int P, pw, max_dets, sel_rows, zero_pad;
  float mask_threshold;
};

template <int PPL> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<4> { typedef float4 type; };

template <int PPL>
__global__ __launch_bounds__(256) void mask_assemble_kernel(const MaskArgs p) {
  const int b = blockIdx.y;
  const int pix0 = (blockIdx.x * 256 + threadIdx.x) * PPL;
  const bool live = pix0 < p.P;   // P % PPL == 0 is guaranteed by the launcher
  const int N = min(p.num_det[b], p.sel_rows);

  float pr[PPL][YP_MASK_DIM];
  float frow[PPL], fcol[PPL];
  if (live) {
    const float4* src = reinterpret_cast<const float4*>(p.proto + ((size_t)b * p.P + pix0) * YP_MASK_DIM);
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
#pragma unroll
      for (int v = 0; v < YP_MASK_DIM / 4; ++v) {
        const float4 x = src[q * (YP_MASK_DIM / 4) + v];
        pr[q][4 * v + 0] = x.x; pr[q][4 * v + 1] = x.y; pr[q][4 * v + 2] = x.z; pr[q][4 * v + 3] = x.w;
      }
      const int pix = pix0 + q;
      const int r = pix / p.pw;
      frow[q] = (float)r;                  // tf.range(..., dtype=float32), utils/utils.py:119-120
      fcol[q] = (float)(pix - r * p.pw);
    }
  } else {
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
      frow[q] = -1.0f; fcol[q] = -1.0f;
#pragma unroll
      for (int k = 0; k < YP_MASK_DIM; ++k) pr[q][k] = 0.0f;
    }
  }

  typedef typename VecT<PPL>::type vec_t;
  float* const out_img = p.masks + (size_t)b * p.max_dets * p.P;
  uint8_t* const bit_img = p.bitmaps ? p.bitmaps + (size_t)b * p.max_dets * p.P : nullptr;

  // Crop edges and coefficients of detection n are wave-uniform and were written by the previous
  // kernel: read them through the constant address space so they arrive by scalar loads (SGPRs,
  // scalar cache) and never wait on this wave's own outstanding vector stores.
  typedef const __attribute__((address_space(4))) float cfloat;
  cfloat* const crop_c = (cfloat*)(p.crop + (size_t)b * p.sel_rows);
  cfloat* const coef_c = (cfloat*)(p.coef + (size_t)b * p.sel_rows * YP_MASK_DIM);
  for (int n = 0; n < N; ++n) {
    float4 cb;
    cb.x = crop_c[4 * n + 0]; cb.y = crop_c[4 * n + 1]; cb.z = crop_c[4 * n + 2]; cb.w = crop_c[4 * n + 3];
    cfloat* const cf = coef_c + n * YP_MASK_DIM;
    bool in[PPL];
    bool any = false;
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
      // mask_left & mask_right & mask_bottom & mask_top, float compares (utils/utils.py:125-130)
      in[q] = live && (fcol[q] >= cb.x) && (fcol[q] < cb.y) && (frow[q] >= cb.z) && (frow[q] < cb.w);
      any = any || in[q];
    }
    float v[PPL];
#pragma unroll
    for (int q = 0; q < PPL; ++q) v[q] = 0.0f;
    if (__ballot(any)) {
      float acc[PPL];
#pragma unroll
      for (int q = 0; q < PPL; ++q) acc[q] = 0.0f;
#pragma unroll
      for (int k = 0; k < YP_MASK_DIM; ++k) {
        const float c = cf[k];
#pragma unroll
        for (int q = 0; q < PPL; ++q) acc[q] = __builtin_fmaf(pr[q][k], c, acc[q]);     // detection.py:89
      }
#pragma unroll
      for (int q = 0; q < PPL; ++q) {
        const float s = yp_sigmoid(acc[q]);                                              // detection.py:90
        v[q] = yp_clip(in[q] ? s : 0.0f, 0.0f, 1.0f);                                    // utils.py:133, L105
      }
    }
    if (live) {
      const size_t o = (size_t)n * p.P + pix0;
      if (PPL == 1) out_img[o] = v[0];
      else {
        vec_t w;
        float* wf = reinterpret_cast<float*>(&w);
#pragma unroll
        for (int q = 0; q < PPL; ++q) wf[q] = v[q];
        *reinterpret_cast<vec_t*>(out_img + o) = w;
      }
      if (bit_img) {
#pragma unroll
        for (int q = 0; q < PPL; ++q) bit_img[o + q] = v[q] > p.mask_threshold ? 1 : 0;   // train.py:497
      }
    }
  }
  if (p.zero_pad && live) {   // tf.zeros rows of detection.py:68
    for (int n = N; n < p.max_dets; ++n) {
      const size_t o = (size_t)n * p.P + pix0;
#pragma unroll
      for (int q = 0; q < PPL; ++q) {
        out_img[o + q] = 0.0f;
        if (bit_img) bit_img[o + q] = 0;
      }
    }
  }
}

hipError_t yp_launch_mask_assemble(const yp_handle* h, int batch, const float* proto, const int32_t* num_det,
                                   float* det_masks, uint8_t* det_bitmaps, hipStream_t st) {
  MaskArgs a;
  a.proto = proto; a.coef = h->s.coef; a.crop = h->s.crop; a.num_det = num_det;
  a.masks = det_masks; a.bitmaps = det_bitmaps;
  a.P = h->cfg.proto_h * h->cfg.proto_w; a.pw = h->cfg.proto_w;
  a.max_dets = h->cfg.max_dets; a.sel_rows = h->sel_rows; a.zero_pad = h->cfg.zero_pad_masks;
  a.mask_threshold = h->cfg.mask_threshold;
  const bool al16 = (((uintptr_t)det_masks) & 15u) == 0;
  const int want = getenv("YP_MASK_PPL") ? atoi(getenv("YP_MASK_PPL")) : 1;
  if (want == 4 && a.P % 4 == 0 && al16) {
    const dim3 grid((a.P / 4 + 255) / 256, batch);
    hipLaunchKernelGGL(mask_assemble_kernel<4>, grid, dim3(256), 0, st, a);
  } else if (want >= 2 && a.P % 2 == 0 && al16) {
    const dim3 grid((a.P / 2 + 255) / 256, batch);
    hipLaunchKernelGGL(mask_assemble_kernel<2>, grid, dim3(256), 0, st, a);
  } else {
    const dim3 grid((a.P + 255) / 256, batch);
    hipLaunchKernelGGL(mask_assemble_kernel<1>, grid, dim3(256), 0, st, a);
  }
  return hipGetLastError();
}
