// Gradients of the training loss's mask assembly (backward).
//
// The reference has no code for this step: TensorFlow's autodiff differentiates
// loss/loss_yolact.py:239-253 (matmul, tf.sigmoid, utils.crop).  With m = the forward's cropped output
// and g = dL/dm:   dz = (g * m) * (1 - m)      (SigmoidGrad; 0 outside the box because m is 0 there)
//   grad_proto[b,p,k] = sum_n dz[n,p] * coef[b,n,k]        (matmul: dA = dC B)
//   grad_coef[b,n,k]  = sum_p dz[n,p] * proto[b,p,k]       (matmul: dB = A^T dC)
// TF fixes no summation order.  These kernels do - the tests' C restatement follows the same one, so
// that the comparison is bit-exact - and a result is reproducible run to run (no float atomics):
//   grad_proto: one fma chain over n ascending;
//   grad_coef:  YP_GC_STREAMS fma chains, stream s over the pixel rows y % YP_GC_STREAMS == s in (y, x)
//               order, added up in stream order.
// A term with dz == 0 leaves a chain as it is (finite inputs), so only pixels inside a row's crop box
// are read: per image ~10 % of the 2 x rows x P mask and gradient values.
//
// Both kernels are bound by cache traffic and latency, not by HBM or the FMA rate: grad_proto reads
// the masks and gradients inside the boxes once and writes 4PK bytes per image; grad_coef re-reads
// the image's prototypes (4PK bytes, L2 / Infinity Cache) for every row's box.
#include "common.hpp"

#define YP_GC_STREAMS 16
#define YP_GC_THREADS (YP_GC_STREAMS * YP_MASK_DIM)
#define YP_LG_BLOCK 256

struct LossGradArgs {
  const float* proto;        // [B,P,32]
  const float* coef;         // [B,rows,32]
  const float4* crop;        // [B,rows] x1,x2,y1,y2 in proto pixels (crop_edges_kernel, mask.hip)
  const int32_t* num_rows;   // [B]
  const float* masks;        // [B,rows,P] forward output
  const float* grad;         // [B,rows,P] dL/dmasks
  float* gproto;             // [B,P,32]
  float* gcoef;              // [B,rows,32]
  int rows, P, pw, ph;
};

__device__ __forceinline__ float dz_of(float g, float m) { return (g * m) * (1.0f - m); }

// grad_proto: a lane owns a pixel and its 32 accumulators; rows whose box misses the whole wave are skipped.
__global__ __launch_bounds__(YP_LG_BLOCK) void loss_grad_proto_kernel(const LossGradArgs a, int bpi) {
  const int b = (int)blockIdx.x / bpi, blk = (int)blockIdx.x - b * bpi;
  const int p = blk * YP_LG_BLOCK + (int)threadIdx.x;
  const bool live = p < a.P;
  const int pc = live ? p : a.P - 1;
  const int row = pc / a.pw, col = pc - row * a.pw;
  const float frow = (float)row, fcol = (float)col;
  const int N = min(a.num_rows[b], a.rows);
  const float4* crop = a.crop + (size_t)b * a.rows;
  const float* coef = a.coef + (size_t)b * a.rows * YP_MASK_DIM;
  const float* mk = a.masks + (size_t)b * a.rows * a.P + pc;
  const float* gr = a.grad + (size_t)b * a.rows * a.P + pc;
  float acc[YP_MASK_DIM];
#pragma unroll
  for (int k = 0; k < YP_MASK_DIM; ++k) acc[k] = 0.0f;
  for (int n = 0; n < N; ++n) {
    const float4 cb = crop[n];
    const bool in = live && (fcol >= cb.x) && (fcol < cb.y) && (frow >= cb.z) && (frow < cb.w);   // utils/utils.py:119-126
    if (__ballot(in) == 0) continue;   // wave-uniform
    float dz = 0.0f;
    if (in) dz = dz_of(gr[(size_t)n * a.P], mk[(size_t)n * a.P]);
    const float* cf = coef + (size_t)n * YP_MASK_DIM;   // wave-uniform address
#pragma unroll
    for (int k = 0; k < YP_MASK_DIM; ++k) acc[k] = __builtin_fmaf(dz, cf[k], acc[k]);
  }
  if (live) {
    float4* out = reinterpret_cast<float4*>(a.gproto + ((size_t)b * a.P + p) * YP_MASK_DIM);
#pragma unroll
    for (int k = 0; k < YP_MASK_DIM; k += 4) out[k / 4] = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
  }
}

// grad_coef: one workgroup per (image, row); 32 lanes (one per coefficient) form a stream.
__global__ __launch_bounds__(YP_GC_THREADS) void loss_grad_coef_kernel(const LossGradArgs a) {
  __shared__ float part[YP_GC_STREAMS][YP_MASK_DIM];
  const int b = (int)blockIdx.x / a.rows, n = (int)blockIdx.x - b * a.rows;
  const int t = (int)threadIdx.x, k = t & (YP_MASK_DIM - 1), s = t / YP_MASK_DIM;
  float* const out = a.gcoef + ((size_t)b * a.rows + n) * YP_MASK_DIM;
  if (n >= a.num_rows[b]) {   // workgroup-uniform
    if (t < YP_MASK_DIM) out[t] = 0.0f;
    return;
  }
  const float4 cb = a.crop[(size_t)b * a.rows + n];
  // an integer superset of the box (pixels outside it have m == 0 and add nothing); NaN edges compare
  // false everywhere in the forward pass, so any range is right for them
  const int x0 = (int)fminf(fmaxf(floorf(cb.x), 0.0f), (float)a.pw), x1 = (int)fminf(fmaxf(ceilf(cb.y), 0.0f), (float)a.pw);
  const int y0 = (int)fminf(fmaxf(floorf(cb.z), 0.0f), (float)a.ph), y1 = (int)fminf(fmaxf(ceilf(cb.w), 0.0f), (float)a.ph);
  const float* const mk = a.masks + ((size_t)b * a.rows + n) * a.P;
  const float* const gr = a.grad + ((size_t)b * a.rows + n) * a.P;
  const float* const pr = a.proto + (size_t)b * a.P * YP_MASK_DIM + k;
  float acc = 0.0f;
  for (int y = y0 + ((s - y0) % YP_GC_STREAMS + YP_GC_STREAMS) % YP_GC_STREAMS; y < y1; y += YP_GC_STREAMS) {
    const int base = y * a.pw;
    int x = x0;
    for (; x + 8 <= x1; x += 8) {          // eight pixels' loads in flight, the chain in order
      float m[8], g[8], pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        m[u] = mk[base + x + u];
        g[u] = gr[base + x + u];
        pv[u] = pr[(size_t)(base + x + u) * YP_MASK_DIM];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_fmaf(dz_of(g[u], m[u]), pv[u], acc);
    }
    for (; x < x1; ++x) acc = __builtin_fmaf(dz_of(gr[base + x], mk[base + x]), pr[(size_t)(base + x) * YP_MASK_DIM], acc);
  }
  part[s][k] = acc;
  __syncthreads();
  if (t < YP_MASK_DIM) {
    float total = part[0][t];
#pragma unroll
    for (int q = 1; q < YP_GC_STREAMS; ++q) total = total + part[q][t];
    out[t] = total;
  }
}

hipError_t yp_launch_loss_grads(int batch, int rows, int ph, int pw, int img_h, int img_w, int use_crop, const float* proto,
                                const float* coef, const float* boxes, const int32_t* num_rows, const float* masks,
                                const float* grad_masks, float* crop_ws, float* grad_proto, float* grad_coef, hipStream_t st) {
  hipError_t e = yp_launch_crop_edges(batch * rows, ph, pw, img_h, img_w, use_crop, boxes, crop_ws, st);
  if (e != hipSuccess) return e;
  LossGradArgs a;
  a.proto = proto; a.coef = coef; a.crop = reinterpret_cast<const float4*>(crop_ws); a.num_rows = num_rows;
  a.masks = masks; a.grad = grad_masks; a.gproto = grad_proto; a.gcoef = grad_coef;
  a.rows = rows; a.P = ph * pw; a.pw = pw; a.ph = ph;
  const int bpi = (a.P + YP_LG_BLOCK - 1) / YP_LG_BLOCK;
  hipLaunchKernelGGL(loss_grad_proto_kernel, dim3(batch * bpi), dim3(YP_LG_BLOCK), 0, st, a, bpi);
  hipLaunchKernelGGL(loss_grad_coef_kernel, dim3(batch * rows), dim3(YP_GC_THREADS), 0, st, a);
  return hipGetLastError();
}
