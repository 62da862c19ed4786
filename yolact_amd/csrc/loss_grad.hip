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
//   grad_coef:  per pixel row y one fma chain over x ascending; the rows' values added up in ascending y,
//               starting from +0.
// A term with dz == 0 leaves a chain as it is (finite inputs), so only pixels inside a row's crop box
// are read: per image ~10 % of the 2 x rows x P mask and gradient values.
//
// grad_proto: a lane owns a pixel (32 accumulators), visits the rows whose box can touch its wave's 64
// pixels, reads mask and gradient inside the boxes once and writes 4PK bytes per image.
// grad_coef: a workgroup owns one pixel row of one image: the row's prototypes go to LDS once and serve
// every detection whose box covers the row (first version: one workgroup per detection re-read the
// image's prototypes from L2 / Infinity Cache for every box - 1.2 GB per call at 13 % box coverage,
// 173 us); the per-row values go to the workspace and a second kernel adds them up in row order.
#include "common.hpp"

#define YP_LG_BLOCK 256
#define YP_GC_THREADS 256                        // grad_coef: 8 half-waves, each one detection at a time
#define YP_GC_HALVES (YP_GC_THREADS / YP_MASK_DIM)

struct LossGradArgs {
  const float* proto;        // [B,P,32]
  const float* coef;         // [B,rows,32]
  const float4* crop;        // [B,rows] x1,x2,y1,y2 in proto pixels (crop_edges_kernel, mask.hip)
  const int32_t* num_rows;   // [B]
  const float* masks;        // [B,rows,P] forward output
  const float* grad;         // [B,rows,P] dL/dmasks
  float* gproto;             // [B,P,32]
  float* gcoef;              // [B,rows,32]
  float* part;               // [B,rows,ph,32] grad_coef per pixel row (written where a box covers the row)
  int rows, P, pw, ph;
};

__device__ __forceinline__ float dz_of(float g, float m) { return (g * m) * (1.0f - m); }

// grad_proto: a lane owns a pixel and its 32 accumulators; rows whose box misses the whole wave are skipped.
// The image's coefficient rows and crop edges are staged in LDS by the workgroup (dynamic: rows x 36 floats)
// and read from there as broadcasts: fetched per visited row with scalar loads, every row cost the wave a
// dependent scalar-memory round trip before its 32 FMAs could start (67 % of the wave cycles were waits).
__global__ __launch_bounds__(YP_LG_BLOCK) void loss_grad_proto_kernel(const LossGradArgs a, int bpi) {
  extern __shared__ __align__(16) float lgp_lds[];
  float4* const s_crop = reinterpret_cast<float4*>(lgp_lds);              // [rows]
  float* const s_coef = lgp_lds + (size_t)a.rows * 4;                     // [rows][32]
  const int b = (int)blockIdx.x / bpi, blk = (int)blockIdx.x - b * bpi;
  const int p = blk * YP_LG_BLOCK + (int)threadIdx.x;
  const bool live = p < a.P;
  const int pc = live ? p : a.P - 1;
  const int row = pc / a.pw, col = pc - row * a.pw;
  const float frow = (float)row, fcol = (float)col;
  const int N = min(a.num_rows[b], a.rows);
  {
    const float4* gc = a.crop + (size_t)b * a.rows;
    const float4* gf = reinterpret_cast<const float4*>(a.coef + (size_t)b * a.rows * YP_MASK_DIM);
    float4* sf = reinterpret_cast<float4*>(s_coef);
    for (int i = (int)threadIdx.x; i < N; i += YP_LG_BLOCK) s_crop[i] = gc[i];
    for (int i = (int)threadIdx.x; i < N * (YP_MASK_DIM / 4); i += YP_LG_BLOCK) sf[i] = gf[i];
  }
  __syncthreads();
  const float* mk = a.masks + (size_t)b * a.rows * a.P + pc;
  const float* gr = a.grad + (size_t)b * a.rows * a.P + pc;
  float acc[YP_MASK_DIM];
#pragma unroll
  for (int k = 0; k < YP_MASK_DIM; ++k) acc[k] = 0.0f;
  // the wave's 64 consecutive pixels lie in rows ra..rb; over one row, columns ca..cb2, else all of them
  const int lane = (int)threadIdx.x & 63;
  const int pfirst = min(p - lane, a.P - 1), plast = min(p - lane + 63, a.P - 1);
  const int ra = pfirst / a.pw, rb = plast / a.pw;
  const float fra = (float)ra, frb = (float)rb;
  const float fca = ra == rb ? (float)(pfirst - ra * a.pw) : 0.0f, fcb = ra == rb ? (float)(plast - rb * a.pw) : (float)(a.pw - 1);
  for (int n0 = 0; n0 < N; n0 += 64) {
    const float4 ce = s_crop[min(n0 + lane, N - 1)];
    const bool touch = (n0 + lane < N) && (fcb >= ce.x) && (fca < ce.y) && (frb >= ce.z) && (fra < ce.w);
    unsigned long long rel = __ballot(touch);
    while (rel) {                      // wave-uniform: ascending n, as the chain's order demands
      const int n = n0 + __builtin_ctzll(rel);
      rel &= rel - 1ull;
      const float4 cb = s_crop[n];
      const bool in = live && (fcol >= cb.x) && (fcol < cb.y) && (frow >= cb.z) && (frow < cb.w);   // utils/utils.py:119-126
      float dz = 0.0f;
      if (in) dz = dz_of(gr[(size_t)n * a.P], mk[(size_t)n * a.P]);
      const float4* cf = reinterpret_cast<const float4*>(s_coef + (size_t)n * YP_MASK_DIM);   // one address for the wave: broadcasts
#pragma unroll
      for (int k = 0; k < YP_MASK_DIM; k += 4) {
        const float4 c4 = cf[k / 4];
        acc[k] = __builtin_fmaf(dz, c4.x, acc[k]);
        acc[k + 1] = __builtin_fmaf(dz, c4.y, acc[k + 1]);
        acc[k + 2] = __builtin_fmaf(dz, c4.z, acc[k + 2]);
        acc[k + 3] = __builtin_fmaf(dz, c4.w, acc[k + 3]);
      }
    }
  }
  if (live) {
    float4* out = reinterpret_cast<float4*>(a.gproto + ((size_t)b * a.P + p) * YP_MASK_DIM);
#pragma unroll
    for (int k = 0; k < YP_MASK_DIM; k += 4) out[k / 4] = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
  }
}

// The integer superset of a crop box's pixels that both grad_coef kernels use (pixels outside the box have
// m == 0 and add nothing; NaN edges compare false everywhere in the forward pass, so any range is right).
__device__ __forceinline__ void box_range(const float4 cb, int pw, int ph, int& x0, int& x1, int& y0, int& y1) {
  x0 = (int)fminf(fmaxf(floorf(cb.x), 0.0f), (float)pw); x1 = (int)fminf(fmaxf(ceilf(cb.y), 0.0f), (float)pw);
  y0 = (int)fminf(fmaxf(floorf(cb.z), 0.0f), (float)ph); y1 = (int)fminf(fmaxf(ceilf(cb.w), 0.0f), (float)ph);
}

// grad_coef, step 1: workgroup (b, y) - the prototypes of pixel row y in LDS; a half-wave (lane = coefficient
// k) takes one detection whose box covers the row at a time: dz of the row's pixels into LDS (x-parallel),
// then the chain over x.  Dynamic LDS: prow[pw][32] | dzs[YP_GC_HALVES][pwp] | rel[256] ints.
__global__ __launch_bounds__(YP_GC_THREADS) void loss_grad_coef_rows_kernel(const LossGradArgs a, int pwp) {
  extern __shared__ __align__(16) float lg_lds[];
  float* const prow = lg_lds;
  float* const dzs = prow + (size_t)a.pw * YP_MASK_DIM;
  int* const rel = reinterpret_cast<int*>(dzs + YP_GC_HALVES * pwp);
  __shared__ int nrel;
  const int b = (int)blockIdx.x / a.ph, y = (int)blockIdx.x - b * a.ph;
  const int t = (int)threadIdx.x, k = t & (YP_MASK_DIM - 1), h = t / YP_MASK_DIM;
  const int N = min(a.num_rows[b], a.rows);
  if (t == 0) nrel = 0;
  {
    const float4* src = reinterpret_cast<const float4*>(a.proto + ((size_t)b * a.P + (size_t)y * a.pw) * YP_MASK_DIM);
    float4* dst = reinterpret_cast<float4*>(prow);
    for (int i = t; i < a.pw * (YP_MASK_DIM / 4); i += YP_GC_THREADS) dst[i] = src[i];
  }
  __syncthreads();
  for (int n = t; n < N; n += YP_GC_THREADS) {   // (rows <= 256: one trip; the list's order does not matter)
    int x0, x1, y0, y1;
    box_range(a.crop[(size_t)b * a.rows + n], a.pw, a.ph, x0, x1, y0, y1);
    if (y >= y0 && y < y1 && x0 < x1) rel[atomicAdd(&nrel, 1)] = n;
  }
  __syncthreads();
  const int nr = nrel;
  float* const dzh = dzs + h * pwp;
  for (int i = h; i < nr; i += YP_GC_HALVES) {   // (uniform per half-wave; the two halves of a wave may differ)
    const int n = rel[i];
    int x0, x1, y0, y1;
    box_range(a.crop[(size_t)b * a.rows + n], a.pw, a.ph, x0, x1, y0, y1);
    const float* const mk = a.masks + ((size_t)b * a.rows + n) * a.P + (size_t)y * a.pw;
    const float* const gr = a.grad + ((size_t)b * a.rows + n) * a.P + (size_t)y * a.pw;
    for (int x = x0 + k; x < x1; x += YP_MASK_DIM) dzh[x - x0] = dz_of(gr[x], mk[x]);
    __builtin_amdgcn_wave_barrier();           // the half-wave's LDS stores precede its loads (one wave: in order)
    float acc = 0.0f;
    const int w = x1 - x0;
    int j = 0;
    for (; j + 8 <= w; j += 8) {
      float dz[8], pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { dz[u] = dzh[j + u]; pv[u] = prow[(x0 + j + u) * YP_MASK_DIM + k]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_fmaf(dz[u], pv[u], acc);
    }
    for (; j < w; ++j) acc = __builtin_fmaf(dzh[j], prow[(x0 + j) * YP_MASK_DIM + k], acc);
    a.part[(((size_t)b * a.rows + n) * a.ph + y) * YP_MASK_DIM + k] = acc;
    __builtin_amdgcn_wave_barrier();           // dzh is rewritten by the next detection
  }
}

#ifdef YP_EXP_GC_MFMA
// EXPERIMENT BUILD (-DYP_EXP_GC_MFMA, never the shipped library): step 1 on the matrix cores.  The one GEMM-shaped
// piece of this path is [detections x pixels of a row] x [pixels x 32 coefficients]; here a wave takes 32 of the
// row's detections at a time, lays their dz out in LDS in chunks of 64 pixels (zeros outside each detection's
// x-range) and accumulates D[32 det][32 k] with v_mfma_f32_32x32x2_f32 (K = 2 pixels per instruction).  The sum over
// x is then in the matrix core's order, not the fma chain's: results agree with the chain to rounding, not to the bit
// (tools/gc_mfma_check.py reports the distance).  MEASURED (config 2's shape, 3200 rows, boxes of 10-60 % of the
// image side, rocprofv3 --kernel-trace, profiles/r03/gc_mfma_experiment.md): 177.5 us against the chain kernel's
// 72.5 us (first version, one load round trip per detection: 276 us).  SQ_VALU_MFMA_BUSY_CYCLES = 30.8 M = 481 k
// instructions x 64 cycles, 6.8 % of the launch's SIMD cycles: the matrix cores are not the limit, the dense A
// operand is - a row's 32 detections cover ~36 % of the chunk columns each, so two thirds of the LDS fill and of the
// products are zeros, while the chain visits only in-box pixels.  The results came out bit-identical to the chain's
// (the f32 MFMA accumulated its K steps as fused multiply-adds in k order on this hardware; observed, not specified).
// The chain kernel stays the shipped one.
typedef float yp_f32x16 __attribute__((ext_vector_type(16)));
#define YP_GM_CHUNK 64
#define YP_GM_STRIDE (YP_GM_CHUNK + 1)          // odd: the 32 detections of an A operand read hit 32 banks
__global__ __launch_bounds__(YP_GC_THREADS) void loss_grad_coef_rows_mfma_kernel(const LossGradArgs a, int pwc) {
  extern __shared__ __align__(16) float lg_lds[];
  float* const prow = lg_lds;                                            // [pwc][32], rows >= pw are zeros
  float* const dzs = prow + (size_t)pwc * YP_MASK_DIM;                   // [waves][32][YP_GM_STRIDE]
  int* const rel = reinterpret_cast<int*>(dzs + (YP_GC_THREADS / 64) * 32 * YP_GM_STRIDE);   // [256] detection
  int* const relx = rel + 256;                                           // [256] x0 | x1 << 16
  __shared__ int nrel;
  const int b = (int)blockIdx.x / a.ph, y = (int)blockIdx.x - b * a.ph;
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const int N = min(a.num_rows[b], a.rows);
  if (t == 0) nrel = 0;
  {
    const float4* src = reinterpret_cast<const float4*>(a.proto + ((size_t)b * a.P + (size_t)y * a.pw) * YP_MASK_DIM);
    float4* dst = reinterpret_cast<float4*>(prow);
    const int live = a.pw * (YP_MASK_DIM / 4), all = pwc * (YP_MASK_DIM / 4);
    for (int i = t; i < all; i += YP_GC_THREADS) dst[i] = i < live ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  for (int n = t; n < N; n += YP_GC_THREADS) {
    int x0, x1, y0, y1;
    box_range(a.crop[(size_t)b * a.rows + n], a.pw, a.ph, x0, x1, y0, y1);
    if (y >= y0 && y < y1 && x0 < x1) { const int s = atomicAdd(&nrel, 1); rel[s] = n; relx[s] = x0 | (x1 << 16); }
  }
  __syncthreads();
  const int nr = nrel;
  float* const dzw = dzs + wave * 32 * YP_GM_STRIDE;
  const int li = lane & 31, lh = lane >> 5;
  for (int g0 = wave * 32; g0 < nr; g0 += (YP_GC_THREADS / 64) * 32) {   // (uniform per wave)
    const int cnt = min(32, nr - g0);
    int xa = a.pw, xb = 0;
    for (int i = 0; i < cnt; ++i) { const int xx = relx[g0 + i]; xa = min(xa, xx & 0xffff); xb = max(xb, xx >> 16); }
    yp_f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
    for (int c0 = xa & ~(YP_GM_CHUNK - 1); c0 < xb; c0 += YP_GM_CHUNK) {
      const int x = c0 + lane;
      for (int i0 = 0; i0 < 32; i0 += 8) {       // eight detections' loads in flight: clamped addresses, no branches
        float gv[8], mv[8];
        bool ok[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int s = g0 + min(i0 + u, cnt - 1);
          const int xx = relx[s], x0 = xx & 0xffff, x1 = xx >> 16;
          const size_t o = ((size_t)b * a.rows + rel[s]) * a.P + (size_t)y * a.pw + min(max(x, x0), x1 - 1);
          gv[u] = a.grad[o]; mv[u] = a.masks[o];
          ok[u] = (i0 + u < cnt) && x >= x0 && x < x1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) dzw[(i0 + u) * YP_GM_STRIDE + lane] = ok[u] ? dz_of(gv[u], mv[u]) : 0.0f;
      }
      __builtin_amdgcn_wave_barrier();
      const int ce = min(YP_GM_CHUNK, xb - c0);
      for (int j = 0; j < ce; j += 2) {
        const float av = dzw[li * YP_GM_STRIDE + j + lh];
        const float bv = prow[(size_t)(c0 + j + lh) * YP_MASK_DIM + li];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int i = 8 * (v >> 2) + lh * 4 + (v & 3);
      if (i < cnt) a.part[(((size_t)b * a.rows + rel[g0 + i]) * a.ph + y) * YP_MASK_DIM + li] = acc[v];
    }
  }
}
#endif

// grad_coef, step 2: the rows' values of a detection added up in ascending y, from +0 (a half-wave per detection).
__global__ __launch_bounds__(YP_GC_THREADS) void loss_grad_coef_sum_kernel(const LossGradArgs a, int total) {
  const int t = (int)threadIdx.x, k = t & (YP_MASK_DIM - 1), h = t / YP_MASK_DIM;
  const int i = (int)blockIdx.x * YP_GC_HALVES + h;   // b * rows + n
  if (i >= total) return;
  const int b = i / a.rows, n = i - b * a.rows;
  float tot = 0.0f;
  if (n < a.num_rows[b]) {
    int x0, x1, y0, y1;
    box_range(a.crop[i], a.pw, a.ph, x0, x1, y0, y1);
    if (x0 < x1) {
      const float* pt = a.part + (size_t)i * a.ph * YP_MASK_DIM + k;
      int y = y0;
      for (; y + 8 <= y1; y += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pt[(size_t)(y + u) * YP_MASK_DIM];
#pragma unroll
        for (int u = 0; u < 8; ++u) tot = tot + v[u];
      }
      for (; y < y1; ++y) tot = tot + pt[(size_t)y * YP_MASK_DIM];
    }
  }
  a.gcoef[(size_t)i * YP_MASK_DIM + k] = tot;
}

size_t yp_loss_grads_workspace_floats(int batch, int rows, int ph) { return (size_t)batch * rows * (4 + (size_t)ph * YP_MASK_DIM); }

hipError_t yp_launch_loss_grads(int batch, int rows, int ph, int pw, int img_h, int img_w, int use_crop, const float* proto,
                                const float* coef, const float* boxes, const int32_t* num_rows, const float* masks,
                                const float* grad_masks, float* workspace, float* grad_proto, float* grad_coef, hipStream_t st) {
  float* const crop_ws = workspace;                        // [B,rows,4]
  float* const part = workspace + (size_t)batch * rows * 4;   // [B,rows,ph,32]
  hipError_t e = yp_launch_crop_edges(batch * rows, ph, pw, img_h, img_w, use_crop, boxes, crop_ws, st);
  if (e != hipSuccess) return e;
  LossGradArgs a;
  a.proto = proto; a.coef = coef; a.crop = reinterpret_cast<const float4*>(crop_ws); a.num_rows = num_rows;
  a.masks = masks; a.grad = grad_masks; a.gproto = grad_proto; a.gcoef = grad_coef; a.part = part;
  a.rows = rows; a.P = ph * pw; a.pw = pw; a.ph = ph;
  const int bpi = (a.P + YP_LG_BLOCK - 1) / YP_LG_BLOCK;
  hipLaunchKernelGGL(loss_grad_proto_kernel, dim3(batch * bpi), dim3(YP_LG_BLOCK), (size_t)rows * (4 + YP_MASK_DIM) * sizeof(float), st, a, bpi);
  const int pwp = (pw + 3) / 4 * 4;
  const size_t lds = ((size_t)pw * YP_MASK_DIM + (size_t)YP_GC_HALVES * pwp) * sizeof(float) + 256 * sizeof(int);
  if (lds > 48 * 1024) {   // (proto rows wider than ~350 pixels)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_grad_coef_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
#ifdef YP_EXP_GC_MFMA
  {
    const int pwc = (pw + YP_GM_CHUNK - 1) / YP_GM_CHUNK * YP_GM_CHUNK;     // the chunks' last pixels read zero rows
    const size_t ml = ((size_t)pwc * YP_MASK_DIM + (size_t)(YP_GC_THREADS / 64) * 32 * YP_GM_STRIDE) * sizeof(float) + 512 * sizeof(int);
    if (ml > 64 * 1024) return hipErrorInvalidValue;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&loss_grad_coef_rows_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ml);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(loss_grad_coef_rows_mfma_kernel, dim3(batch * ph), dim3(YP_GC_THREADS), ml, st, a, pwc);
  }
#else
  hipLaunchKernelGGL(loss_grad_coef_rows_kernel, dim3(batch * ph), dim3(YP_GC_THREADS), lds, st, a, pwp);
#endif
  const int total = batch * rows;
  hipLaunchKernelGGL(loss_grad_coef_sum_kernel, dim3((total + YP_GC_HALVES - 1) / YP_GC_HALVES), dim3(YP_GC_THREADS), 0, st, a, total);
  return hipGetLastError();
}
