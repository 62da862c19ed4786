// Prior boxes on the device: Anchor._generate_anchors + Anchor.get_anchors (data/anchor.py:23-68),
// the producer of Detect's `priors` input.
//
// One thread per anchor, anchors in the reference's order (level, row j, column i, aspect ratio).
// The reference builds [cx, cy, w, h] from Python floats (float64), converts once to float32
// (anchor.py:52-53) and makes the [ymin, xmin, ymax, xmax] form in float32 (L59-66); the same
// operations, in the same order and precisions, run here.  sqrt(aspect_ratio) is taken on the host
// (three values), so the device only needs IEEE double multiply and divide.
#include "common.hpp"

#define YP_PRIOR_MAX_LEVELS 8
#define YP_PRIOR_MAX_RATIOS 8

struct PriorArgs {
  int num_levels, num_ratios;
  int fh[YP_PRIOR_MAX_LEVELS], fw[YP_PRIOR_MAX_LEVELS];
  long long first[YP_PRIOR_MAX_LEVELS + 1];   // first anchor of every level
  double scale[YP_PRIOR_MAX_LEVELS];
  double sqrt_ar[YP_PRIOR_MAX_RATIOS];
  double img_h, img_w;
  float4* out;
};

__global__ __launch_bounds__(256) void make_priors_kernel(const PriorArgs p) {
  const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= p.first[p.num_levels]) return;
  int l = 0;
  while (a >= p.first[l + 1]) ++l;
  const long long r = a - p.first[l];
  const int k = (int)(r % p.num_ratios);
  const long long cell = r / p.num_ratios;
  const int i = (int)(cell % p.fw[l]), j = (int)(cell / p.fw[l]);
  const double x = ((double)i + 0.5) / (double)p.fw[l] * p.img_w;     // anchor.py:37
  const double y = ((double)j + 0.5) / (double)p.fh[l] * p.img_h;     // anchor.py:38
  const double w = p.scale[l] * p.sqrt_ar[k];                          // anchor.py:41
  const double h = p.scale[l] / p.sqrt_ar[k];                          // anchor.py:42
  const float cx = (float)x, cy = (float)y, fw = (float)w, fh = (float)h;   // anchor.py:52-53
  float4 o;
  o.x = cy - (fh / 2.0f);   // ymin   anchor.py:62
  o.y = cx - (fw / 2.0f);   // xmin
  o.z = cy + (fh / 2.0f);   // ymax
  o.w = cx + (fw / 2.0f);   // xmax
  p.out[a] = o;
}

// Returns the number of anchors, or -1 if the description does not fit the kernel's tables.
long long yp_priors_count(int num_levels, const int32_t* fmap_hw, int num_ratios) {
  if (num_levels < 1 || num_levels > YP_PRIOR_MAX_LEVELS || num_ratios < 1 || num_ratios > YP_PRIOR_MAX_RATIOS) return -1;
  long long n = 0;
  for (int l = 0; l < num_levels; ++l) {
    if (fmap_hw[2 * l] < 1 || fmap_hw[2 * l + 1] < 1) return -1;
    n += (long long)fmap_hw[2 * l] * fmap_hw[2 * l + 1] * num_ratios;
  }
  return n;
}

hipError_t yp_launch_make_priors(int img_h, int img_w, int num_levels, const int32_t* fmap_hw, const double* scales,
                                 int num_ratios, const double* sqrt_ratios, float* priors, hipStream_t st) {
  PriorArgs a;
  a.num_levels = num_levels; a.num_ratios = num_ratios;
  a.first[0] = 0;
  for (int l = 0; l < num_levels; ++l) {
    a.fh[l] = fmap_hw[2 * l]; a.fw[l] = fmap_hw[2 * l + 1];
    a.scale[l] = scales[l];
    a.first[l + 1] = a.first[l] + (long long)a.fh[l] * a.fw[l] * num_ratios;
  }
  for (int l = num_levels; l < YP_PRIOR_MAX_LEVELS; ++l) { a.fh[l] = a.fw[l] = 1; a.scale[l] = 0.0; a.first[l + 1] = a.first[num_levels]; }
  for (int k = 0; k < YP_PRIOR_MAX_RATIOS; ++k) a.sqrt_ar[k] = k < num_ratios ? sqrt_ratios[k] : 1.0;
  a.img_h = (double)img_h; a.img_w = (double)img_w;
  a.out = reinterpret_cast<float4*>(priors);
  const long long n = a.first[num_levels];
  hipLaunchKernelGGL(make_priors_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}
