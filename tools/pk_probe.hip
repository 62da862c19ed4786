// Probe: is v_pk_fma_f32 worth it on gfx950?  Same flops as scalar-form v_fma_f32 chains, half the
// instructions.  Single-wave workgroups like score_filter's, at 1 and 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/pk_probe.hip -o tools/pk_probe && tools/pk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096
__global__ __launch_bounds__(64) void k_fma(float* out, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
__global__ __launch_bounds__(64) void k_pk(float* out, float a, float b) {
  f2 x[4]; f2 av = {a, a}, bv = {b, b};
  for (int i = 0; i < 4; ++i) { x[i].x = threadIdx.x * 1e-3f + i; x[i].y = x[i].x + 4; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
  }
  float s = 0; for (int i = 0; i < 4; ++i) s += x[i].x + x[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 256 * 32 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpc : {4, 12, 32}) {
    for (int which = 0; which < 2; ++which) {
      float best = 1e9;
      for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_fma, dim3(256 * wpc), dim3(64), 0, 0, out, 0.999f, 0.001f);
        else hipLaunchKernelGGL(k_pk, dim3(256 * wpc), dim3(64), 0, 0, out, 0.999f, 0.001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      const double flops = 256.0 * wpc * 64 * ITERS * 8 * 2;
      printf("%s waves/CU=%2d  %.3f ms  %.1f TFLOP/s\n", which ? "v_pk_fma_f32" : "v_fma_f32   ", wpc, best, flops / best / 1e9);
    }
  }
  return 0;
}
