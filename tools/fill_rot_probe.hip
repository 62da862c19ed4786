// Probe: how fast can 244 MB of zeros be written when three output sets rotate (nothing stays in the
// 256 MiB Infinity Cache)?  Variants of the fill's shape.
//   hipcc --offload-arch=gfx950 -O3 tools/fill_rot_probe.hip -o tools/fill_rot_probe && tools/fill_rot_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float nat4 __attribute__((ext_vector_type(4)));
template <int THREADS, int UNROLL, bool NT>
__global__ __launch_bounds__(THREADS) void fill_chunks(char* base, long long len) {
  const long long lo = (long long)blockIdx.x * THREADS * UNROLL * 16;
  const nat4 z = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    const long long o = lo + ((long long)u * THREADS + threadIdx.x) * 16;
    if (o < len) { if (NT) __builtin_nontemporal_store(z, (nat4*)(base + o)); else *(nat4*)(base + o) = z; }
  }
}
template <bool NT>
__global__ __launch_bounds__(256) void fill_stride(char* base, long long len) {
  const nat4 z = {0, 0, 0, 0};
  for (long long o = ((long long)blockIdx.x * 256 + threadIdx.x) * 16; o < len; o += (long long)gridDim.x * 256 * 16) {
    if (NT) __builtin_nontemporal_store(z, (nat4*)(base + o)); else *(nat4*)(base + o) = z;
  }
}
int main() {
  const long long len = 32ll * 100 * 19044 * 4;   // config 2's masks
  char* buf[3];
  for (int i = 0; i < 3; ++i) (void)hipMalloc(&buf[i], len + 256);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int shift = 0; shift <= 64; shift += 64) {
      for (int i = 0; i < 6; ++i) launch(buf[i % 3] + shift);
      (void)hipEventRecord(e0);
      for (int i = 0; i < 30; ++i) launch(buf[i % 3] + shift);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%-44s base+%2d  %6.1f us  %5.2f TB/s\n", name, shift, ms / 30 * 1e3, len / (ms / 30 * 1e-3) / 1e12);
    }
  };
  run("chunks 256thr x4 (16 KiB per WG)", [&](char* b) { hipLaunchKernelGGL((fill_chunks<256, 4, false>), dim3((len + 16383) / 16384), dim3(256), 0, 0, b, len); });
  run("chunks 512thr x16 (128 KiB per WG)", [&](char* b) { hipLaunchKernelGGL((fill_chunks<512, 16, false>), dim3((len + 131071) / 131072), dim3(512), 0, 0, b, len); });
  run("chunks 1024thr x16 (256 KiB per WG)", [&](char* b) { hipLaunchKernelGGL((fill_chunks<1024, 16, false>), dim3((len + 262143) / 262144), dim3(1024), 0, 0, b, len); });
  run("chunks 256thr x4 nt", [&](char* b) { hipLaunchKernelGGL((fill_chunks<256, 4, true>), dim3((len + 16383) / 16384), dim3(256), 0, 0, b, len); });
  run("chunks 512thr x16 nt", [&](char* b) { hipLaunchKernelGGL((fill_chunks<512, 16, true>), dim3((len + 131071) / 131072), dim3(512), 0, 0, b, len); });
  run("grid-stride 2048 x 256", [&](char* b) { hipLaunchKernelGGL((fill_stride<false>), dim3(2048), dim3(256), 0, 0, b, len); });
  run("grid-stride 2048 x 256 nt", [&](char* b) { hipLaunchKernelGGL((fill_stride<true>), dim3(2048), dim3(256), 0, 0, b, len); });
  run("grid-stride 8192 x 256", [&](char* b) { hipLaunchKernelGGL((fill_stride<false>), dim3(8192), dim3(256), 0, 0, b, len); });
  run("hipMemsetAsync", [&](char* b) { (void)hipMemsetAsync(b, 0, len, 0); });
  return 0;
}
