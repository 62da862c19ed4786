// Probe: what do score_filter's per-key list appends cost if the lists are shared per (image, class group)
// and the slot comes from an agent-scope atomic?  3072 single-wave workgroups (the score_filter grid at
// config 2), each doing `flushes` appends of 64 keys spread over 16 group counters of its image.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_probe.hip -o tools/atomic_probe && tools/atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;

// MODE 0: one atomic per lane.  MODE 1: one atomic per (wave, group present), issued back to back, one wait.
template <int MODE>
__global__ __launch_bounds__(64) void append_kernel(int* cnt, u64* lists, int cap, int flushes, int items_per_image, int ngroups, int stride) {
  const int lane = threadIdx.x;
  const int img = blockIdx.x / items_per_image;
  int* const c = cnt + (size_t)img * ngroups * stride;
  for (int it = 0; it < flushes; ++it) {
    const int g = (lane * 7 + it * 3 + blockIdx.x) % ngroups;
    int slot;
    if (MODE == 0) {
      slot = __hip_atomic_fetch_add(&c[g * stride], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // leaders: the lowest lane of every group present
      u64 rest = __ballot(1);
      int base = 0, leader = 0, rank = 0;
      while (rest) {
        const int l0 = __ffsll((long long)rest) - 1;
        const int g0 = __builtin_amdgcn_readlane(g, l0);
        const u64 m = __ballot(g == g0) & rest;
        if (g == g0) { leader = l0; rank = __popcll(m & ((1ull << lane) - 1ull)); }
        if (lane == l0) base = __hip_atomic_fetch_add(&c[g0 * stride], __popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        rest &= ~m;
      }
      slot = __shfl(base, leader) + rank;
    }
    lists[((size_t)(img * ngroups + g)) * cap + (slot & (cap - 1))] = ((u64)slot << 32) | (unsigned)blockIdx.x;
  }
}

int main() {
  const int B = 32, ipi = 96, cap = 8192;
  int* cnt; u64* lists;
  (void)hipMalloc(&cnt, (size_t)B * 16 * 4096 * sizeof(int));
  (void)hipMalloc(&lists, (size_t)B * 16 * cap * sizeof(u64));
  (void)hipMemset(cnt, 0, (size_t)B * 16 * 4096 * sizeof(int));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, int flushes, int ng, int stride) {
    for (int i = 0; i < 3; ++i) launch(flushes, ng, stride);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch(flushes, ng, stride);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-20s flushes %2d groups %2d stride %5d ints: %7.1f us per launch (%d keys)\n", name, flushes, ng, stride, ms / 20 * 1e3, B * ipi * flushes * 64);
  };
  for (int stride : {1, 32, 64, 1024, 4096})
    for (int ng : {16, 1})
      for (int f : {1, 4}) {
        if (stride == 1 || ng == 16) run("wave-aggregated", [&](int fl, int n, int st) { hipLaunchKernelGGL((append_kernel<1>), dim3(B * ipi), dim3(64), 0, 0, cnt, lists, cap, fl, ipi, n, st); }, f, ng, stride);
      }
  run("per-lane", [&](int fl, int n, int st) { hipLaunchKernelGGL((append_kernel<0>), dim3(B * ipi), dim3(64), 0, 0, cnt, lists, cap, fl, ipi, n, st); }, 1, 16, 32);
  run("empty-ish (0 flushes)", [&](int fl, int n, int st) { hipLaunchKernelGGL((append_kernel<1>), dim3(B * ipi), dim3(64), 0, 0, cnt, lists, cap, fl, ipi, n, st); }, 0, 16, 32);
  return 0;
}
