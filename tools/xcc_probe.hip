// Probe: (1) is HW_REG_XCC_ID readable and does workgroup id % 8 predict it?  (2) what do list appends cost when the
// slot counters are per XCD and the atomics stay in that XCD's L2 (no sc1), against agent-scope atomics?
//   hipcc --offload-arch=gfx950 -O3 tools/xcc_probe.hip -o tools/xcc_probe && tools/xcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned long long u64;

__device__ __forceinline__ int xcc_id() {
  int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
__device__ __forceinline__ int atomic_add_l2(int* p, int v) {   // executed in this XCD's L2 (no sc1)
  int old;
  asm volatile("global_atomic_add_u32 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(old) : "v"(p), "v"(v) : "memory");
  return old;
}

__global__ __launch_bounds__(64) void ids_kernel(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// MODE 0: agent-scope atomics on counters [img][16]; MODE 1: L2 atomics on counters [img][xcc][16]
template <int MODE>
__global__ __launch_bounds__(64) void append_kernel(int* cnt, u64* lists, int cap, int flushes, int items_per_image, int stride) {
  const int lane = threadIdx.x;
  const int img = blockIdx.x / items_per_image;
  const int x = MODE == 1 ? xcc_id() & 7 : 0;
  int* const c = cnt + ((size_t)img * 8 + x) * 16 * stride;
  for (int it = 0; it < flushes; ++it) {
    const int g = (lane * 7 + it * 3 + blockIdx.x) % 16;
    u64 rest = __ballot(1);
    int base = 0, leader = 0, rank = 0, share = 0;
    while (rest) {
      const int l0 = __ffsll((long long)rest) - 1;
      const int g0 = __builtin_amdgcn_readlane(g, l0);
      const u64 m = __ballot(g == g0) & rest;
      if (g == g0) { leader = l0; rank = __popcll(m & ((1ull << lane) - 1ull)); }
      if (lane == l0) share = __popcll(m);
      rest &= ~m;
    }
    if (share) base = MODE == 1 ? atomic_add_l2(&c[g * stride], share) : __hip_atomic_fetch_add(&c[g * stride], share, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int slot = __shfl(base, leader) + rank;
    lists[(((size_t)img * 8 + x) * 16 + g) * cap + (slot & (cap - 1))] = ((u64)slot << 32) | (unsigned)blockIdx.x;
  }
}

int main() {
  const int B = 32, ipi = 96, cap = 1024, stride = 32;
  int *cnt, *ids; u64* lists;
  (void)hipMalloc(&cnt, (size_t)B * 8 * 16 * stride * sizeof(int));
  (void)hipMalloc(&ids, 4096 * sizeof(int));
  (void)hipMalloc(&lists, (size_t)B * 8 * 16 * cap * sizeof(u64));
  hipLaunchKernelGGL(ids_kernel, dim3(4096), dim3(64), 0, 0, ids);
  int h[4096];
  (void)hipMemcpy(h, ids, sizeof(h), hipMemcpyDeviceToHost);
  int match = 0, hist[16] = {0};
  for (int i = 0; i < 4096; ++i) { match += h[i] == (i & 7); hist[h[i] & 15]++; }
  printf("XCC ids of 4096 workgroups: id == wg %% 8 for %d; histogram:", match);
  for (int i = 0; i < 16; ++i) printf(" %d", hist[i]);
  printf("\n");
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch, int flushes) {
    (void)hipMemset(cnt, 0, (size_t)B * 8 * 16 * stride * sizeof(int));
    for (int i = 0; i < 3; ++i) launch(flushes);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch(flushes);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // lost updates? every launch adds B*ipi*flushes*64 to the counters in total
    static int hc[32 * 8 * 16 * 32];
    (void)hipMemcpy(hc, cnt, sizeof(hc), hipMemcpyDeviceToHost);
    long long sum = 0;
    for (int i = 0; i < B * 8 * 16; ++i) sum += hc[i * stride];
    printf("%-28s flushes %d: %7.1f us per launch; counted %lld of %lld\n", name, flushes, ms / 20 * 1e3, sum, 23ll * B * ipi * flushes * 64);
  };
  for (int f : {1, 4}) {
    run("agent-scope atomics", [&](int fl) { hipLaunchKernelGGL((append_kernel<0>), dim3(B * ipi), dim3(64), 0, 0, cnt, lists, cap, fl, ipi, stride); }, f);
    run("L2 atomics per XCD", [&](int fl) { hipLaunchKernelGGL((append_kernel<1>), dim3(B * ipi), dim3(64), 0, 0, cnt, lists, cap, fl, ipi, stride); }, f);
  }
  return 0;
}
