// Diagnostic: LDS limits and occupancy the runtime reports for single-wave workgroups.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void k(float* o) { extern __shared__ float s[]; s[threadIdx.x] = 1; o[threadIdx.x] = s[threadIdx.x ^ 1]; }
// census: how many single-wave workgroups are really co-resident per CU
__global__ __launch_bounds__(64) void census(int* counter, int* maxseen, int spin) {
  extern __shared__ float s[];
  s[threadIdx.x] = 1;
  if (threadIdx.x == 0) {
    int v = atomicAdd(counter, 1) + 1;
    atomicMax(maxseen, v);
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    atomicMax(maxseen, atomicAdd(counter, 0));
    atomicSub(counter, 1);
  }
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d maxThreadsPerMP %d\n", p.multiProcessorCount, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock, p.maxThreadsPerMultiProcessor);
  int sizes[] = {1024, 5632, 6656, 10240, 11648, 13568, 20480, 23808, 32768, 65536};
  for (int sz : sizes) {
    int nb = -1; hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64, sz);
    int *c, *m; hipMalloc(&c, 4); hipMalloc(&m, 4); hipMemset(c, 0, 4); hipMemset(m, 0, 4);
    hipLaunchKernelGGL(census, dim3(256 * 40), dim3(64), sz, 0, c, m, 200000);
    hipError_t e2 = hipDeviceSynchronize();
    int hm = 0; hipMemcpy(&hm, m, 4, hipMemcpyDeviceToHost);
    printf("lds %6d: api blocks/CU %d (%s)  census max co-resident %d => %.2f per CU (%s)\n", sz, nb, hipGetErrorString(e), hm, hm / 256.0, hipGetErrorString(e2));
    hipFree(c); hipFree(m);
  }
  return 0;
}
