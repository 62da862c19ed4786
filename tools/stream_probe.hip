// Diagnostic microbenchmark (not part of the library): how fast can gfx950 stream a 205 MB fp32
// tensor from HBM with the access shapes score_filter could use, right after a kernel that left
// ~250 MB of dirty lines behind (as mask_assemble does every step)?
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_probe tools/stream_probe.hip && tools/stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void dirty_kernel(float4* w, size_t n4, int nt) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  typedef float nat4 __attribute__((ext_vector_type(4)));
  nat4 nv = {1.f, 2.f, 3.f, 4.f};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    if (nt) __builtin_nontemporal_store(nv, reinterpret_cast<nat4*>(w + i)); else w[i] = v;
  }
}

// v0 / v1: grid-stride float4 loads (optionally non-temporal), UNR independent loads in flight per lane
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void read_plain(const float4* __restrict__ a, size_t n4, float* out) {
  typedef float nat4 __attribute__((ext_vector_type(4)));
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
    nat4 v[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const nat4* p = reinterpret_cast<const nat4*>(a + i + u * stride);
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; i < n4; i += stride) { const float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) out[0] = acc;
}

// v2: persistent waves, each block of BLK contiguous bytes is read by one wave with UNR float4 loads
// per lane in flight (chunked: wave w takes blocks w, w+G, ...)
template <int UNR>
__global__ __launch_bounds__(64) void read_tiles(const char* __restrict__ a, size_t bytes, int tile_bytes, float* out) {
  const int lane = threadIdx.x;
  const size_t ntiles = bytes / tile_bytes;
  float acc = 0.f;
  for (size_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const char* base = a + t * (size_t)tile_bytes;
    for (int o = lane * 16; o < tile_bytes; o += 1024 * UNR) {
      float4 v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int oo = o + u * 1024;
        v[u] = *reinterpret_cast<const float4*>(base + (oo < tile_bytes ? oo : tile_bytes - 16));
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

typedef __attribute__((address_space(3))) void lds_void;
// v3: LDS-DMA, persistent single-wave workgroups, NBUF tiles of tile_bytes (multiple of 1024) in flight
__global__ __launch_bounds__(64) void read_ldsdma(const char* __restrict__ a, size_t bytes, int tile_bytes, int nbuf, float* out) {
  extern __shared__ __align__(16) char lds[];
  const int lane = threadIdx.x;
  const uint32_t lane16 = lane * 16u;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_void*)lds;
  const size_t ntiles = bytes / tile_bytes;
  const int npiece = tile_bytes / 1024;
  auto issue = [&](size_t t, int buf) {
    const char* g = a + t * (size_t)tile_bytes;
    uint32_t dst = lds_base + buf * tile_bytes;
    for (int i = 0; i < npiece; ++i, g += 1024, dst += 1024)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane16), "s"(g), "s"(dst) : "memory");
  };
  size_t t = blockIdx.x, tp = blockIdx.x;
  int pend = 0, pbuf = 0;
  for (int d = 0; d < nbuf && tp < ntiles; ++d, tp += gridDim.x) { issue(tp, pbuf); pbuf = (pbuf + 1) % nbuf; ++pend; }
  float acc = 0.f;
  int cbuf = 0;
  for (; t < ntiles; t += gridDim.x) {
    // wait for the oldest tile: (pend-1)*npiece instructions may stay in flight
    const int allow = (pend - 1) * npiece;
    if (allow == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (allow <= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (allow <= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (allow <= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (allow <= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (allow <= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
    --pend;
    acc += reinterpret_cast<const float*>(lds + cbuf * tile_bytes)[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (tp < ntiles) { issue(tp, cbuf); tp += gridDim.x; ++pend; }
    cbuf = (cbuf + 1) % nbuf;
  }
  if (acc == 123.456f) out[0] = acc;
}

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ a, float4* __restrict__ w, size_t n4) {
  typedef float nat4 __attribute__((ext_vector_type(4)));
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const nat4* p = reinterpret_cast<const nat4*>(a + i);
    const nat4 v = NTL ? __builtin_nontemporal_load(p) : *p;
    if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<nat4*>(w + i)); else *reinterpret_cast<nat4*>(w + i) = v;
  }
}

int main(int argc, char** argv) {
  const size_t A_BYTES = (size_t)32 * 19248 * 81 * 4;   // 199.6 MB of logits
  const size_t W_BYTES = (size_t)32 * 100 * 19044 * 4;  // 243.8 MB of masks
  char *a, *w; float* out;
  CK(hipMalloc(&a, A_BYTES + 65536)); CK(hipMalloc(&w, W_BYTES)); CK(hipMalloc(&out, 256));
  CK(hipMemset(a, 0, A_BYTES + 65536));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int REPS = 12;
  auto run = [&](const char* name, int dirty_mode, auto launch) {
    float tot = 0.f; int cnt = 0;
    for (int r = 0; r < REPS; ++r) {
      if (dirty_mode) hipLaunchKernelGGL(dirty_kernel, dim3(256 * 8), dim3(256), 0, 0, (float4*)w, W_BYTES / 16, dirty_mode == 2);
      CK(hipEventRecord(e0, 0));
      launch();
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { tot += ms; ++cnt; }
    }
    CK(hipGetLastError());
    const float ms = tot / cnt;
    printf("%-44s dirty=%d  %.1f us  %.2f TB/s\n", name, dirty_mode, ms * 1e3, A_BYTES / (ms * 1e-3) / 1e12);
    fflush(stdout);
  };
  const size_t n4 = A_BYTES / 16;
  {
    // steady-state write-only and copy rates (bytes counted once for writes, read+write for copies)
    auto timeit = [&](const char* name, double bytes, auto launch) {
      float tot = 0.f; int cnt = 0;
      for (int r = 0; r < REPS; ++r) {
        CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) { tot += ms; ++cnt; }
      }
      const float ms = tot / cnt;
      printf("%-52s %.1f us  %.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12); fflush(stdout);
    };
    timeit("write-only 244MB default stores (steady state)", (double)W_BYTES, [&] { hipLaunchKernelGGL(dirty_kernel, dim3(2048), dim3(256), 0, 0, (float4*)w, W_BYTES / 16, 0); });
    timeit("write-only 244MB nt stores (steady state)", (double)W_BYTES, [&] { hipLaunchKernelGGL(dirty_kernel, dim3(2048), dim3(256), 0, 0, (float4*)w, W_BYTES / 16, 1); });
    timeit("copy 200MB->200MB default/default (r+w bytes)", 2.0 * A_BYTES, [&] { hipLaunchKernelGGL((copy_kernel<false, false>), dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)w, n4); });
    timeit("copy 200MB->200MB nt load/default store", 2.0 * A_BYTES, [&] { hipLaunchKernelGGL((copy_kernel<true, false>), dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)w, n4); });
    timeit("copy 200MB->200MB default load/nt store", 2.0 * A_BYTES, [&] { hipLaunchKernelGGL((copy_kernel<false, true>), dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)w, n4); });
    timeit("copy 200MB->200MB nt/nt", 2.0 * A_BYTES, [&] { hipLaunchKernelGGL((copy_kernel<true, true>), dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)w, n4); });
    // alternate: read A (default / nt) then write W, as the pipeline does; report the pair
    for (int ntl = 0; ntl <= 1; ++ntl) for (int nts = 0; nts <= 1; ++nts) {
      char nm[128]; snprintf(nm, sizeof nm, "pair: read 200MB (%s) + write 244MB (%s)", ntl ? "nt" : "default", nts ? "nt" : "default");
      timeit(nm, (double)A_BYTES + (double)W_BYTES, [&] {
        if (ntl) hipLaunchKernelGGL((read_plain<4, true>), dim3(2048), dim3(256), 0, 0, (const float4*)a, n4, out);
        else hipLaunchKernelGGL((read_plain<4, false>), dim3(2048), dim3(256), 0, 0, (const float4*)a, n4, out);
        hipLaunchKernelGGL(dirty_kernel, dim3(2048), dim3(256), 0, 0, (float4*)w, W_BYTES / 16, nts);
      });
    }
  }
  for (int dm = 1; dm <= 1 && argc > 1; ++dm) {
    run("plain float4 grid-stride 2048x256 unr1", dm, [&] { hipLaunchKernelGGL((read_plain<1, false>), dim3(2048), dim3(256), 0, 0, (const float4*)a, n4, out); });
    run("plain float4 grid-stride 2048x256 unr4", dm, [&] { hipLaunchKernelGGL((read_plain<4, false>), dim3(2048), dim3(256), 0, 0, (const float4*)a, n4, out); });
    run("plain float4 grid-stride 4096x256 unr4", dm, [&] { hipLaunchKernelGGL((read_plain<4, false>), dim3(4096), dim3(256), 0, 0, (const float4*)a, n4, out); });
    run("plain float4 grid-stride 2048x256 unr4 nt", dm, [&] { hipLaunchKernelGGL((read_plain<4, true>), dim3(2048), dim3(256), 0, 0, (const float4*)a, n4, out); });
    run("plain float4 one-shot (n4/256 blocks) unr1", dm, [&] { hipLaunchKernelGGL((read_plain<1, false>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const float4*)a, n4, out); });
    run("wave tiles 10368B, 7 waves/CU, unr4", dm, [&] { hipLaunchKernelGGL((read_tiles<4>), dim3(256 * 7), dim3(64), 0, 0, a, A_BYTES, 10368, out); });
    run("wave tiles 10368B, 16 waves/CU, unr4", dm, [&] { hipLaunchKernelGGL((read_tiles<4>), dim3(256 * 16), dim3(64), 0, 0, a, A_BYTES, 10368, out); });
    run("wave tiles 10368B, 16 waves/CU, unr11", dm, [&] { hipLaunchKernelGGL((read_tiles<11>), dim3(256 * 16), dim3(64), 0, 0, a, A_BYTES, 10368, out); });
    for (int wpc : {4, 7}) for (int nbuf : {1, 2, 3}) {
      const int tb = 10240;
      if ((size_t)wpc * nbuf * tb > 150 * 1024) continue;
      char nm[128]; snprintf(nm, sizeof nm, "LDS-DMA tiles %dB, %d waves/CU, %d in flight", tb, wpc, nbuf);
      run(nm, dm, [&] { hipLaunchKernelGGL(read_ldsdma, dim3(256 * wpc), dim3(64), (size_t)nbuf * tb, 0, a, A_BYTES, tb, nbuf, out); });
    }
    for (int wpc : {1, 2}) {
      const int tb = 16384, nbuf = wpc == 1 ? 3 : 3;
      char nm[128]; snprintf(nm, sizeof nm, "LDS-DMA tiles %dB, %d waves/CU, %d in flight", tb, wpc, nbuf);
      run(nm, dm, [&] { hipLaunchKernelGGL(read_ldsdma, dim3(256 * wpc), dim3(64), (size_t)nbuf * tb, 0, a, A_BYTES, tb, nbuf, out); });
    }
  }
  return 0;
}
