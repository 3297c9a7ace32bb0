// Microbenchmark: cost of one all-to-all "edge" inside a persistent kernel (MI355X).
// G single-wave workgroups; per edge every workgroup publishes V/G (>=1) tagged 8-byte granules
// {float, tag} with agent-scope relaxed stores and then gathers all V granules, spinning on the tags.
// build: hipcc --offload-arch=gfx950 -O3 -o edge_latency edge_latency.hip ; run: ./edge_latency [G] [V] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void put(uint64_t* p, float v, uint32_t tag) {
  uint64_t g = ((uint64_t)tag << 32) | __float_as_uint(v);
  __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t get(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(64) void edge_kernel(uint64_t* box, int V, int iters, int nbox, float* sink, int* abort_flag, long long* cycles, int do_sleep) {
  const int g = blockIdx.x, G = gridDim.x, lane = threadIdx.x;
  const int per = (V + G - 1) / G;
  float acc = 0.f;
  long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    uint64_t* b = box + (size_t)(it % nbox) * V;
    const uint32_t tag = it / nbox + 1;
    // publish my elements
    if (lane < per && g * per + lane < V) put(b + g * per + lane, acc + it, tag);
    // gather everything: all loads of a sweep in flight, then the tag checks
    {
      const int nk = (V + 63) / 64;     // <= 16
      uint64_t v[16];
      int spins = 0;
      for (;;) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (i < nk) { const int k = lane + 64 * i; v[i] = k < V ? get(b + k) : ((uint64_t)tag << 32); }
#pragma unroll
        for (int i = 0; i < 16; ++i) if (i < nk) ok = ok && ((uint32_t)(v[i] >> 32) == tag);
        if (__all(ok)) break;
        if (++spins > 2000000) { *abort_flag = 1; break; }
        if (do_sleep) __builtin_amdgcn_s_sleep(1);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) if (i < nk) acc += __uint_as_float((uint32_t)v[i]) * 1e-9f;
    }
    if (*((volatile int*)abort_flag)) break;
  }
  long long t1 = wall_clock64();
  if (lane == 0) { sink[g] = acc; if (g == 0) *cycles = t1 - t0; }
}

int main(int argc, char** argv) {
  int G = argc > 1 ? atoi(argv[1]) : 256, V = argc > 2 ? atoi(argv[2]) : 256, iters = argc > 3 ? atoi(argv[3]) : 20000;
  const int nbox = 64;      // distinct mailboxes in rotation (a real step has ~45)
  uint64_t* box; float* sink; int* ab; long long* cyc;
  CHECK(hipMalloc(&box, (size_t)nbox * V * 8)); CHECK(hipMemset(box, 0, (size_t)nbox * V * 8));
  CHECK(hipMalloc(&sink, G * 4)); CHECK(hipMalloc(&ab, 4)); CHECK(hipMemset(ab, 0, 4)); CHECK(hipMalloc(&cyc, 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(edge_kernel, dim3(G), dim3(64), 0, 0, box, V, iters, nbox, sink, ab, cyc, argc > 4 ? atoi(argv[4]) : 1);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  int h_ab; CHECK(hipMemcpy(&h_ab, ab, 4, hipMemcpyDeviceToHost));
  printf("G=%d V=%d iters=%d: %.3f ms total, %.3f us per edge, abort=%d\n", G, V, iters, ms, 1e3 * ms / iters, h_ab);
  return 0;
}
