// probe: 8-byte buffer loads at 2-byte-aligned addresses on gfx950 (do they return the right bytes, and at what cost?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int rsrc_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned short* p, int shift, unsigned long long* out, int n, int iters) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p), 0, 0x7fffffff, 0x00020000);
  unsigned long long acc = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned i = (blockIdx.x * blockDim.x + threadIdx.x + it * 7919u) % (unsigned)(n / 4 - 2);
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, 8u * i, 2u * (unsigned)shift, 0);
    acc += ((unsigned long long)v[1] << 32) | v[0];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  const int n = 1 << 24;
  std::vector<unsigned short> h(n);
  for (int i = 0; i < n; ++i) h[i] = (unsigned short)(i * 2654435761u >> 13);
  unsigned short* d; unsigned long long* o;
  hipMalloc(&d, n * 2); hipMalloc(&o, 256 * 1024 * 8);
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 4; ++shift) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<1024, 256>>>(d, shift, o, n, 1);
    hipDeviceSynchronize();
    std::vector<unsigned long long> got(256 * 1024);
    hipMemcpy(got.data(), o, got.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256 * 1024; ++t) {
      const unsigned i = (unsigned)t % (unsigned)(n / 4 - 2);
      unsigned long long want = 0;
      for (int e = 3; e >= 0; --e) want = (want << 16) | h[4 * i + shift + e];
      if (want != got[t]) ++bad;
    }
    hipEventRecord(e0); k<<<1024, 256>>>(d, shift, o, n, 64); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("shift %d elements (byte offset %d): %d wrong of %d, 64 loads/thread: %.1f us\n", shift, 2 * shift, bad, 256 * 1024, ms * 1e3);
  }
  return 0;
}
