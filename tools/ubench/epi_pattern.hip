// Does the MFMA C-layout access pattern of the conv epilogues (one dword per lane, a wave-level
// access = two 128-byte row segments, 64 of them per lane and tile) cost HBM bandwidth against a
// row-contiguous float4 pattern?  Both kernels do y = x + 1 over a (B, 256, T) tensor in 256 x 128
// tiles of 512 threads, one tile per workgroup, 1 workgroup per CU (LDS padding), tiles in the
// product's order.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 epi_pattern.hip -o epi_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)
constexpr int ROWS = 256, BN = 128;
__global__ __launch_bounds__(512, 2) void k_mfma_layout(const float* __restrict__ x, float* __restrict__ y, int T, int ntile_n) {
  __shared__ float pad[20000];                     // 80 KB: one workgroup per CU, like the product kernels
  if (threadIdx.x == 9999) pad[0] = 0.f;
  const int tile = blockIdx.x, nt = tile % ntile_n, b = tile / ntile_n, t0 = nt * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const float* xb = x + (long)b * ROWS * T;
  float* yb = y + (long)b * ROWS * T;
  float v[4][16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int mi = q >> 1, ni = q & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      v[q][r] = xb[(long)row * T + t0 + wn * 64 + ni * 32 + li];
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int mi = q >> 1, ni = q & 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      yb[(long)row * T + t0 + wn * 64 + ni * 32 + li] = v[q][r] + 1.f;
    }
  }
}
__global__ __launch_bounds__(512, 2) void k_rows_f4(const float* __restrict__ x, float* __restrict__ y, int T, int ntile_n) {
  __shared__ float pad[20000];
  if (threadIdx.x == 9999) pad[0] = 0.f;
  const int tile = blockIdx.x, nt = tile % ntile_n, b = tile / ntile_n, t0 = nt * BN;
  const float* xb = x + (long)b * ROWS * T;
  float* yb = y + (long)b * ROWS * T;
  float4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const float4*>(xb + (long)(threadIdx.x / 32 + 16 * i) * T + t0 + 4 * (threadIdx.x % 32));
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float4 o = make_float4(v[i].x + 1.f, v[i].y + 1.f, v[i].z + 1.f, v[i].w + 1.f);
    *reinterpret_cast<float4*>(yb + (long)(threadIdx.x / 32 + 16 * i) * T + t0 + 4 * (threadIdx.x % 32)) = o;
  }
}
int main() {
  const int B = 16, T = 7680, ntile_n = T / BN, grid = B * ntile_n;
  const size_t n = (size_t)B * ROWS * T;
  float *x, *y; CHECK(hipMalloc(&x, n * 4)); CHECK(hipMalloc(&y, n * 4)); CHECK(hipMemset(x, 0, n * 4));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, void (*k)(const float*, float*, int, int)) {
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, x, y, T, ntile_n);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, x, y, T, ntile_n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.1f us  %.2f TB/s (read + write)\n", name, 1e3 * ms / 20, 2.0 * n * 4 / (ms / 20 * 1e-3) / 1e12);
  };
  run("MFMA C-layout dwords", k_mfma_layout);
  run("row-contiguous float4", k_rows_f4);
  run("MFMA C-layout dwords", k_mfma_layout);
  run("row-contiguous float4", k_rows_f4);
  return 0;
}
