// Microbenchmark: fp32 MFMA issue rate under the conv_gemm wave shape (dev tool).
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NACC>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    a += 1e-6f;
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 8192 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {960, 1024, 1920, 2048, 3840, 4096, 7680}) {
    const int iters = 32 * 1920 / blocks;   // constant total work = one K4 launch (1024 MFMAs/wave at 1920 blocks)
    k<4><<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); k<4><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 8 * 4 * 4096.0;
    printf("blocks %d: %.3f ms  %.1f TFLOP/s\n", blocks, ms, flop / ms / 1e9);
  }
  return 0;
}
