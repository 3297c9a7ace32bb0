// Microbenchmark: does independent VALU work co-issue with fp32 MFMA on gfx950?  NV = independent
// v_fma_f32 instructions inserted per MFMA (the conv_gemm kernels carry ~4.5 VALU per MFMA).
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NV>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[j & 7]) : "v"(b));
      }
    a += 1e-6f;
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV> void run(float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 1024, iters = 60;
  k<NV><<<blocks, 256>>>(d, iters); hipDeviceSynchronize();
  hipEventRecord(e0); k<NV><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flop = (double)blocks * 4 * iters * 8 * 4 * 4096.0;
  printf("VALU per MFMA %d: %.3f ms  %.1f TFLOP/s (MFMA)\n", NV, ms, flop / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 8192 * 256 * 4);
  run<0>(d); run<1>(d); run<2>(d); run<4>(d); run<6>(d); run<8>(d); run<12>(d); run<16>(d);
  return 0;
}
