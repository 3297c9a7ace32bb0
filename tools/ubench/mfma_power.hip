// Microbenchmark (dev tool): what the matrix pipe SUSTAINS on the whole chip for the float32x3 product
// stream -- v_mfma_f32_32x32x16_bf16 only, no LDS, no memory -- as a function of the operand DATA.
// The conv kernels issue, per 16-deep K step and wave, 6 products x (TI x TJ) tiles from three bf16
// pieces of each operand; this kernel does exactly that from one register-resident operand set (24 fragments:
// consecutive MFMAs see different A / B fragments exactly as in the product's K step).
//   data 0: all operands zero          data 1: every element 1.0 (pieces 1, 0, 0)
//   data 2: N(0,1) values split exactly into three bf16 pieces (what the product computes on)
//   data 3: as 2, but only the leading piece non-zero (the `bfloat16` mode's stream, NP = 1 -> 1 product)
// Prints TFLOP/s of bf16 MFMA work for a series of launches sized like the gate kernel's series.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ inline unsigned short bf16_rn(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ inline float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

template <int TI, int TJ, int NPROD, int WPE>
__global__ __launch_bounds__(256, WPE) void k(const float* __restrict__ src, float* out, int iters) {
  // operand sets: [set][piece][tile] fragments of 8 bf16 per lane
  uint4 A[1][3][TI], B[1][3][TJ];
  const float* p = src + (size_t)threadIdx.x * 8;
  int q = 0;
#pragma unroll
  for (int s = 0; s < 1; ++s)
#pragma unroll
    for (int t = 0; t < TI + TJ; ++t) {
      unsigned short h[8], m[8], l[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float x = p[(size_t)(q * 256) * 8 + e];
        h[e] = bf16_rn(x); float r = x - bf16_f(h[e]);
        m[e] = bf16_rn(r); l[e] = bf16_rn(r - bf16_f(m[e]));
      }
      ++q;
      uint4 H = {(unsigned)h[0] | (unsigned)h[1] << 16, (unsigned)h[2] | (unsigned)h[3] << 16,
                 (unsigned)h[4] | (unsigned)h[5] << 16, (unsigned)h[6] | (unsigned)h[7] << 16};
      uint4 M = {(unsigned)m[0] | (unsigned)m[1] << 16, (unsigned)m[2] | (unsigned)m[3] << 16,
                 (unsigned)m[4] | (unsigned)m[5] << 16, (unsigned)m[6] | (unsigned)m[7] << 16};
      uint4 L = {(unsigned)l[0] | (unsigned)l[1] << 16, (unsigned)l[2] | (unsigned)l[3] << 16,
                 (unsigned)l[4] | (unsigned)l[5] << 16, (unsigned)l[6] | (unsigned)l[7] << 16};
      if (t < TI) { A[s][0][t] = H; A[s][1][t] = M; A[s][2][t] = L; }
      else { B[s][0][t - TI] = H; B[s][1][t - TI] = M; B[s][2][t - TI] = L; }
    }
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // the six products of the float32x3 mode, in the product's order (piece of A, piece of B)
  constexpr int PA[6] = {0, 0, 1, 0, 1, 2}, PB[6] = {0, 1, 0, 2, 1, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int pr = 0; pr < NPROD; ++pr)
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TJ; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                __builtin_bit_cast(bf16x8, A[0][PA[pr]][i]), __builtin_bit_cast(bf16x8, B[0][PB[pr]][j]),
                acc[i][j], 0, 0, 0);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

static float gauss() {
  float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = (rand() + 1.0f) / (RAND_MAX + 2.0f);
  return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v);
}

template <int TI, int TJ, int NPROD, int WPE>
static void run(const char* name, const float* dsrc, float* dout, int blocks, int iters, int launches) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) k<TI, TJ, NPROD, WPE><<<blocks, 256>>>(dsrc, dout, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int l = 0; l < launches; ++l) k<TI, TJ, NPROD, WPE><<<blocks, 256>>>(dsrc, dout, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mfma = (double)blocks * 4 * iters * 2 * NPROD * TI * TJ * launches;
  double flop = mfma * 2.0 * 32 * 32 * 16;
  printf("%-34s blocks %4d  %7.1f us/launch  %7.1f TFLOP/s bf16  (%.3f of 2500)\n", name, blocks,
         ms * 1e3 / launches, flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
}

int main(int argc, char** argv) {
  const size_t n = (size_t)64 * 256 * 8;
  std::vector<float> h(n);
  float *dsrc[4], *dout;
  hipMalloc(&dout, 4096 * 256 * 4);
  srand(7);
  for (int d = 0; d < 4; ++d) {
    for (size_t i = 0; i < n; ++i) {
      float g = gauss();
      if (d == 0) h[i] = 0.f;
      else if (d == 1) h[i] = 1.f;
      else if (d == 2) h[i] = g;
      else { unsigned u; memcpy(&u, &g, 4); u &= 0xffff0000u; memcpy(&h[i], &u, 4); }
    }
    hipMalloc(&dsrc[d], n * 4);
    hipMemcpy(dsrc[d], h.data(), n * 4, hipMemcpyHostToDevice);
  }
  const char* dn[4] = {"zeros", "ones", "N(0,1) split3", "N(0,1) bf16 only"};
  // one gate-kernel tile = 36 K-steps of 96 MFMAs per wave: 512 tiles on 256 persistent blocks = 36 iters
  // (each iteration is two K-steps)
  const int iters = 36, launches = 20;
  for (int d = 0; d < 4; ++d) {
    char nm[96];
    snprintf(nm, sizeof nm, "4x4 tiles, 6 products, %s", dn[d]);
    run<4, 4, 6, 1>(nm, dsrc[d], dout, 256, iters, launches);
  }
  for (int d = 0; d < 4; ++d) {
    char nm[96];
    snprintf(nm, sizeof nm, "4x4 tiles, 1 product, %s", dn[d]);
    run<4, 4, 1, 1>(nm, dsrc[d], dout, 256, iters * 6, launches);
  }
  for (int d = 0; d < 4; d += 2) {
    char nm[96];
    snprintf(nm, sizeof nm, "2x4 tiles x2 WG/CU, 6 prod, %s", dn[d]);
    run<2, 4, 6, 2>(nm, dsrc[d], dout, 512, iters, launches);
  }
  // a quarter of the chip: is the per-CU rate higher when the rest idles?
  for (int d = 0; d < 4; d += 2) {
    char nm[96];
    snprintf(nm, sizeof nm, "4x4, 6 prod, 64 blocks, %s", dn[d]);
    run<4, 4, 6, 1>(nm, dsrc[d], dout, 64, iters, launches);
  }
  return 0;
}
