// Microbenchmark of the fp32 MFMA conv-GEMM main loop (the ResidualBlock's dilated conv at BASELINE
// configs[1]: M = 256 output rows, K = 2 taps x 256 channels, N = B*T = 16 x 7680 columns), used to
// choose the LDS image / fragment-read scheme of csrc/conv_gemm.hip.  Every variant computes the
// same Y[b][m][t] = sum_{tap,ci} W[m][ci][tap] * X[b][ci][t - (1-tap)*dil] with a trivial epilogue
// and is checked on sampled outputs.
//   V0  the round-1 loop: As[k][m], Bs[k][n], one ds_read_b32 per operand value (ds_read2_b32)
//   V1  fragment-ordered LDS images: A packed so that a lane's (2 k-steps x 2 row tiles) are one
//       ds_read_b128; B columns interleaved (lane li <-> columns 2li, 2li+1) and k-paired so that a
//       lane's (2 k-steps x 2 column tiles) are one ds_read_b128; all 8 fragments of a K step read
//       up front; float2 epilogue stores
//   V2  V1 with the fragment reads split in two halves (16 fragment registers)
//   V3  V1 + s_setprio(1) around the MFMA block
//   V4  V1 with A staged by global_load_lds (no VGPRs, no ds_write for A)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 conv_loop.hip -o conv_loop ; run: ./conv_loop [dil]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int BM = 256, BN = 128, BK = 16, NT = 512;

struct Args {
  const float* x;      // (B, Cin, T)
  const float* wpk;    // packed weights, layout per variant
  float* y;            // (B, 256, T)
  int Cin, T, B, dil, ntile_n;
};

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)

__device__ __forceinline__ void tile_of(const Args& a, int& b, int& t0) {
  const int nblk = gridDim.x;
  const int id = blockIdx.x;
  const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
  const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  const int nt = logical % a.ntile_n;
  b = logical / a.ntile_n;
  t0 = nt * BN;
}

// ------------------------------------------------------------------------------------------------
// V0: round-1 structure.  wpk layout: [tap][k][m] (m contiguous, 256).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT, 2) void conv_v0(const Args a) {
  __shared__ float As[2][BK][BM];
  __shared__ float Bs[2][BK][BN];
  int b, t0; tile_of(a, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = 2 * a.Cin / BK;
  const int a_k = tid / 64, a_col = (tid % 64) * 4;
  const int v_k = tid >> 5, v_col = (tid & 31) * 4;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  float4 ra0, ra1, rb0;
  auto load = [&](int it) {
    const int tap = it / (a.Cin / BK), c0 = (it % (a.Cin / BK)) * BK;
    const float* wp = a.wpk + ((long)(tap * a.Cin + c0 + a_k)) * BM + a_col;
    ra0 = *reinterpret_cast<const float4*>(wp);
    ra1 = *reinterpret_cast<const float4*>(wp + 8L * BM);
    const int tw = t0 - (1 - tap) * a.dil;
    rb0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tw >= 0) rb0 = *reinterpret_cast<const float4*>(xb + (long)(c0 + v_k) * a.T + tw + v_col);
  };
  auto store = [&](int buf) {
    *reinterpret_cast<float4*>(&As[buf][a_k][a_col]) = ra0;
    *reinterpret_cast<float4*>(&As[buf][a_k + 8][a_col]) = ra1;
    *reinterpret_cast<float4*>(&Bs[buf][v_k][v_col]) = rb0;
  };
  load(0); store(0); __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    if (it + 1 < nk) load(it + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float a0 = As[cur][kk * 2 + lk][wm * 64 + li];
      const float a1 = As[cur][kk * 2 + lk][wm * 64 + 32 + li];
      const float b0 = Bs[cur][kk * 2 + lk][wn * 64 + li];
      const float b1 = Bs[cur][kk * 2 + lk][wn * 64 + 32 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (it + 1 < nk) store(cur ^ 1);
    __syncthreads();
  }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        yb[(long)m * a.T + t0 + wn * 64 + ni * 32 + li] = acc[mi][ni][r];
      }
}

// ------------------------------------------------------------------------------------------------
// V1..V4: fragment-ordered images.
//   A image of one K step (16 k x 256 m = 4096 floats): float index
//       ((((kk2*2 + lk)*4 + wm)*32 + li)*4 + j*2 + mi)  <->  k = 4*kk2 + 2*j + lk, m = wm*64 + mi*32 + li
//   B image (16 k x 128 n = 2048 floats): float index
//       ((((kk2*2 + lk)*2 + wn)*32 + li)*4 + j*2 + ni)  <->  k = 4*kk2 + 2*j + lk, n = wn*64 + 2*li + ni
//   wpk: [tap][kstep][4096] in the A image order.
// ------------------------------------------------------------------------------------------------
template <int VAR>
__global__ __launch_bounds__(NT, 2) void conv_v1(const Args a) {
  __shared__ float4 As[2][BK * BM / 4];
  __shared__ float4 Bs[2][BK * BN / 4];
  int b, t0; tile_of(a, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ksteps = a.Cin / BK;
  const int nk = 2 * ksteps;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  // B staging role: row v_k (k within the step), columns v_col .. v_col+3
  const int v_k = tid >> 5, v_col = (tid & 31) * 4;
  // destination of this thread's two column pairs in the B image (float2 units)
  const int b_kk2 = v_k >> 2, b_j = (v_k >> 1) & 1, b_lk = v_k & 1;
  const int b_wn = v_col >> 6, b_p = (v_col & 63) >> 1;
  const int b_dst = ((((b_kk2 * 2 + b_lk) * 2 + b_wn) * 32 + b_p) * 4 + b_j * 2) / 2;   // float2 index; +2 for the next pair
  float4 ra0, ra1, rb0;
  auto load = [&](int it) {
    const int tap = it / ksteps, ks = it % ksteps;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + ((long)(tap * ksteps + ks)) * (BK * BM / 4);
    if (VAR != 4) {
      ra0 = wp[tid];
      ra1 = wp[tid + NT];
    }
    const int tw = t0 - (1 - tap) * a.dil;
    rb0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tw >= 0) rb0 = *reinterpret_cast<const float4*>(xb + (long)(ks * BK + v_k) * a.T + tw + v_col);
  };
  auto load_a_lds = [&](int it, int buf) {     // VAR == 4: A straight into LDS
    const int tap = it / ksteps, ks = it % ksteps;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + ((long)(tap * ksteps + ks)) * (BK * BM / 4);
    // wave-uniform LDS base + lane*16: wave w copies float4 [w*64, w*64+64) and [512 + w*64, ...)
    __builtin_amdgcn_global_load_lds(wp + tid, (__attribute__((address_space(3))) void*)&As[buf][wave * 64], 16, 0, 0);
    __builtin_amdgcn_global_load_lds(wp + tid + NT, (__attribute__((address_space(3))) void*)&As[buf][NT + wave * 64], 16, 0, 0);
  };
  auto store = [&](int buf) {
    if (VAR != 4) {
      As[buf][tid] = ra0;
      As[buf][tid + NT] = ra1;
    }
    float2* bd = reinterpret_cast<float2*>(&Bs[buf][0]);
    bd[b_dst] = make_float2(rb0.x, rb0.y);
    bd[b_dst + 2] = make_float2(rb0.z, rb0.w);
  };
  if (VAR == 4) load_a_lds(0, 0);
  load(0); store(0);
  if (VAR == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int fa = ((lk * 4 + wm) * 32 + li);        // float4 index of this lane's A fragment at kk2 = 0
  const int fb = ((lk * 2 + wn) * 32 + li);
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    const bool more = it + 1 < nk;
    if (more) {
      if (VAR == 4) load_a_lds(it + 1, cur ^ 1);
      load(it + 1);
    }
    if (VAR == 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float4 af[2], bf[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          af[q] = As[cur][fa + (2 * h + q) * 256];
          bf[q] = Bs[cur][fb + (2 * h + q) * 128];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bf[q].x, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bf[q].y, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bf[q].x, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bf[q].y, acc[1][1], 0, 0, 0);
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bf[q].z, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bf[q].w, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bf[q].z, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bf[q].w, acc[1][1], 0, 0, 0);
        }
      }
    } else {
      float4 af[4], bf[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        af[q] = As[cur][fa + q * 256];
        bf[q] = Bs[cur][fb + q * 128];
      }
      if (VAR == 3) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bf[q].x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].x, bf[q].y, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bf[q].x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].y, bf[q].y, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bf[q].z, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].z, bf[q].w, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bf[q].z, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[q].w, bf[q].w, acc[1][1], 0, 0, 0);
      }
      if (VAR == 3) __builtin_amdgcn_s_setprio(0);
    }
    if (more) store(cur ^ 1);
    if (VAR == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      *reinterpret_cast<float2*>(&yb[(long)m * a.T + t0 + wn * 64 + 2 * li]) = make_float2(acc[mi][0][r], acc[mi][1][r]);
    }
}

// ------------------------------------------------------------------------------------------------
static void pack_v0(const std::vector<float>& W, int Cin, std::vector<float>& out) {   // W[m][ci][tap]
  out.assign((size_t)2 * Cin * BM, 0.f);
  for (int tap = 0; tap < 2; ++tap)
    for (int k = 0; k < Cin; ++k)
      for (int m = 0; m < BM; ++m) out[((size_t)tap * Cin + k) * BM + m] = W[((size_t)m * Cin + k) * 2 + tap];
}
static void pack_v1(const std::vector<float>& W, int Cin, std::vector<float>& out) {
  out.assign((size_t)2 * Cin * BM, 0.f);
  const int ksteps = Cin / BK;
  for (int tap = 0; tap < 2; ++tap)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int kk2 = 0; kk2 < 4; ++kk2) for (int lk = 0; lk < 2; ++lk) for (int wm = 0; wm < 4; ++wm)
        for (int li = 0; li < 32; ++li) for (int j = 0; j < 2; ++j) for (int mi = 0; mi < 2; ++mi) {
          const int k = ks * BK + 4 * kk2 + 2 * j + lk, m = wm * 64 + mi * 32 + li;
          const size_t idx = ((size_t)(tap * ksteps + ks)) * (BK * BM) + ((((kk2 * 2 + lk) * 4 + wm) * 32 + li) * 4 + j * 2 + mi);
          out[idx] = W[((size_t)m * Cin + k) * 2 + tap];
        }
}

int main(int argc, char** argv) {
  const int B = 16, T = 7680, Cin = 256;
  const int dil = argc > 1 ? atoi(argv[1]) : 64;
  const int reps = 20;
  std::vector<float> hx((size_t)B * Cin * T), hW((size_t)BM * Cin * 2);
  srand(1);
  for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hW) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  float *dx, *dw0, *dw1, *dy;
  CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dy, (size_t)B * BM * T * 4));
  CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> p0, p1; pack_v0(hW, Cin, p0); pack_v1(hW, Cin, p1);
  CHECK(hipMalloc(&dw0, p0.size() * 4)); CHECK(hipMalloc(&dw1, p1.size() * 4));
  CHECK(hipMemcpy(dw0, p0.data(), p0.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw1, p1.data(), p1.size() * 4, hipMemcpyHostToDevice));
  Args a; a.x = dx; a.y = dy; a.Cin = Cin; a.T = T; a.B = B; a.dil = dil; a.ntile_n = T / BN;
  const int grid = B * (T / BN);
  const double flop = 2.0 * B * T * BM * Cin * 2;
  std::vector<float> hy((size_t)B * BM * T);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto check = [&](const char* name) {
    CHECK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int s = 0; s < 200; ++s) {
      const int b = rand() % B, m = rand() % BM, t = (s < 20) ? (s * 7) % 200 : rand() % T;
      double ref = 0;
      for (int tap = 0; tap < 2; ++tap) {
        const int ts = t - (1 - tap) * dil;
        if (ts < 0) continue;
        for (int c = 0; c < Cin; ++c) ref += (double)hW[((size_t)m * Cin + c) * 2 + tap] * hx[((size_t)b * Cin + c) * T + ts];
      }
      const double err = fabs(ref - hy[((size_t)b * BM + m) * T + t]);
      if (err > worst) worst = err;
    }
    printf("  %-4s max |err| on 200 samples: %.3e %s\n", name, worst, worst < 1e-4 ? "ok" : "WRONG");
  };
  auto run = [&](const char* name, void (*kern)(const Args), const float* w) {
    a.wpk = w;
    CHECK(hipMemset(dy, 0, hy.size() * 4));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), 0, 0, a);
    CHECK(hipDeviceSynchronize());
    check(name);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-4s dil %4d: %8.1f us/launch  %6.1f TFLOP/s  (%.3f of 157.3)\n", name, dil, 1e3 * ms / reps, flop / (ms / reps) / 1e9, flop / (ms / reps) / 1e9 / 157.3);
  };
  for (int round = 0; round < 2; ++round) {
    run("V0", conv_v0, dw0);
    run("V1", conv_v1<1>, dw1);
    run("V2", conv_v1<2>, dw1);
    run("V3", conv_v1<3>, dw1);
    run("V4", conv_v1<4>, dw1);
  }
  return 0;
}
