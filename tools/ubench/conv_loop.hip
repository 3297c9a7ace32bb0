// Microbenchmark of the fp32 MFMA conv-GEMM main loop (the ResidualBlock's dilated conv at BASELINE
// configs[1]: M = 256 output rows, K = 2 taps x 256 channels, N = B*T = 16 x 7680 columns), used to
// choose the LDS image / fragment-read scheme of csrc/conv_gemm_x3.hip.  Every variant computes the
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
//   V5*/V6*  diagnostic upper bounds (no global traffic / no barrier): not valid convolutions
//   V8  register prefetch two K steps ahead (hipcc serialises it with vmcnt(0): slower)
//   V11 persistent 2-workgroup/CU walk over the tiles (spills at 128 VGPRs: slower)
//   V12 LDS-DMA (global_load_lds_dwordx4 in inline asm) 3-stage ring, 8 waves along M, float4 epilogue;
//       modes isolate the cost of the A / B load streams, the barrier and the LDS reads
//   V13 one 4-wave workgroup per CU, 256 x 256 tile, fragments prefetched across k-pairs and stages
//       (DISABLED by default: intermittently wrong results -- a race we did not resolve -- and no faster)
//   V14 V13's in-wave pipelining with two 4-wave workgroups per CU (256 x 128 tiles)
//   V16 V11 without spills (buffer-resource stores): persistent 107-109, the same code one tile per workgroup 107-109
//   V15 V12 with a dedicated loader wave (9 waves; waves 0-7 issue no VMEM at all): 93 TFLOP/s, slower
// Measured on MI355X (T = 7680, dil 64; run-to-run +-3 %): V0 108-113, V1/V2 108-112, V4 112-114,
// V12 109-113, V14 112 TFLOP/s; V5* 125-132, V12 without loads 123-126, MFMA only (no LDS, no barrier,
// no loads) 127-136; exactly two residency rounds (T = 8192) +4 %.  Every load stream costs ~4-6 %
// whether it comes from HBM or from L2; latency hiding (distance-2 prefetch, LDS-DMA) changes nothing.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 conv_loop.hip -o conv_loop ; run: ./conv_loop [dil] [T]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int BM = 256, BN = 128, BK = 16, NT = 512;

struct Args {
  unsigned long long* clk;   // per block: {shader cycles, 100 MHz wall ticks} of the block's lifetime (conv_v1 only)
  const float* x;      // (B, Cin, T)
  const float* wpk;    // packed weights, layout per variant
  float* y;            // (B, 256, T)
  int Cin, T, B, dil, ntile_n;
  int stagger;   // V12 mode 32: delay (100 MHz ticks) of workgroups 256..511
  int mode;   // V12 diagnostics: 1 = no A loads, 2 = no B loads, 4 = B always from one L2-resident tile, 8 = no barrier, 16 = no LDS reads
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
// V8: V1 with the global loads issued TWO K steps ahead (two register sets): a load has a whole
// K step (~3.4 us of wall time on a busy SIMD) more to come back before its ds_write needs it.
__global__ __launch_bounds__(NT, 2) void conv_v8(const Args a) {
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
  const int v_k = tid >> 5, v_col = (tid & 31) * 4;
  const int b_kk2 = v_k >> 2, b_j = (v_k >> 1) & 1, b_lk = v_k & 1;
  const int b_wn = v_col >> 6, b_p = (v_col & 63) >> 1;
  const int b_dst = ((((b_kk2 * 2 + b_lk) * 2 + b_wn) * 32 + b_p) * 4 + b_j * 2) / 2;
  float4 ra0[2], ra1[2], rb0[2];
  auto load = [&](int it, auto setc) {
    constexpr int set = decltype(setc)::value;
    const int tap = it / ksteps, ks = it % ksteps;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + ((long)(tap * ksteps + ks)) * (BK * BM / 4);
    ra0[set] = wp[tid];
    ra1[set] = wp[tid + NT];
    const int tw = t0 - (1 - tap) * a.dil;
    rb0[set] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tw >= 0) rb0[set] = *reinterpret_cast<const float4*>(xb + (long)(ks * BK + v_k) * a.T + tw + v_col);
  };
  auto store = [&](int buf, auto setc) {
    constexpr int set = decltype(setc)::value;
    As[buf][tid] = ra0[set];
    As[buf][tid + NT] = ra1[set];
    float2* bd = reinterpret_cast<float2*>(&Bs[buf][0]);
    bd[b_dst] = make_float2(rb0[set].x, rb0[set].y);
    bd[b_dst + 2] = make_float2(rb0[set].z, rb0[set].w);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  load(0, I0{}); store(0, I0{});
  load(1, I1{});
  __syncthreads();
  const int fa = ((lk * 4 + wm) * 32 + li);
  const int fb = ((lk * 2 + wn) * 32 + li);
  auto step = [&](int it, auto curc) {
    constexpr int cur = decltype(curc)::value;       // LDS buffer == register set parity
    // loads for step it+2 go to register set `cur` (its previous content, step it, is already in LDS)
    if (it + 2 < nk) load(it + 2, curc);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 af = As[cur][fa + q * 256];
      const float4 bf = Bs[cur][fb + q * 128];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.w, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.z, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[1][1], 0, 0, 0);
    }
    // step it+1's data sits in register set cur^1 (loaded a whole step ago): to LDS buffer cur^1
    if (it + 1 < nk) store(cur ^ 1, std::integral_constant<int, cur ^ 1>{});
    __syncthreads();
  };
  for (int it = 0; it < nk; it += 2) {
    step(it, I0{});
    if (it + 1 < nk) step(it + 1, I1{});
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

template <int VAR>
__global__ __launch_bounds__(NT, 2) void conv_v1(const Args a) {
  __shared__ float4 As[2][BK * BM / 4];
  __shared__ float4 Bs[2][BK * BN / 4];
  const unsigned long long c_beg = clock64(), w_beg = wall_clock64();
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
    if (more && VAR != 5 && VAR != 6) {
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
    if (more && VAR != 5 && VAR != 6) store(cur ^ 1);
    if (VAR == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (VAR != 6) __syncthreads();
  }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      *reinterpret_cast<float2*>(&yb[(long)m * a.T + t0 + wn * 64 + 2 * li]) = make_float2(acc[mi][0][r], acc[mi][1][r]);
    }
  if (tid == 0 && a.clk) { a.clk[2 * blockIdx.x] = clock64() - c_beg; a.clk[2 * blockIdx.x + 1] = wall_clock64() - w_beg; }
}


// ------------------------------------------------------------------------------------------------
// V9: one persistent workgroup per CU, tile 256 rows x 240 columns on v_mfma_f32_16x16x4_f32.
// B*T = 122 880 columns = 256 CUs x 2 tiles x 240: every CU does exactly the same work (the
// 256 x 128 tiling needs 1.875 residency rounds).  8 waves, wave w owns rows [32w, 32w+32) = 2 row
// tiles x 15 column tiles = 30 accumulators of 4 registers.  A image per K step (fragment order, one
// ds_read_b128 = 4 k4-steps of one row tile): ((((w*2+mt)*4 + kg)*16 + ml)*4 + s) <-> m = 32w+16mt+ml,
// k = 4s+kg.  B image: natural [k][240].  PIPE: the next tile's first K step is loaded during the
// epilogue of the current one.
// ------------------------------------------------------------------------------------------------
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int BN9 = 240;
template <bool PIPE>
__global__ __launch_bounds__(NT, 1) void conv_v9(const Args a) {
  __shared__ float4 As[2][BK * BM / 4];
  __shared__ float Bs[2][BK][BN9];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ml = lane & 15, kg = lane >> 4;
  const int ksteps = a.Cin / BK;
  const int nk = 2 * ksteps;
  const int tiles_per_b = a.T / BN9;
  const int ntiles = a.B * tiles_per_b;
  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int tile_lo = blockIdx.x * per, tile_hi = min(ntiles, tile_lo + per);
  float4 ra0, ra1, rb0, rb1;
  const int i0 = tid, i1 = tid + NT;                 // B staging: float4 index in [0, 960)
  const int r0 = i0 / 60, c0 = (i0 % 60) * 4, r1 = i1 / 60, c1 = (i1 % 60) * 4;
  const bool ok1 = i1 < BK * BN9 / 4;
  auto load = [&](int tile, int it) {
    const int b = tile / tiles_per_b, t0 = (tile % tiles_per_b) * BN9;
    const float* xb = a.x + (long)b * a.Cin * a.T;
    const int tap = it / ksteps, ks = it % ksteps;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + ((long)(tap * ksteps + ks)) * (BK * BM / 4);
    ra0 = wp[tid];
    ra1 = wp[tid + NT];
    const int tw = t0 - (1 - tap) * a.dil;
    rb0 = make_float4(0.f, 0.f, 0.f, 0.f); rb1 = rb0;
    if (tw >= 0) {
      rb0 = *reinterpret_cast<const float4*>(xb + (long)(ks * BK + r0) * a.T + tw + c0);
      if (ok1) rb1 = *reinterpret_cast<const float4*>(xb + (long)(ks * BK + r1) * a.T + tw + c1);
    }
  };
  auto store = [&](int buf) {
    As[buf][tid] = ra0;
    As[buf][tid + NT] = ra1;
    *reinterpret_cast<float4*>(&Bs[buf][r0][c0]) = rb0;
    if (ok1) *reinterpret_cast<float4*>(&Bs[buf][r1][c1]) = rb1;
  };
  const int fa = ((wave * 2) * 4 + kg) * 16 + ml;      // float4 index of row tile 0; row tile 1 is +64
  if (tile_lo < tile_hi) { load(tile_lo, 0); store(0); }
  __syncthreads();
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    f32x4 acc[2][15];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 15; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < nk; ++it) {
      const int cur = it & 1;
      const bool more = it + 1 < nk;
      const bool next_tile = PIPE && !more && tile + 1 < tile_hi;
      if (more) load(tile, it + 1);
      else if (next_tile) load(tile + 1, 0);
      const float4 a0 = As[cur][fa], a1 = As[cur][fa + 64];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const float av0 = s4 == 0 ? a0.x : s4 == 1 ? a0.y : s4 == 2 ? a0.z : a0.w;
        const float av1 = s4 == 0 ? a1.x : s4 == 1 ? a1.y : s4 == 2 ? a1.z : a1.w;
        float bv[15];
#pragma unroll
        for (int j = 0; j < 15; ++j) bv[j] = Bs[cur][4 * s4 + kg][16 * j + ml];
#pragma unroll
        for (int j = 0; j < 15; ++j) {
          acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bv[j], acc[0][j], 0, 0, 0);
          acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bv[j], acc[1][j], 0, 0, 0);
        }
      }
      if (more || next_tile) store(cur ^ 1);          // nk is even: the next tile starts in buffer 0 again
      __syncthreads();
    }
    {
      const int b = tile / tiles_per_b, t0 = (tile % tiles_per_b) * BN9;
      float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* row = yb + (long)(wave * 32 + mt * 16 + 4 * kg + r) * a.T + t0 + ml;
#pragma unroll
          for (int j = 0; j < 15; ++j) row[16 * j] = acc[mt][j][r];
        }
    }
    if (!PIPE && tile + 1 < tile_hi) { load(tile + 1, 0); store(0); __syncthreads(); }
  }
}

// ------------------------------------------------------------------------------------------------
// V11: the V1 loop as a persistent kernel: 512 workgroups (2 per CU) walk the 960 tiles with stride
// gridDim.x; the first K step of the next tile is fetched during the last K step of the current one
// and the epilogue's stores drain behind the next tile's MFMAs, so no CU ever sits in a prologue or
// epilogue with nothing else to do.  GLDS: A staged by global_load_lds.
// ------------------------------------------------------------------------------------------------
template <bool GLDS>
__global__ __launch_bounds__(NT, 4) void conv_v11(const Args a) {
  __shared__ float4 As[2][BK * BM / 4];
  __shared__ float4 Bs[2][BK * BN / 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int ksteps = a.Cin / BK;
  const int nk = 2 * ksteps;
  const int ntiles = a.B * a.ntile_n;
  const int v_k = tid >> 5, v_col = (tid & 31) * 4;
  const int b_kk2 = v_k >> 2, b_j = (v_k >> 1) & 1, b_lk = v_k & 1;
  const int b_wn = v_col >> 6, b_p = (v_col & 63) >> 1;
  const int b_dst = ((((b_kk2 * 2 + b_lk) * 2 + b_wn) * 32 + b_p) * 4 + b_j * 2) / 2;
  float4 ra0, ra1, rb0;
  auto load = [&](int tile, int it, int buf) {
    const int b = tile / a.ntile_n, t0 = (tile % a.ntile_n) * BN;
    const float* xb = a.x + (long)b * a.Cin * a.T;
    const int tap = it / ksteps, ks = it % ksteps;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + ((long)(tap * ksteps + ks)) * (BK * BM / 4);
    if (GLDS) {
      __builtin_amdgcn_global_load_lds(wp + tid, (__attribute__((address_space(3))) void*)&As[buf][wave * 64], 16, 0, 0);
      __builtin_amdgcn_global_load_lds(wp + tid + NT, (__attribute__((address_space(3))) void*)&As[buf][NT + wave * 64], 16, 0, 0);
    } else {
      ra0 = wp[tid];
      ra1 = wp[tid + NT];
    }
    const int tw = t0 - (1 - tap) * a.dil;
    rb0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tw >= 0) rb0 = *reinterpret_cast<const float4*>(xb + (long)(ks * BK + v_k) * a.T + tw + v_col);
  };
  auto store = [&](int buf) {
    if (!GLDS) {
      As[buf][tid] = ra0;
      As[buf][tid + NT] = ra1;
    }
    float2* bd = reinterpret_cast<float2*>(&Bs[buf][0]);
    bd[b_dst] = make_float2(rb0.x, rb0.y);
    bd[b_dst + 2] = make_float2(rb0.z, rb0.w);
  };
  const int fa = ((lk * 4 + wm) * 32 + li);
  const int fb = ((lk * 2 + wn) * 32 + li);
  int tile = blockIdx.x;
  if (tile < ntiles) { load(tile, 0, 0); store(0); }
  if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nxt = tile + gridDim.x;
    for (int it = 0; it < nk; ++it) {
      const int cur = it & 1;
      const bool more = it + 1 < nk;
      const bool fetch = more || nxt < ntiles;
      if (fetch) load(more ? tile : nxt, more ? it + 1 : 0, cur ^ 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 af = As[cur][fa + q * 256];
        const float4 bf = Bs[cur][fb + q * 128];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.y, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.w, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.z, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[1][1], 0, 0, 0);
      }
      if (fetch) store(cur ^ 1);
      if (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const int b = tile / a.ntile_n, t0 = (tile % a.ntile_n) * BN;
    float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        *reinterpret_cast<float2*>(&yb[(long)m * a.T + t0 + wn * 64 + 2 * li]) = make_float2(acc[mi][0][r], acc[mi][1][r]);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// V12: both operands staged by LDS-DMA (global_load_lds_dwordx4) into a 3-stage LDS ring, loads
// issued TWO K steps ahead, counted vmcnt + raw s_barrier (one per K step), no staging VGPRs and no
// ds_write at all.  8 waves along M (32 rows each); lane li owns columns 4li..4li+3 of the 128-wide
// tile, so the B image is the natural [k][128] row-major tile (what LDS-DMA writes) and one
// ds_read_b128 feeds 4 MFMAs; the A image is fragment-ordered by the pack kernel:
//   float4 index ((kq*2 + lk)*8 + w)*32 + li, component c  <->  m = 32w+li, k = 2*(4kq+c)+lk.
// Epilogue: one 16-byte store per accumulator row (512 B contiguous per half wave).
// ------------------------------------------------------------------------------------------------
constexpr int STAGE_F4 = (BK * BM + BK * BN) / 4;      // float4 per stage: 1024 (A) + 512 (B)
// LDS-DMA hidden from hipcc's wait bookkeeping (it would drain vmcnt(0) before every ds_read of the
// same array): M0 = wave-uniform LDS byte address, every lane supplies its own 16-byte source.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int NSTAGE>
__global__ __launch_bounds__(NT, 4) void conv_v12(const Args a) {
  extern __shared__ float4 lds[];
  int b, t0; tile_of(a, b, t0);
  if ((a.mode & 32) && blockIdx.x >= 256 && blockIdx.x < 512) {      // stagger the second workgroup of every CU by ~half a tile
    const unsigned long long w0 = wall_clock64();
    while (wall_clock64() - w0 < (unsigned long long)a.stagger) __builtin_amdgcn_s_sleep(32);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int ksteps = a.Cin / BK;
  const int nk = 2 * ksteps;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  const int brow = 2 * wave + lk, bcol = 4 * li;       // this lane's piece of the B tile
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)lds;
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int it, int stage) {
    const int tap = it / ksteps, ks = it % ksteps;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + ((long)(tap * ksteps + ks)) * (BK * BM / 4);
    const unsigned st = lds_base + (unsigned)(stage * STAGE_F4 + uwave * 64) * 16u;     // wave-uniform byte address
    if (!(a.mode & 1)) {
      glds16(wp + wave * 64 + lane, st);
      glds16(wp + (8 + wave) * 64 + lane, st + 8 * 64 * 16);
    }
    const int tw = t0 - (1 - tap) * a.dil;             // the harness only times tiles with tw >= 0 correctly
    const float* src = ((a.mode & 4) ? a.x : xb) + (long)(ks * BK + brow) * a.T + ((tw >= 0 && !(a.mode & 4)) ? tw : 0) + bcol;
    if (!(a.mode & 2)) glds16(src, st + 1024 * 16);
  };
  issue(0, 0);
  issue(1, 1);
  if ((a.mode & 31) == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int fa = (lk * 8 + wave) * 32 + li;            // + kq*512
  const int fb = 1024 + lk * 32 + li;                  // + kk*64
  int stage = 0;
  for (int it = 0; it < nk; ++it) {
    const bool more = it + 2 < nk;
    int s2 = stage + 2; if (s2 >= NSTAGE) s2 -= NSTAGE;
    if (more) issue(it + 2, s2);
    const float4* st = lds + stage * STAGE_F4;
    float4 a0, a1;
    if (a.mode & 16) { a0 = make_float4(1.f, 2.f, 3.f, 4.f); a1 = a0; } else { a0 = st[fa]; a1 = st[fa + 512]; }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float4 bf;
      if (a.mode & 16) bf = make_float4(0.5f, 0.25f, 0.125f, 1.f); else bf = st[fb + kk * 64];
      const float av = kk == 0 ? a0.x : kk == 1 ? a0.y : kk == 2 ? a0.z : kk == 3 ? a0.w : kk == 4 ? a1.x : kk == 5 ? a1.y : kk == 6 ? a1.z : a1.w;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf.w, acc[3], 0, 0, 0);
    }
    // step it+1's loads (issued one step ago) must have landed; the 3 just issued may stay in flight
    if (more && (a.mode & 31) == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (diagnostic modes issue fewer loads: drain)
    if (!(a.mode & 8)) __builtin_amdgcn_s_barrier();
    if (++stage >= NSTAGE) stage = 0;
  }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
    *reinterpret_cast<float4*>(&yb[(long)m * a.T + t0 + 4 * li]) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
  }
}

// ------------------------------------------------------------------------------------------------
// V13: ONE workgroup of 4 waves per CU (one wave per SIMD, 512 registers each), tile 256 x 256, wave
// tile 128 x 128 = 16 accumulators: 16 MFMAs per pair of ds_read_b128, 16 KB of operands per MFLOP
// (24 in the 256 x 128 tiling).  3-stage LDS-DMA ring; the fragments of the next k-pair -- and, at
// the end of a K step, of the NEXT stage -- are read before the current MFMAs are issued, so the
// single wave of a SIMD never waits on LDS and the per-step barrier only covers write-after-read.
//   A image per K step: float4 ((kk*2+lk)*2 + wm)*32 + li, component i <-> m = 128wm + 32i + li, k = 2kk+lk
//   B image per K step: natural [16 k][256 columns]; lane li owns columns 128wn + 4li .. +3
// ------------------------------------------------------------------------------------------------
constexpr int BN13 = 256, NT13 = 256, STAGE13_F4 = (BK * BM + BK * BN13) / 4;     // 2048 float4 = 32 KB
__global__ __launch_bounds__(NT13, 1) void conv_v13(const Args a) {
  extern __shared__ float4 lds[];
  const int tiles_per_b = a.T / BN13;
  int logical;
  {
    const int nblk = gridDim.x, id = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int b = logical / tiles_per_b, t0 = (logical % tiles_per_b) * BN13;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ksteps = a.Cin / BK;
  const int nk = 2 * ksteps;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)lds;
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int it, int stage) {
    const int tap = it / ksteps, ks = it % ksteps;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + ((long)(tap * ksteps + ks)) * (BK * BM / 4);
    const unsigned st = lds_base + (unsigned)(stage * STAGE13_F4) * 16u;
#pragma unroll
    for (int i = 0; i < 4; ++i)                            // A: 16 pieces of 1 KB, 4 per wave
      glds16(wp + (i * 4 + wave) * 64 + lane, st + (unsigned)((i * 4 + uwave) * 64) * 16u);
    const int tw = t0 - (1 - tap) * a.dil;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                          // B: 16 rows of 1 KB, 4 per wave
      const int row = i * 4 + wave;
      const float* src = xb + (long)(ks * BK + row) * a.T + (tw >= 0 ? tw : 0) + 4 * lane;
      glds16(src, st + (unsigned)(1024 + (i * 4 + uwave) * 64) * 16u);
    }
  };
  issue(0, 0);
  issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const int fa = (lk * 2 + wm) * 32 + li;                  // + kk*128
  const int fb = 1024 + lk * 64 + wn * 32 + li;            // + kk*128
  float4 ca = lds[fa], cb = lds[fb];                        // fragments of (stage 0, kk 0)
  int stage = 0;
  for (int it = 0; it < nk; ++it) {
    int s1 = stage + 1; if (s1 >= 3) s1 -= 3;
    int s2 = stage + 2; if (s2 >= 3) s2 -= 3;
    if (it + 2 < nk) issue(it + 2, s2);
    const float4* st = lds + stage * STAGE13_F4;
    const float4* sn = lds + s1 * STAGE13_F4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float4 na, nb;
      if (kk < 7) { na = st[fa + (kk + 1) * 128]; nb = st[fb + (kk + 1) * 128]; }
      else { na = sn[fa]; nb = sn[fb]; }                   // first fragments of the next K step (junk after the last)
      const float av[4] = {ca.x, ca.y, ca.z, ca.w};
      const float bv[4] = {cb.x, cb.y, cb.z, cb.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      ca = na; cb = nb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage it+2 has landed (issued a whole K step ago)
    __builtin_amdgcn_s_barrier();                          // ... for everyone; and stage `stage` is free to be refilled
    stage = s1;
  }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      *reinterpret_cast<float4*>(&yb[(long)m * a.T + t0 + wn * 128 + 4 * li]) = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
    }
}

// ------------------------------------------------------------------------------------------------
// V14: V13's in-wave pipelining with TWO independent workgroups per CU: 4 waves each (2 per SIMD,
// 256 registers), tile 256 x 128, wave tile 128 x 64 = 8 accumulators; lane li owns columns
// 64wn + 2li, +1 (one ds_read_b64 per k-pair), A fragment as in V13.  3-stage ring of 24 KB.
// ------------------------------------------------------------------------------------------------
constexpr int STAGE14_F4 = (BK * BM + BK * BN) / 4;      // 1536 float4 = 24 KB
__global__ __launch_bounds__(256, 2) void conv_v14(const Args a) {
  extern __shared__ float4 lds[];
  int b, t0; tile_of(a, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ksteps = a.Cin / BK;
  const int nk = 2 * ksteps;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)lds;
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int it, int stage) {
    const int tap = it / ksteps, ks = it % ksteps;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + ((long)(tap * ksteps + ks)) * (BK * BM / 4);
    const unsigned st = lds_base + (unsigned)(stage * STAGE14_F4) * 16u;
#pragma unroll
    for (int i = 0; i < 4; ++i)                            // A: 16 KB, 4 pieces per wave
      glds16(wp + (i * 4 + wave) * 64 + lane, st + (unsigned)((i * 4 + uwave) * 64) * 16u);
    const int tw = t0 - (1 - tap) * a.dil;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                          // B: 16 rows of 512 B = 8 pieces, 2 per wave (2 rows each)
      const int row = (i * 4 + wave) * 2 + lk;
      const float* src = xb + (long)(ks * BK + row) * a.T + (tw >= 0 ? tw : 0) + 4 * li;
      glds16(src, st + (unsigned)(1024 + (i * 4 + uwave) * 64) * 16u);
    }
  };
  issue(0, 0);
  issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const float2* lds2 = reinterpret_cast<const float2*>(lds);
  const int fa = (lk * 2 + wm) * 32 + li;                  // float4 index, + kk*128
  const int fb = 2 * 1024 + lk * 64 + wn * 32 + li;        // float2 index, + kk*128
  float4 ca = lds[fa];
  float2 cb = lds2[fb];
  int stage = 0;
  for (int it = 0; it < nk; ++it) {
    int s1 = stage + 1; if (s1 >= 3) s1 -= 3;
    int s2 = stage + 2; if (s2 >= 3) s2 -= 3;
    if (it + 2 < nk) issue(it + 2, s2);
    const float4* st = lds + stage * STAGE14_F4;
    const float4* sn = lds + s1 * STAGE14_F4;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      float4 na; float2 nb;
      if (kk < 7) { na = st[fa + (kk + 1) * 128]; nb = reinterpret_cast<const float2*>(st)[fb + (kk + 1) * 128]; }
      else { na = sn[fa]; nb = reinterpret_cast<const float2*>(sn)[fb]; }
      const float av[4] = {ca.x, ca.y, ca.z, ca.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], cb.x, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], cb.y, acc[i][1], 0, 0, 0);
      }
      ca = na; cb = nb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage = s1;
  }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      *reinterpret_cast<float2*>(&yb[(long)m * a.T + t0 + wn * 64 + 2 * li]) = make_float2(acc[i][0][r], acc[i][1][r]);
    }
}

// ------------------------------------------------------------------------------------------------
// V15: V12 with a dedicated LOADER wave: 9 waves per workgroup, waves 0-7 only ds_read + MFMA, wave 8
// issues every LDS-DMA piece of the K step (16 A + 8 B) and is the only wave with VMEM instructions
// in the loop -- tests whether the load cost is VMEM issue inside the MFMA waves' instruction streams.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(576, 5) void conv_v15(const Args a) {
  extern __shared__ float4 lds[];
  int b, t0; tile_of(a, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const bool loader = __builtin_amdgcn_readfirstlane(wave) == 8;
  const int ksteps = a.Cin / BK;
  const int nk = 2 * ksteps;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float4*)lds;
  auto issue = [&](int it, int stage) {                    // loader wave only: the whole stage
    const int tap = it / ksteps, ks = it % ksteps;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + ((long)(tap * ksteps + ks)) * (BK * BM / 4);
    const unsigned st = lds_base + (unsigned)(stage * STAGE_F4) * 16u;
    const float4* ap = wp + lane;
#pragma unroll 1
    for (int i = 0; i < 16; ++i) { glds16(ap, st + (unsigned)(i * 64) * 16u); ap += 64; }
    const int tw = t0 - (1 - tap) * a.dil;
    const float* src = xb + (long)(ks * BK + lk) * a.T + (tw >= 0 ? tw : 0) + 4 * li;
#pragma unroll 1
    for (int i = 0; i < 8; ++i) { glds16(src, st + (unsigned)(1024 + i * 64) * 16u); src += 2L * a.T; }
  };
  if (loader) {
    issue(0, 0); issue(1, 1);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int stage = 0;
    for (int it = 0; it < nk; ++it) {
      int s2 = stage + 2; if (s2 >= 3) s2 -= 3;
      if (it + 2 < nk) { issue(it + 2, s2); asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (++stage >= 3) stage = 0;
    }
    return;
  }
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  __builtin_amdgcn_s_barrier();
  const int fa = (lk * 8 + wave) * 32 + li;
  const int fb = 1024 + lk * 32 + li;
  int stage = 0;
  for (int it = 0; it < nk; ++it) {
    const float4* st = lds + stage * STAGE_F4;
    const float4 a0 = st[fa], a1 = st[fa + 512];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float4 bf = st[fb + kk * 64];
      const float av = kk == 0 ? a0.x : kk == 1 ? a0.y : kk == 2 ? a0.z : kk == 3 ? a0.w : kk == 4 ? a1.x : kk == 5 ? a1.y : kk == 6 ? a1.z : a1.w;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf.y, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf.z, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf.w, acc[3], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
    if (++stage >= 3) stage = 0;
  }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
    *reinterpret_cast<float4*>(&yb[(long)m * a.T + t0 + 4 * li]) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
  }
}

// ------------------------------------------------------------------------------------------------
// V16: V11 again (persistent workgroups, next tile's first K step fetched during the last K step of
// the current tile, stores draining behind the next tile's MFMAs) with a register-lean epilogue:
// buffer-resource stores (SGPR base + one VGPR lane offset + SGPR row offset) instead of 32 64-bit
// store addresses per lane, so the kernel fits 128 VGPRs without spilling.
// ------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc16_t;
__global__ __launch_bounds__(NT, 4) void conv_v16(const Args a) {
  __shared__ float4 As[2][BK * BM / 4];
  __shared__ float4 Bs[2][BK * BN / 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int ksteps = a.Cin / BK;
  const int nk = 2 * ksteps;
  const int ntiles = a.B * a.ntile_n;
  const int v_k = tid >> 5, v_col = (tid & 31) * 4;
  const int b_kk2 = v_k >> 2, b_j = (v_k >> 1) & 1, b_lk = v_k & 1;
  const int b_wn = v_col >> 6, b_p = (v_col & 63) >> 1;
  const int b_dst = ((((b_kk2 * 2 + b_lk) * 2 + b_wn) * 32 + b_p) * 4 + b_j * 2) / 2;
  float4 ra0, ra1, rb0;
  auto load = [&](int tile, int it) {
    const int b = tile / a.ntile_n, t0 = (tile % a.ntile_n) * BN;
    const float* xb = a.x + (long)b * a.Cin * a.T;
    const int tap = it / ksteps, ks = it % ksteps;
    const float4* wp = reinterpret_cast<const float4*>(a.wpk) + ((long)(tap * ksteps + ks)) * (BK * BM / 4);
    ra0 = wp[tid];
    ra1 = wp[tid + NT];
    const int tw = t0 - (1 - tap) * a.dil;
    rb0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tw >= 0) rb0 = *reinterpret_cast<const float4*>(xb + (long)(ks * BK + v_k) * a.T + tw + v_col);
  };
  auto store = [&](int buf) {
    As[buf][tid] = ra0;
    As[buf][tid + NT] = ra1;
    float2* bd = reinterpret_cast<float2*>(&Bs[buf][0]);
    bd[b_dst] = make_float2(rb0.x, rb0.y);
    bd[b_dst + 2] = make_float2(rb0.z, rb0.w);
  };
  const int fa = ((lk * 4 + wm) * 32 + li);
  const int fb = ((lk * 2 + wn) * 32 + li);
  // lane part of every store address: row 4*lk (+ the r-dependent rows through the SGPR offset), column pair 2*li
  const unsigned voff = 4u * (unsigned)((wm * 64 + 4 * lk) * a.T + wn * 64 + 2 * li);
  int tile = blockIdx.x;
  if (tile < ntiles) { load(tile, 0); store(0); }
  __syncthreads();
  for (; tile < ntiles; tile += gridDim.x) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nxt = tile + gridDim.x;
    for (int it = 0; it < nk; ++it) {
      const int cur = it & 1;
      const bool more = it + 1 < nk;
      const bool fetch = more || nxt < ntiles;
      if (fetch) load(more ? tile : nxt, more ? it + 1 : 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 af = As[cur][fa + q * 256];
        const float4 bf = Bs[cur][fb + q * 128];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.y, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.x, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[1][1], 0, 0, 0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.w, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.z, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[1][1], 0, 0, 0);
      }
      if (fetch) store(cur ^ 1);
      __syncthreads();
    }
    const int b = __builtin_amdgcn_readfirstlane(tile / a.ntile_n);
    const int t0 = __builtin_amdgcn_readfirstlane((tile % a.ntile_n) * BN);
    const rsrc16_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (long)b * BM * a.T + t0, 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned soff = 4u * (unsigned)((mi * 32 + (r & 3) + 8 * (r >> 2)) * a.T);
        const float2 v = make_float2(acc[mi][0][r], acc[mi][1][r]);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned, v), ry, voff, soff, 0);
      }
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

static void pack_v9(const std::vector<float>& W, int Cin, std::vector<float>& out) {
  out.assign((size_t)2 * Cin * BM, 0.f);
  const int ksteps = Cin / BK;
  for (int tap = 0; tap < 2; ++tap)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int w = 0; w < 8; ++w) for (int mt = 0; mt < 2; ++mt) for (int kg = 0; kg < 4; ++kg)
        for (int ml = 0; ml < 16; ++ml) for (int s4 = 0; s4 < 4; ++s4) {
          const int k = ks * BK + 4 * s4 + kg, m = 32 * w + 16 * mt + ml;
          const size_t idx = ((size_t)(tap * ksteps + ks)) * (BK * BM) + (((((w * 2 + mt) * 4 + kg) * 16 + ml) * 4) + s4);
          out[idx] = W[((size_t)m * Cin + k) * 2 + tap];
        }
}

static void pack_v12(const std::vector<float>& W, int Cin, std::vector<float>& out) {
  out.assign((size_t)2 * Cin * BM, 0.f);
  const int ksteps = Cin / BK;
  for (int tap = 0; tap < 2; ++tap)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int kq = 0; kq < 2; ++kq) for (int lk = 0; lk < 2; ++lk) for (int w = 0; w < 8; ++w)
        for (int li = 0; li < 32; ++li) for (int c = 0; c < 4; ++c) {
          const int k = ks * BK + 2 * (4 * kq + c) + lk, m = 32 * w + li;
          const size_t idx = ((size_t)(tap * ksteps + ks)) * (BK * BM) + ((((kq * 2 + lk) * 8 + w) * 32 + li) * 4 + c);
          out[idx] = W[((size_t)m * Cin + k) * 2 + tap];
        }
}

static void pack_v13(const std::vector<float>& W, int Cin, std::vector<float>& out) {
  out.assign((size_t)2 * Cin * BM, 0.f);
  const int ksteps = Cin / BK;
  for (int tap = 0; tap < 2; ++tap)
    for (int ks = 0; ks < ksteps; ++ks)
      for (int kk = 0; kk < 8; ++kk) for (int lk = 0; lk < 2; ++lk) for (int wm = 0; wm < 2; ++wm)
        for (int li = 0; li < 32; ++li) for (int i = 0; i < 4; ++i) {
          const int k = ks * BK + 2 * kk + lk, m = 128 * wm + 32 * i + li;
          const size_t idx = ((size_t)(tap * ksteps + ks)) * (BK * BM) + ((((kk * 2 + lk) * 2 + wm) * 32 + li) * 4 + i);
          out[idx] = W[((size_t)m * Cin + k) * 2 + tap];
        }
}

int main(int argc, char** argv) {
  const int B = 16, Cin = 256;
  const int dil = argc > 1 ? atoi(argv[1]) : 64;
  const int T = argc > 2 ? atoi(argv[2]) : 7680;      // 8192 -> 1024 tiles = exactly two residency rounds
  const int reps = 20;
  std::vector<float> hx((size_t)B * Cin * T), hW((size_t)BM * Cin * 2);
  srand(1);
  for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hW) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  float *dx, *dw0, *dw1, *dy;
  CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dy, (size_t)B * BM * T * 4));
  CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> p0, p1, p9; pack_v0(hW, Cin, p0); pack_v1(hW, Cin, p1); pack_v9(hW, Cin, p9);
  std::vector<float> p13; pack_v13(hW, Cin, p13);
  float* dw13; CHECK(hipMalloc(&dw13, p13.size() * 4));
  CHECK(hipMemcpy(dw13, p13.data(), p13.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> p12; pack_v12(hW, Cin, p12);
  float* dw12; CHECK(hipMalloc(&dw12, p12.size() * 4));
  CHECK(hipMemcpy(dw12, p12.data(), p12.size() * 4, hipMemcpyHostToDevice));
  float* dw9; CHECK(hipMalloc(&dw9, p9.size() * 4));
  CHECK(hipMemcpy(dw9, p9.data(), p9.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&dw0, p0.size() * 4)); CHECK(hipMalloc(&dw1, p1.size() * 4));
  CHECK(hipMemcpy(dw0, p0.data(), p0.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw1, p1.data(), p1.size() * 4, hipMemcpyHostToDevice));
  unsigned long long* dclk; CHECK(hipMalloc(&dclk, 16 * 4096));
  Args a; a.mode = 0; a.stagger = 0; a.clk = dclk; a.x = dx; a.y = dy; a.Cin = Cin; a.T = T; a.B = B; a.dil = dil; a.ntile_n = T / BN;
  const int grid = B * (T / BN);
  const double flop = 2.0 * B * T * BM * Cin * 2;
  std::vector<float> hy((size_t)B * BM * T);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto check = [&](const char* name) {
    CHECK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int s = 0; s < 200; ++s) {
      const int b = rand() % B, m = rand() % BM, t = dil + 128 + rand() % (T - dil - 128);   // the harness zero-pads whole tiles
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
  int grid_override = 0, nt_override = 0;
  size_t dyn_lds = 0;
  auto run = [&](const char* name, void (*kern)(const Args), const float* w) {
    if (dyn_lds) CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
    a.wpk = w;
    const int grid = grid_override ? grid_override : B * (T / BN);
    CHECK(hipMemset(dy, 0, hy.size() * 4));
    CHECK(hipMemset(dclk, 0, 16 * 4096));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nt_override ? nt_override : NT), dyn_lds, 0, a);
    CHECK(hipDeviceSynchronize());
    check(name);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nt_override ? nt_override : NT), dyn_lds, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nt_override ? nt_override : NT), dyn_lds, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(2 * grid);
    CHECK(hipMemcpy(hc.data(), dclk, 16 * grid, hipMemcpyDeviceToHost));
    double cs = 0, ws = 0; for (int i = 0; i < grid; ++i) { cs += hc[2 * i]; ws += hc[2 * i + 1]; }
    printf("%-9s dil %4d: %8.1f us/launch  %6.1f TFLOP/s  (%.3f of 157.3)   block life %.1f us, shader clock %.0f MHz\n", name, dil, 1e3 * ms / reps, flop / (ms / reps) / 1e9, flop / (ms / reps) / 1e9 / 157.3, ws / grid / 100.0, ws > 0 ? cs / ws * 100.0 : 0.0);
  };
  for (int round = 0; round < 2; ++round) {
    run("V0", conv_v0, dw0);
    run("V1", conv_v1<1>, dw1);
    run("V2", conv_v1<2>, dw1);
    run("V3", conv_v1<3>, dw1);
    run("V4", conv_v1<4>, dw1);
    run("V8", conv_v8, dw1);
    if (T % BN13 == 0 && getenv("CONV_LOOP_V13")) {
      dyn_lds = 3 * STAGE13_F4 * 16; grid_override = B * (T / BN13); nt_override = NT13;
      run("V13", conv_v13, dw13);
      grid_override = 0; nt_override = 0;
    }
    dyn_lds = 3 * STAGE_F4 * 16; nt_override = 576;
    run("V15", conv_v15, dw12);
    dyn_lds = 3 * STAGE14_F4 * 16; nt_override = 256;
    run("V14", conv_v14, dw13);
    nt_override = 0;
    dyn_lds = 3 * STAGE_F4 * 16;
    run("V12", conv_v12<3>, dw12);
    a.mode = 1; run("V12*noA", conv_v12<3>, dw12);
    a.mode = 2; run("V12*noB", conv_v12<3>, dw12);
    a.mode = 4; run("V12*B-L2", conv_v12<3>, dw12);
    a.mode = 3; run("V12*none", conv_v12<3>, dw12);
    a.mode = 3 + 8; run("V12*nobar", conv_v12<3>, dw12);
    a.mode = 3 + 16; run("V12*nolds", conv_v12<3>, dw12);
    a.mode = 3 + 8 + 16; run("V12*mfma", conv_v12<3>, dw12);
    a.mode = 32; a.stagger = 3000; run("V12stg30", conv_v12<3>, dw12);
    a.mode = 32; a.stagger = 5500; run("V12stg55", conv_v12<3>, dw12);
    a.mode = 32; a.stagger = 8000; run("V12stg80", conv_v12<3>, dw12);
    a.mode = 0;
    dyn_lds = 0;
    grid_override = 512;
    run("V16", conv_v16, dw1);
    grid_override = 960;
    run("V16x", conv_v16, dw1);       // same code, one tile per workgroup
    grid_override = 512;
    run("V11", conv_v11<false>, dw1);
    run("V11g", conv_v11<true>, dw1);
    grid_override = 1024;
    run("V11x", conv_v11<false>, dw1);      // non-persistent launch of the same code (one tile per block)
    grid_override = 0;
    if (T % BN9 == 0 && false) {
      grid_override = 256;
      run("V9", conv_v9<false>, dw9);
      run("V9p", conv_v9<true>, dw9);
      grid_override = 0;
    }
    run("V5*", conv_v1<5>, dw1);      // * = not a valid convolution: diagnostic upper bounds
    run("V6*", conv_v1<6>, dw1);
  }
  return 0;
}
