// Microbenchmark: the conv-GEMM main loop of tools/ubench/conv_loop.hip (M = 256 rows, K = 2 taps x
// 256 channels, N = 16 x 7680 columns, fp32 tensors in HBM) with the fp32 products computed on the
// bf16 matrix pipe from an exact three-way split of every operand:
//     x = x_h + x_m + x_l   (each a bf16, RNE; the split is exact for every finite fp32 away from
//                            the denormal range: 3 x 8 significand bits + the signs cover 24 bits)
//     a*b ~= a_l*b_h + a_h*b_l + a_m*b_m + a_m*b_h + a_h*b_m + a_h*b_h      (fp32 accumulate)
// The three dropped products (m*l, l*m, l*l) are below 2^-25 of |a*b| -- under the rounding of one
// fp32 multiply -- and each kept product of two bf16 is exact in fp32.  v_mfma_f32_32x32x16_bf16 runs
// at 16x the rate of v_mfma_f32_32x32x2_f32, so six of them per 16 k cost 6/16 of the fp32 MFMA time.
//   X1  256 x 128 tile, 8 waves as 4 x 2 (64 x 64 each: the accumulator layout of csrc/conv_gemm_x3.hip),
//       weights pre-split and packed in fragment order, activations split while they are staged
//   X2  X1 with the fetches two K steps ahead (two register sets): the product kernel's loop
//       (modes: ablations -- no A / B loads, no split, no barrier, no fragment reads; 128: the
//       LDS stores of the next step issued before the MFMAs; 512: float4 stores through LDS)
//   X3  4-wave workgroups, 64 x 128 per wave, two workgroups per CU
//   X5  wave specialisation: 4 MFMA waves + 4 staging waves (d4: fetches four steps ahead)
//   X6  three LDS stages and the fragments double-buffered in registers (spills 340 VGPRs)
//   X7  weights by LDS-DMA into a three-stage ring (no staging VGPRs, no ds_write for 2/3 of the bytes)
//   X8  persistent workgroups (one per CU), the next tile's first two K steps fetched before the epilogue
// Measured on MI355X (T = 7680, dil 64; fp32 MFMA loop of conv_loop.hip: 285 us): X1 241, X2 202-212,
// X3 208-218, X5 225, X5d4 213, X2 stage-first 210 (205 with an MFMA/VALU interleave), X7 207-214,
// X8 207-211 us.
// Ablations of X2: no A loads 162-169, no B loads 147-160, neither 122-124, no fragment reads 180,
// no split VALU 206, MFMA only 107 (at 1.84 GHz: 94 % of the clock-adjusted matrix pipe).
// The chip is at its power limit: cycles per tile do not depend on how many workgroups run (80 k for
// 32 K steps with 32 or with 960 workgroups, X3_GRID), the shader clock does (2.35 GHz -> 1.6-1.7 GHz),
// and a schedule that saves cycles at low load (stage-first: -5.5 %) gives them back in clock at
// full load.  At the power limit a stall is free -- an idle CU's share of the budget clocks the others
// higher -- so hiding latency (X5, X6, X7, X8, deeper prefetch, more workgroups per CU) buys nothing;
// what helps is spending less energy per FLOP: fewer bytes moved (the product's 256-column tiles, the
// tap-interleaved K order that keeps the second tap's fetch in L1/L2).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 x3_loop.hip -o x3_loop ; run: ./x3_loop [dil] [T]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <type_traits>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
constexpr int BM = 256, BN = 128, BK = 16, NT = 512;

struct Args {
  const float* x;        // (B, Cin, T)
  const uint4* wpk;      // [kstep][piece 3][lk 2][m 256] x (8 bf16 = 16 B)
  float* y;              // (B, 256, T)
  int Cin, T, B, dil, ntile_n;
  unsigned long long* clk;   // per block {shader cycles, 100 MHz ticks}
  int mode;   // diagnostics (results invalid): 1 no A loads, 2 no B loads, 4 no split, 8 no barrier, 16 no LDS fragment reads
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

__device__ __forceinline__ unsigned pk(float lo, float hi) {      // RNE, v_cvt_pk_bf16_f32
  bf16x2 v;
  v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = pk(x0, x1);
  float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = pk(r0, r1);
  r0 -= __builtin_bit_cast(float, m << 16); r1 -= __builtin_bit_cast(float, m & 0xffff0000u);
  l = pk(r0, r1);
}

template <int MODE>
__global__ __launch_bounds__(NT, 2) void conv_x1(const Args a) {
  __shared__ uint4 As[2][3][2][BM];      // 24 KB per buffer
  __shared__ uint4 Bs[2][3][2][BN];      // 12 KB per buffer
  int b, t0; tile_of(a, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ksteps_tap = a.Cin / BK, nk = 2 * ksteps_tap;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  const int s_n = tid & 127, s_q = tid >> 7;          // staging role: column s_n, channels 4*s_q .. 4*s_q+3
  uint4 ra0 = make_uint4(tid, 1, 2, 3), ra1 = ra0, ra2 = ra0;
  float rb0 = tid, rb1 = 1.f, rb2 = 2.f, rb3 = 3.f;
  bool rb_ok = true;
  auto load = [&](int it) {
    const uint4* wp = a.wpk + (size_t)it * (3 * 2 * BM) + tid;
    if (!(MODE & 1)) { ra0 = wp[0]; ra1 = wp[NT]; ra2 = wp[2 * NT]; }
    const int tap = it / ksteps_tap, c0 = (it % ksteps_tap) * BK + 4 * s_q;
    const int ts = t0 + s_n - (1 - tap) * a.dil;
    const float* xp = xb + (long)c0 * a.T + ts;
    if (!(MODE & 2)) {
      const float* xs = ts >= 0 ? xp : xb;           // branch-free: always load, then select
      rb0 = xs[0]; rb1 = xs[(long)a.T]; rb2 = xs[2L * a.T]; rb3 = xs[3L * a.T];
      rb_ok = ts >= 0;                               // applied in store(): nothing here may wait on the loads
    }
  };
  auto store = [&](int buf) {
    uint4* ad = &As[buf][0][0][0];
    ad[tid] = ra0; ad[NT + tid] = ra1; ad[2 * NT + tid] = ra2;
    unsigned h0, m0, l0, h1, m1, l1;
    if (MODE & 4) {
      h0 = m0 = l0 = __builtin_bit_cast(unsigned, rb0) ^ __builtin_bit_cast(unsigned, rb1);
      h1 = m1 = l1 = __builtin_bit_cast(unsigned, rb2) ^ __builtin_bit_cast(unsigned, rb3);
    } else {
      if (!rb_ok) rb0 = rb1 = rb2 = rb3 = 0.f;
      split3(rb0, rb1, h0, m0, l0);
      split3(rb2, rb3, h1, m1, l1);
    }
    uint2* bd = reinterpret_cast<uint2*>(&Bs[buf][0][s_q >> 1][s_n]) + (s_q & 1);
    bd[0 * 2 * 2 * BN] = make_uint2(h0, h1);
    bd[1 * 2 * 2 * BN] = make_uint2(m0, m1);
    bd[2 * 2 * 2 * BN] = make_uint2(l0, l1);
  };
  load(0);
  store(0);
  __syncthreads();
  for (int it = 0; it < nk; ++it) {
    const int cur = it & 1;
    const bool more = it + 1 < nk;
    if (more) load(it + 1);
    bf16x8 af[2][3], bf[2][3];
    if (MODE & 16) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) { af[i][p] = __builtin_bit_cast(bf16x8, ra0); bf[i][p] = __builtin_bit_cast(bf16x8, ra1); }
    } else
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        af[i][p] = __builtin_bit_cast(bf16x8, As[cur][p][lk][wm * 64 + i * 32 + li]);
        bf[i][p] = __builtin_bit_cast(bf16x8, Bs[cur][p][lk][wn * 64 + i * 32 + li]);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
    if (more) store(cur ^ 1);
    if (!(MODE & 8)) __syncthreads();
  }
  // C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        yb[(long)row * a.T + t0 + wn * 64 + j * 32 + li] = acc[i][j][r];
      }
}

// X2: X1 with the global loads two K steps ahead (two named register sets, loop unrolled by two so
// that every wait is a counted vmcnt).
template <int MODE>
__global__ __launch_bounds__(NT, 2) void conv_x2(const Args a) {
  __shared__ uint4 As[2][3][2][BM];
  __shared__ uint4 Bs[2][3][2][BN];
  const long long c_beg = clock64(), w_beg = wall_clock64();
  int b, t0; tile_of(a, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ksteps_tap = a.Cin / BK, nk = 2 * ksteps_tap;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  const int s_n = tid & 127, s_q = tid >> 7;
  uint4 pa0 = make_uint4(tid, 1, 2, 3), pa1 = pa0, pa2 = pa0, qa0 = pa0, qa1 = pa0, qa2 = pa0;   // register sets P (even steps), Q (odd)
  float pb0 = tid, pb1 = 1.f, pb2 = 2.f, pb3 = 3.f, qb0 = tid, qb1 = 1.f, qb2 = 2.f, qb3 = 3.f;
  bool pok = true, qok = true;
#define X3_LOAD(...) X3_LOAD_(__VA_ARGS__)
#define X3_STORE(...) X3_STORE_(__VA_ARGS__)
#define X3_LOAD_(a0, a1, a2, b0, b1, b2, b3, ok, it_)                                   \
  {                                                                                     \
    const uint4* wp = a.wpk + (size_t)(it_) * (3 * 2 * BM) + tid;                       \
    if (!(MODE & 1)) { a0 = wp[0]; a1 = wp[NT]; a2 = wp[2 * NT]; }                      \
    const int tap = (it_) / ksteps_tap, c0 = ((it_) % ksteps_tap) * BK + 4 * s_q;       \
    const int ts = t0 + s_n - (1 - tap) * a.dil;                                        \
    const float* xs = (MODE & 32) ? a.x + (long)(4 * s_q) * a.T + s_n : (ts >= 0 ? xb + (long)c0 * a.T + ts : xb); \
    if (MODE & 64) { const float4 v = *reinterpret_cast<const float4*>(xb + (long)(c0 >> 2) * a.T + ((t0 + 4 * s_n) & ~3)); b0 = v.x; b1 = v.y; b2 = v.z; b3 = v.w; } \
    else if (!(MODE & 2)) { b0 = xs[0]; b1 = xs[(long)a.T]; b2 = xs[2L * a.T]; b3 = xs[3L * a.T]; } \
    ok = ts >= 0;                                                                       \
  }
#define X3_STORE_(a0, a1, a2, b0, b1, b2, b3, ok, buf)                                   \
  {                                                                                     \
    uint4* ad = &As[buf][0][0][0];                                                      \
    ad[tid] = a0; ad[NT + tid] = a1; ad[2 * NT + tid] = a2;                             \
    unsigned h0, m0, l0, h1, m1, l1;                                                    \
    if (MODE & 4) {                                                                     \
      h0 = m0 = l0 = __builtin_bit_cast(unsigned, b0) ^ __builtin_bit_cast(unsigned, b1); \
      h1 = m1 = l1 = __builtin_bit_cast(unsigned, b2) ^ __builtin_bit_cast(unsigned, b3); \
    } else {                                                                            \
      split3(ok ? b0 : 0.f, ok ? b1 : 0.f, h0, m0, l0);                                 \
      split3(ok ? b2 : 0.f, ok ? b3 : 0.f, h1, m1, l1);                                 \
    }                                                                                   \
    uint2* bd = reinterpret_cast<uint2*>(&Bs[buf][0][s_q >> 1][s_n]) + (s_q & 1);       \
    bd[0 * 2 * 2 * BN] = make_uint2(h0, h1);                                            \
    bd[1 * 2 * 2 * BN] = make_uint2(m0, m1);                                            \
    bd[2 * 2 * 2 * BN] = make_uint2(l0, l1);                                            \
  }
#define P_SET pa0, pa1, pa2, pb0, pb1, pb2, pb3, pok
#define Q_SET qa0, qa1, qa2, qb0, qb1, qb2, qb3, qok
  auto mma = [&](auto curc) {
    constexpr int cur = decltype(curc)::value;
    bf16x8 af[2][3], bf[2][3];
    if (MODE & 16) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) { af[i][p] = __builtin_bit_cast(bf16x8, pa0); bf[i][p] = __builtin_bit_cast(bf16x8, qa0); }
    } else
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        af[i][p] = __builtin_bit_cast(bf16x8, As[cur][p][lk][wm * 64 + i * 32 + li]);
        bf[i][p] = __builtin_bit_cast(bf16x8, Bs[cur][p][lk][wn * 64 + i * 32 + li]);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  X3_LOAD(P_SET, 0);
  X3_LOAD(Q_SET, 1);
  X3_STORE(P_SET, 0);
  if (MODE & 128) X3_LOAD(P_SET, 2);
  __syncthreads();
  // invariant at the top of the pair (it even): LDS buf0 holds step it; r1 holds (in flight) step it+1
  for (int it = 0; it < nk; it += 2) {
    if (MODE & 128) {
      // stage first: the LDS stores of step it+1 are issued at the top of step it (their data was
      // fetched a whole step ago) and drain under the MFMAs instead of in front of the barrier
      X3_STORE(Q_SET, 1);
      X3_LOAD(Q_SET, min(it + 3, nk - 1));
      mma(I0{});
      if (MODE & 256) { _Pragma("unroll") for (int q = 0; q < 24; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); } }
      __syncthreads();
      X3_STORE(P_SET, 0);
      X3_LOAD(P_SET, min(it + 4, nk - 2));
      mma(I1{});
      if (MODE & 256) { _Pragma("unroll") for (int q = 0; q < 24; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); } }
      __syncthreads();
      continue;
    }
    X3_LOAD(P_SET, min(it + 2, nk - 2));      // unconditional (the tail re-loads a valid step): a branch here makes every wait vmcnt(0)
    mma(I0{});
    X3_STORE(Q_SET, 1);                   // step it+1 (nk is even)
    if (!(MODE & 8)) __syncthreads();
    X3_LOAD(Q_SET, min(it + 3, nk - 1));
    mma(I1{});
    X3_STORE(P_SET, 0);
    if (!(MODE & 8)) __syncthreads();
  }
  if (threadIdx.x == 0) { a.clk[2 * blockIdx.x] = clock64() - c_beg; a.clk[2 * blockIdx.x + 1] = wall_clock64() - w_beg; }
  float* yb = a.y + (long)b * BM * a.T;
  if (MODE & 512) {
    // wide stores: the accumulators go through LDS (free after the loop) so that a lane writes 4
    // consecutive columns (16 B) and a wave whole 512-byte row segments, instead of 128-byte ones
    float* stg = reinterpret_cast<float*>(&As[0][0][0][0]);          // [64 rows][132] floats per wave pair... one wave: 64 x 64 block
    float* mine = stg + wave * (64 * 68);                            // 8 waves x 17 KB = 136 KB > LDS: do it in two halves of 32 rows
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * lk) * 68 + j * 32 + li] = acc[i][j][r];
      __syncthreads();
      // 32 rows x 64 columns: 16 lanes per row (float4 each), 4 rows per pass, 8 passes
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int row = ps * 4 + (lane >> 4), c4 = (lane & 15) * 4;
        const float4 v = *reinterpret_cast<const float4*>(&mine[row * 68 + c4]);
        *reinterpret_cast<float4*>(&yb[(long)(wm * 64 + i * 32 + row) * a.T + t0 + wn * 64 + c4]) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        yb[(long)row * a.T + t0 + wn * 64 + j * 32 + li] = acc[i][j][r];
      }
}


// X8: X2 with persistent workgroups (one per CU) walking the tiles; the first two K steps of the NEXT
// tile are fetched before the current tile's epilogue, whose stores then drain under the next loop.
__global__ __launch_bounds__(NT, 2) void conv_x8(const Args a) {
  constexpr int MODE = 0;
  __shared__ uint4 As[2][3][2][BM];
  __shared__ uint4 Bs[2][3][2][BN];
  const long long c_beg = clock64(), w_beg = wall_clock64();
  const int ntiles = a.B * a.ntile_n;
  auto decode = [&](int vb, int& bb, int& tt) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = vb & 7;
    const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    tt = (logical % a.ntile_n) * BN; bb = logical / a.ntile_n;
  };
  int b, t0; decode(blockIdx.x, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ksteps_tap = a.Cin / BK, nk = 2 * ksteps_tap;
  const float* xb = a.x + (long)b * a.Cin * a.T;   // (t0, xb): the tile the fetch macros address
  const int s_n = tid & 127, s_q = tid >> 7;
  uint4 pa0 = make_uint4(tid, 1, 2, 3), pa1 = pa0, pa2 = pa0, qa0 = pa0, qa1 = pa0, qa2 = pa0;   // register sets P (even steps), Q (odd)
  float pb0 = tid, pb1 = 1.f, pb2 = 2.f, pb3 = 3.f, qb0 = tid, qb1 = 1.f, qb2 = 2.f, qb3 = 3.f;
  bool pok = true, qok = true;
#undef X3_LOAD
#undef X3_STORE
#undef X3_LOAD_
#undef X3_STORE_
#undef P_SET
#undef Q_SET
#define X3_LOAD(...) X3_LOAD_(__VA_ARGS__)
#define X3_STORE(...) X3_STORE_(__VA_ARGS__)
#define X3_LOAD_(a0, a1, a2, b0, b1, b2, b3, ok, it_)                                   \
  {                                                                                     \
    const uint4* wp = a.wpk + (size_t)(it_) * (3 * 2 * BM) + tid;                       \
    if (!(MODE & 1)) { a0 = wp[0]; a1 = wp[NT]; a2 = wp[2 * NT]; }                      \
    const int tap = (it_) / ksteps_tap, c0 = ((it_) % ksteps_tap) * BK + 4 * s_q;       \
    const int ts = t0 + s_n - (1 - tap) * a.dil;                                        \
    const float* xs = (MODE & 32) ? a.x + (long)(4 * s_q) * a.T + s_n : (ts >= 0 ? xb + (long)c0 * a.T + ts : xb); \
    if (MODE & 64) { const float4 v = *reinterpret_cast<const float4*>(xb + (long)(c0 >> 2) * a.T + ((t0 + 4 * s_n) & ~3)); b0 = v.x; b1 = v.y; b2 = v.z; b3 = v.w; } \
    else if (!(MODE & 2)) { b0 = xs[0]; b1 = xs[(long)a.T]; b2 = xs[2L * a.T]; b3 = xs[3L * a.T]; } \
    ok = ts >= 0;                                                                       \
  }
#define X3_STORE_(a0, a1, a2, b0, b1, b2, b3, ok, buf)                                   \
  {                                                                                     \
    uint4* ad = &As[buf][0][0][0];                                                      \
    ad[tid] = a0; ad[NT + tid] = a1; ad[2 * NT + tid] = a2;                             \
    unsigned h0, m0, l0, h1, m1, l1;                                                    \
    if (MODE & 4) {                                                                     \
      h0 = m0 = l0 = __builtin_bit_cast(unsigned, b0) ^ __builtin_bit_cast(unsigned, b1); \
      h1 = m1 = l1 = __builtin_bit_cast(unsigned, b2) ^ __builtin_bit_cast(unsigned, b3); \
    } else {                                                                            \
      split3(ok ? b0 : 0.f, ok ? b1 : 0.f, h0, m0, l0);                                 \
      split3(ok ? b2 : 0.f, ok ? b3 : 0.f, h1, m1, l1);                                 \
    }                                                                                   \
    uint2* bd = reinterpret_cast<uint2*>(&Bs[buf][0][s_q >> 1][s_n]) + (s_q & 1);       \
    bd[0 * 2 * 2 * BN] = make_uint2(h0, h1);                                            \
    bd[1 * 2 * 2 * BN] = make_uint2(m0, m1);                                            \
    bd[2 * 2 * 2 * BN] = make_uint2(l0, l1);                                            \
  }
#define P_SET pa0, pa1, pa2, pb0, pb1, pb2, pb3, pok
#define Q_SET qa0, qa1, qa2, qb0, qb1, qb2, qb3, qok
  auto mma = [&](auto curc) {
    constexpr int cur = decltype(curc)::value;
    bf16x8 af[2][3], bf[2][3];
    if (MODE & 16) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) { af[i][p] = __builtin_bit_cast(bf16x8, pa0); bf[i][p] = __builtin_bit_cast(bf16x8, qa0); }
    } else
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        af[i][p] = __builtin_bit_cast(bf16x8, As[cur][p][lk][wm * 64 + i * 32 + li]);
        bf[i][p] = __builtin_bit_cast(bf16x8, Bs[cur][p][lk][wn * 64 + i * 32 + li]);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  X3_LOAD(P_SET, 0);
  X3_LOAD(Q_SET, 1);
  for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    const int cb = b, ct0 = t0;                      // the tile being computed
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    X3_STORE(P_SET, 0);
    __syncthreads();
    for (int it = 0; it < nk; it += 2) {
      X3_LOAD(P_SET, min(it + 2, nk - 2));
      mma(I0{});
      X3_STORE(Q_SET, 1);
      __syncthreads();
      X3_LOAD(Q_SET, min(it + 3, nk - 1));
      mma(I1{});
      if (it + 2 < nk) X3_STORE(P_SET, 0);
      __syncthreads();
    }
    // the next tile's first two steps travel while this tile's results are stored
    if (vb + (int)gridDim.x < ntiles) {
      decode(vb + gridDim.x, b, t0);
      xb = a.x + (long)b * a.Cin * a.T;
    }
    X3_LOAD(P_SET, 0);
    X3_LOAD(Q_SET, 1);
    float* yb = a.y + (long)cb * BM * a.T;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          yb[(long)row * a.T + ct0 + wn * 64 + j * 32 + li] = acc[i][j][r];
        }
  }
  if (threadIdx.x == 0) { a.clk[2 * blockIdx.x] = clock64() - c_beg; a.clk[2 * blockIdx.x + 1] = wall_clock64() - w_beg; }
}

// X3: 4-wave workgroups (one wave per SIMD), 256 x 128 tile, every wave 64 rows x all 128 columns
// (8 accumulator tiles = 128 registers; 18 fragment reads per 48 MFMAs instead of 24), two
// workgroups per CU so that one's staging runs under the other's MFMAs.
template <int MODE>
__global__ __launch_bounds__(256, 2) void conv_x3(const Args a) {
  __shared__ uint4 As[2][3][2][BM];
  __shared__ uint4 Bs[2][3][2][BN];
  const long long c_beg = clock64(), w_beg = wall_clock64();
  int b, t0; tile_of(a, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ksteps_tap = a.Cin / BK, nk = 2 * ksteps_tap;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  const int s_n = tid & 127, s_lk = tid >> 7;          // staging role: column s_n, channels 8*s_lk .. 8*s_lk+7
  uint4 ra0 = make_uint4(tid, 1, 2, 3), ra1 = ra0, ra2 = ra0, ra3 = ra0, ra4 = ra0, ra5 = ra0;
  float rb0 = tid, rb1 = 1, rb2 = 2, rb3 = 3, rb4 = 4, rb5 = 5, rb6 = 6, rb7 = 7;
  bool rok = true;
  auto load = [&](int it) {
    const uint4* wp = a.wpk + (size_t)it * (3 * 2 * BM) + tid;
    if (!(MODE & 1)) { ra0 = wp[0]; ra1 = wp[256]; ra2 = wp[512]; ra3 = wp[768]; ra4 = wp[1024]; ra5 = wp[1280]; }
    const int tap = it / ksteps_tap, c0 = (it % ksteps_tap) * BK + 8 * s_lk;
    const int ts = t0 + s_n - (1 - tap) * a.dil;
    const float* xs = ts >= 0 ? xb + (long)c0 * a.T + ts : xb;
    const long T = a.T;
    if (!(MODE & 2)) {
      rb0 = xs[0]; rb1 = xs[T]; rb2 = xs[2 * T]; rb3 = xs[3 * T];
      rb4 = xs[4 * T]; rb5 = xs[5 * T]; rb6 = xs[6 * T]; rb7 = xs[7 * T];
    }
    rok = ts >= 0;
  };
  auto store = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    uint4* ad = &As[buf][0][0][0];
    ad[tid] = ra0; ad[256 + tid] = ra1; ad[512 + tid] = ra2; ad[768 + tid] = ra3; ad[1024 + tid] = ra4; ad[1280 + tid] = ra5;
    uint4 h, m, l;
    const float z = 0.f;
    split3(rok ? rb0 : z, rok ? rb1 : z, h.x, m.x, l.x);
    split3(rok ? rb2 : z, rok ? rb3 : z, h.y, m.y, l.y);
    split3(rok ? rb4 : z, rok ? rb5 : z, h.z, m.z, l.z);
    split3(rok ? rb6 : z, rok ? rb7 : z, h.w, m.w, l.w);
    Bs[buf][0][s_lk][s_n] = h;
    Bs[buf][1][s_lk][s_n] = m;
    Bs[buf][2][s_lk][s_n] = l;
  };
  auto mma = [&](auto curc) {
    constexpr int cur = decltype(curc)::value;
    bf16x8 af[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) af[i][p] = __builtin_bit_cast(bf16x8, As[cur][p][lk][wave * 64 + i * 32 + li]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 bf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[p] = __builtin_bit_cast(bf16x8, Bs[cur][p][lk][j * 32 + li]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[0], c, 0, 0, 0);
        acc[i][j] = c;
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  load(0);
  store(I0{});
  __syncthreads();
  for (int it = 0; it < nk; it += 2) {
    load(it + 1);
    mma(I0{});
    store(I1{});
    if (!(MODE & 8)) __syncthreads();
    load(min(it + 2, nk - 1));
    mma(I1{});
    store(I0{});
    if (!(MODE & 8)) __syncthreads();
  }
  if (threadIdx.x == 0) { a.clk[2 * blockIdx.x] = clock64() - c_beg; a.clk[2 * blockIdx.x + 1] = wall_clock64() - w_beg; }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        yb[(long)row * a.T + t0 + j * 32 + li] = acc[i][j][r];
      }
}

// X5: wave specialisation.  Waves 0-3 (one per SIMD) only run the matrix pipe: wave w owns rows
// 64w..64w+63 and all 128 columns (8 accumulator tiles), per K step 18 fragment reads and 48 MFMAs.
// Waves 4-7 only stage: fetch (two K steps ahead), split, write LDS.  One barrier per K step, three
// LDS stages.  The split VALU work and the LDS writes never sit in an MFMA wave's instruction stream.
template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_x5(const Args a) {
  constexpr int NS = 3;
  __shared__ uint4 As[NS][3][2][BM];
  __shared__ uint4 Bs[NS][3][2][BN];
  const long long c_beg = clock64(), w_beg = wall_clock64();
  int b, t0; tile_of(a, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int ksteps_tap = a.Cin / BK, nk = 2 * ksteps_tap;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  if (wave >= 4) {
    // ---------------- staging waves ----------------
    const int st = tid - 256;
    const int s_n = st & 127, s_lk = st >> 7;
    uint4 pa0, pa1, pa2, pa3, pa4, pa5, qa0, qa1, qa2, qa3, qa4, qa5;
    float pb0, pb1, pb2, pb3, pb4, pb5, pb6, pb7, qb0, qb1, qb2, qb3, qb4, qb5, qb6, qb7;
    bool pok, qok;
#define X5_LOAD(A0, A1, A2, A3, A4, A5, B0, B1, B2, B3, B4, B5, B6, B7, OK, it_)              \
    {                                                                                          \
      const uint4* wp = a.wpk + (size_t)(it_) * (3 * 2 * BM) + st;                             \
      A0 = wp[0]; A1 = wp[256]; A2 = wp[512]; A3 = wp[768]; A4 = wp[1024]; A5 = wp[1280];      \
      const int tap = (it_) / ksteps_tap, c0 = ((it_) % ksteps_tap) * BK + 8 * s_lk;           \
      const int ts = t0 + s_n - (1 - tap) * a.dil;                                             \
      const float* xs = ts >= 0 ? xb + (long)c0 * a.T + ts : xb;                               \
      const long T = a.T;                                                                      \
      B0 = xs[0]; B1 = xs[T]; B2 = xs[2 * T]; B3 = xs[3 * T];                                  \
      B4 = xs[4 * T]; B5 = xs[5 * T]; B6 = xs[6 * T]; B7 = xs[7 * T];                          \
      OK = ts >= 0;                                                                            \
    }
#define X5_STORE(A0, A1, A2, A3, A4, A5, B0, B1, B2, B3, B4, B5, B6, B7, OK, stage)           \
    {                                                                                          \
      uint4* ad = &As[stage][0][0][0];                                                         \
      ad[st] = A0; ad[256 + st] = A1; ad[512 + st] = A2; ad[768 + st] = A3; ad[1024 + st] = A4; ad[1280 + st] = A5; \
      uint4 h, m, l;                                                                           \
      const float z = 0.f;                                                                     \
      split3(OK ? B0 : z, OK ? B1 : z, h.x, m.x, l.x);                                         \
      split3(OK ? B2 : z, OK ? B3 : z, h.y, m.y, l.y);                                         \
      split3(OK ? B4 : z, OK ? B5 : z, h.z, m.z, l.z);                                         \
      split3(OK ? B6 : z, OK ? B7 : z, h.w, m.w, l.w);                                         \
      Bs[stage][0][s_lk][s_n] = h; Bs[stage][1][s_lk][s_n] = m; Bs[stage][2][s_lk][s_n] = l;  \
    }
#define X5P pa0, pa1, pa2, pa3, pa4, pa5, pb0, pb1, pb2, pb3, pb4, pb5, pb6, pb7, pok
#define X5Q qa0, qa1, qa2, qa3, qa4, qa5, qb0, qb1, qb2, qb3, qb4, qb5, qb6, qb7, qok
#define X5_LOADX(...) X5_LOAD(__VA_ARGS__)
#define X5_STOREX(...) X5_STORE(__VA_ARGS__)
    if (MODE == 0) {
    X5_LOADX(X5P, 0);
    X5_LOADX(X5Q, 1);
    X5_STOREX(X5P, 0);
    __syncthreads();                                   // stage 0 ready
    // step it: consumers read stage it % 3; we write step it + 1 into stage (it + 1) % 3 and fetch it + 2
    int sw = 1;
    for (int it = 0; it < nk; it += 2) {
      X5_LOADX(X5P, min(it + 2, nk - 2));
      X5_STOREX(X5Q, sw);
      sw = sw == 2 ? 0 : sw + 1;
      __syncthreads();
      X5_LOADX(X5Q, min(it + 3, nk - 1));
      X5_STOREX(X5P, sw);
      sw = sw == 2 ? 0 : sw + 1;
      __syncthreads();
    }
    } else {
    // four register sets: a fetch is consumed four steps after it was issued
    uint4 ra0, ra1, ra2, ra3, ra4, ra5, sa0, sa1, sa2, sa3, sa4, sa5;
    float rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7, sb0, sb1, sb2, sb3, sb4, sb5, sb6, sb7;
    bool rok, sok;
#define X5R ra0, ra1, ra2, ra3, ra4, ra5, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7, rok
#define X5S sa0, sa1, sa2, sa3, sa4, sa5, sb0, sb1, sb2, sb3, sb4, sb5, sb6, sb7, sok
    X5_LOADX(X5P, 0);
    X5_LOADX(X5Q, 1);
    X5_LOADX(X5R, 2);
    X5_LOADX(X5S, 3);
    X5_STOREX(X5P, 0);
    __syncthreads();
    int sw = 1;
    for (int it = 0; it < nk; it += 4) {               // nk % 4 == 0 in this harness
      X5_LOADX(X5P, min(it + 4, nk - 4));
      X5_STOREX(X5Q, sw); sw = sw == 2 ? 0 : sw + 1;
      __syncthreads();
      X5_LOADX(X5Q, min(it + 5, nk - 3));
      X5_STOREX(X5R, sw); sw = sw == 2 ? 0 : sw + 1;
      __syncthreads();
      X5_LOADX(X5R, min(it + 6, nk - 2));
      X5_STOREX(X5S, sw); sw = sw == 2 ? 0 : sw + 1;
      __syncthreads();
      X5_LOADX(X5S, min(it + 7, nk - 1));
      X5_STOREX(X5P, sw); sw = sw == 2 ? 0 : sw + 1;
      __syncthreads();
    }
    }
    return;
  }
  // ---------------- matrix waves ----------------
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __syncthreads();
  int sr = 0;
  for (int it = 0; it < nk; ++it) {
    bf16x8 af[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) af[i][p] = __builtin_bit_cast(bf16x8, As[sr][p][lk][wave * 64 + i * 32 + li]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x8 bf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[p] = __builtin_bit_cast(bf16x8, Bs[sr][p][lk][j * 32 + li]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[0], c, 0, 0, 0);
        acc[i][j] = c;
      }
    }
    sr = sr == 2 ? 0 : sr + 1;
    __syncthreads();
  }
  if (lane == 0 && wave == 0) { a.clk[2 * blockIdx.x] = clock64() - c_beg; a.clk[2 * blockIdx.x + 1] = wall_clock64() - w_beg; }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        yb[(long)row * a.T + t0 + j * 32 + li] = acc[i][j][r];
      }
}

// X6: X2 with three LDS stages and the fragments double-buffered in registers.  Within step i a wave
//   (a) splits and stores the data of step i+2 (fetched two steps ago) into stage (i+2)%3,
//   (b) fetches step i+4,
//   (c) reads the fragments of step i+1 from stage (i+1)%3 into the other fragment set,
//   (d) runs the 24 MFMAs of step i on fragments it read during step i-1.
// Nothing in the step waits on something issued in the same step, so the MFMAs can be interleaved
// with everything else and the matrix pipe does not idle through the LDS phases at the step borders
// (X2: ~380 cycles of fragment reads after every barrier and ~450 of LDS stores before it, with all
// eight waves in lockstep: 2500 cycles per step for 1536 of MFMA).
template <int SG>
__global__ __launch_bounds__(NT, 2) void conv_x6(const Args a) {
  __shared__ uint4 As[3][3][2][BM];
  __shared__ uint4 Bs[3][3][2][BN];
  const long long c_beg = clock64(), w_beg = wall_clock64();
  int b, t0; tile_of(a, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ksteps_tap = a.Cin / BK, nk = 2 * ksteps_tap;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  const int s_n = tid & 127, s_q = tid >> 7;
  uint4 pa0, pa1, pa2, qa0, qa1, qa2;
  float pb0, pb1, pb2, pb3, qb0, qb1, qb2, qb3;
  bool pok, qok;
  constexpr int MODE = 0;
  bf16x8 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];      // fragment sets of even / odd steps
#define X6_READ(FA, FB, stage)                                                                 \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                \
    _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                            \
      FA[i][p] = __builtin_bit_cast(bf16x8, As[stage][p][lk][wm * 64 + i * 32 + li]);          \
      FB[i][p] = __builtin_bit_cast(bf16x8, Bs[stage][p][lk][wn * 64 + i * 32 + li]);          \
    }
#define X6_MMA(FA, FB)                                                                         \
  _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                            \
      f32x16 c = acc[i][j];                                                                    \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i][2], FB[j][0], c, 0, 0, 0);             \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i][0], FB[j][2], c, 0, 0, 0);             \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i][1], FB[j][1], c, 0, 0, 0);             \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i][1], FB[j][0], c, 0, 0, 0);             \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i][0], FB[j][1], c, 0, 0, 0);             \
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i][0], FB[j][0], c, 0, 0, 0);             \
      acc[i][j] = c;                                                                           \
    }
#define X6_SCHED()                                                                             \
  if (SG) { _Pragma("unroll") for (int q = 0; q < 24; ++q) {                                   \
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
      __builtin_amdgcn_sched_group_barrier(0x002, SG, 0);                                      \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                       \
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); } }
  // prologue: steps 0 and 1 staged, 2 and 3 in flight, fragments of step 0 in set 0
  X3_LOAD(P_SET, 0);
  X3_LOAD(Q_SET, 1);
  X3_STORE(P_SET, 0);
  X3_STORE(Q_SET, 1);
  X3_LOAD(P_SET, 2);
  X3_LOAD(Q_SET, 3);
  __syncthreads();
  X6_READ(fa0, fb0, 0);
  // steps in groups of six (two fragment sets x three LDS stages); nk % 6 != 0 handled by the bound checks
  int it = 0;
#define X6_STEP(FAc, FBc, FAn, FBn, SETS, st_next, st_store)                                   \
  {                                                                                            \
    X3_STORE(SETS, st_store);                        /* step it + 2 -> stage (it + 2) % 3 */   \
    X3_LOAD(SETS, min(it + 4, nk - 2 + ((it) & 1))); /* step it + 4 */                         \
    X6_READ(FAn, FBn, st_next);                      /* fragments of step it + 1 */            \
    X6_MMA(FAc, FBc);                                                                          \
    X6_SCHED();                                                                                \
    __syncthreads();                                                                           \
    ++it;                                                                                      \
  }
  while (it < nk) {
    X6_STEP(fa0, fb0, fa1, fb1, P_SET, 1, 2); if (it >= nk) break;
    X6_STEP(fa1, fb1, fa0, fb0, Q_SET, 2, 0); if (it >= nk) break;
    X6_STEP(fa0, fb0, fa1, fb1, P_SET, 0, 1); if (it >= nk) break;
    X6_STEP(fa1, fb1, fa0, fb0, Q_SET, 1, 2); if (it >= nk) break;
    X6_STEP(fa0, fb0, fa1, fb1, P_SET, 2, 0); if (it >= nk) break;
    X6_STEP(fa1, fb1, fa0, fb0, Q_SET, 0, 1);
  }
  if (threadIdx.x == 0) { a.clk[2 * blockIdx.x] = clock64() - c_beg; a.clk[2 * blockIdx.x + 1] = wall_clock64() - w_beg; }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        yb[(long)row * a.T + t0 + wn * 64 + j * 32 + li] = acc[i][j][r];
      }
}

// X7: X2 with the pre-split weights brought in by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs,
// no ds_write for 2/3 of the staged bytes) into a three-stage ring; activations as in X2.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ __launch_bounds__(NT, 2) void conv_x7(const Args a) {
  __shared__ uint4 As[3][3][2][BM];      // 3 x 24 KB
  __shared__ uint4 Bs[2][3][2][BN];      // 2 x 12 KB
  const long long c_beg = clock64(), w_beg = wall_clock64();
  int b, t0; tile_of(a, b, t0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int ksteps_tap = a.Cin / BK, nk = 2 * ksteps_tap;
  const float* xb = a.x + (long)b * a.Cin * a.T;
  const int s_n = tid & 127, s_q = tid >> 7;
  const unsigned as_base = (unsigned)(size_t)&As[0][0][0][0];
  const unsigned uwave = __builtin_amdgcn_readfirstlane(wave);
  // A of step it -> ring stage it % 3: 24 wave-instructions of 1 KB, 3 per wave
  auto dma_a = [&](int it, int stage) {
    const uint4* wp = a.wpk + (size_t)it * (3 * 2 * BM);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int chunk = j * 8 + wave;                       // 1 KB chunk of the 24 KB stage
      glds16(wp + chunk * 64 + lane, as_base + (unsigned)(stage * 24576 + (j * 8 + uwave) * 1024));
    }
  };
  float pb0, pb1, pb2, pb3, qb0, qb1, qb2, qb3;
  bool pok, qok;
#define X7_LOADB(b0, b1, b2, b3, ok, it_)                                                 \
  {                                                                                        \
    const int tap = (it_) / ksteps_tap, c0 = ((it_) % ksteps_tap) * BK + 4 * s_q;          \
    const int ts = t0 + s_n - (1 - tap) * a.dil;                                           \
    const float* xs = ts >= 0 ? xb + (long)c0 * a.T + ts : xb;                             \
    b0 = xs[0]; b1 = xs[(long)a.T]; b2 = xs[2L * a.T]; b3 = xs[3L * a.T];                  \
    ok = ts >= 0;                                                                          \
  }
#define X7_STOREB(b0, b1, b2, b3, ok, buf)                                                 \
  {                                                                                        \
    unsigned h0, m0, l0, h1, m1, l1;                                                       \
    split3(ok ? b0 : 0.f, ok ? b1 : 0.f, h0, m0, l0);                                      \
    split3(ok ? b2 : 0.f, ok ? b3 : 0.f, h1, m1, l1);                                      \
    uint2* bd = reinterpret_cast<uint2*>(&Bs[buf][0][s_q >> 1][s_n]) + (s_q & 1);          \
    bd[0 * 2 * 2 * BN] = make_uint2(h0, h1);                                               \
    bd[1 * 2 * 2 * BN] = make_uint2(m0, m1);                                               \
    bd[2 * 2 * 2 * BN] = make_uint2(l0, l1);                                               \
  }
  auto mma = [&](int sa, auto curc) {
    constexpr int cur = decltype(curc)::value;
    bf16x8 af[2][3], bf[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        af[i][p] = __builtin_bit_cast(bf16x8, As[sa][p][lk][wm * 64 + i * 32 + li]);
        bf[i][p] = __builtin_bit_cast(bf16x8, Bs[cur][p][lk][wn * 64 + i * 32 + li]);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  dma_a(0, 0);
  dma_a(1, 1);
  X7_LOADB(pb0, pb1, pb2, pb3, pok, 0);
  X7_LOADB(qb0, qb1, qb2, qb3, qok, 1);
  X7_STOREB(pb0, pb1, pb2, pb3, pok, 0);          // the compiler's wait for P also covers both DMAs (older)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int sa = 0;
  for (int it = 0; it < nk; it += 2) {
    int s2 = sa + 2; if (s2 >= 3) s2 -= 3;
    dma_a(min(it + 2, nk - 2), s2);               // ring stage of step it + 2: last read in step it - 1
    X7_LOADB(pb0, pb1, pb2, pb3, pok, min(it + 2, nk - 2));
    mma(sa, I0{});
    X7_STOREB(qb0, qb1, qb2, qb3, qok, 1);        // waits (in order) for everything older than Q: the DMA of step it + 1 too
    __syncthreads();
    sa = sa == 2 ? 0 : sa + 1;
    s2 = sa + 2; if (s2 >= 3) s2 -= 3;
    dma_a(min(it + 3, nk - 1), s2);
    X7_LOADB(qb0, qb1, qb2, qb3, qok, min(it + 3, nk - 1));
    mma(sa, I1{});
    X7_STOREB(pb0, pb1, pb2, pb3, pok, 0);
    __syncthreads();
    sa = sa == 2 ? 0 : sa + 1;
  }
  if (threadIdx.x == 0) { a.clk[2 * blockIdx.x] = clock64() - c_beg; a.clk[2 * blockIdx.x + 1] = wall_clock64() - w_beg; }
  float* yb = a.y + (long)b * BM * a.T;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        yb[(long)row * a.T + t0 + wn * 64 + j * 32 + li] = acc[i][j][r];
      }
}

// ---- host ------------------------------------------------------------------------------------
static unsigned short bf16_rne(float f) {
  unsigned u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int B = 16, Cin = getenv("X3_CIN") ? atoi(getenv("X3_CIN")) : 256;     // 64: an 8-step contraction, epilogue-dominated
  const int dil = argc > 1 ? atoi(argv[1]) : 64;
  const int T = argc > 2 ? atoi(argv[2]) : 7680;
  const int reps = 20;
  std::vector<float> hx((size_t)B * Cin * T), hW((size_t)BM * Cin * 2);
  srand(1);
  for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hW) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  // pack: k index within the contraction = tap*Cin + c; K step it covers k = 16*it .. 16*it+15
  const int nk = 2 * Cin / BK;
  std::vector<unsigned short> pw((size_t)nk * 3 * 2 * BM * 8);
  for (int it = 0; it < nk; ++it)
    for (int lk = 0; lk < 2; ++lk)
      for (int m = 0; m < BM; ++m)
        for (int e = 0; e < 8; ++e) {
          const int k = it * BK + lk * 8 + e, tap = k / Cin, c = k % Cin;
          const float w = hW[((size_t)m * Cin + c) * 2 + tap];
          const unsigned short h = bf16_rne(w);
          const float r1 = w - bf16_f(h);
          const unsigned short mm = bf16_rne(r1);
          const float r2 = r1 - bf16_f(mm);
          const unsigned short l = bf16_rne(r2);
          const unsigned short pc[3] = {h, mm, l};
          for (int p = 0; p < 3; ++p) pw[((((size_t)it * 3 + p) * 2 + lk) * BM + m) * 8 + e] = pc[p];
        }
  float *dx, *dy; uint4* dw;
  CHECK(hipMalloc(&dx, hx.size() * 4)); CHECK(hipMalloc(&dy, (size_t)B * BM * T * 4));
  CHECK(hipMalloc(&dw, pw.size() * 2));
  CHECK(hipMemcpy(dx, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dw, pw.data(), pw.size() * 2, hipMemcpyHostToDevice));
  unsigned long long* dclk; CHECK(hipMalloc(&dclk, 16 * 4096)); CHECK(hipMemset(dclk, 0, 16 * 4096));
  Args a; a.clk = dclk; a.mode = 0; a.x = dx; a.y = dy; a.wpk = dw; a.Cin = Cin; a.T = T; a.B = B; a.dil = dil; a.ntile_n = T / BN;
  const int full_grid = B * (T / BN);
  const int grid = getenv("X3_GRID") ? atoi(getenv("X3_GRID")) : full_grid;    // fewer workgroups: part of the chip idle (power test)
  const double flop = 2.0 * B * T * BM * Cin * 2;
  std::vector<float> hy((size_t)B * BM * T);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto check = [&](const char* name) {
    CHECK(hipMemcpy(hy.data(), dy, hy.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst32 = 0;
    for (int s = 0; s < 400; ++s) {
      const int b = rand() % B, m = rand() % BM, t = rand() % T;
      double ref = 0; float ref32 = 0.f;
      for (int tap = 0; tap < 2; ++tap) {
        const int ts = t - (1 - tap) * dil;
        if (ts < 0) continue;
        for (int c = 0; c < Cin; ++c) {
          ref += (double)hW[((size_t)m * Cin + c) * 2 + tap] * hx[((size_t)b * Cin + c) * T + ts];
          ref32 = fmaf(hW[((size_t)m * Cin + c) * 2 + tap], hx[((size_t)b * Cin + c) * T + ts], ref32);
        }
      }
      worst = fmax(worst, fabs(ref - hy[((size_t)b * BM + m) * T + t]));
      worst32 = fmax(worst32, fabs(ref - (double)ref32));
    }
    printf("  %-4s max |err| vs float64 on 400 samples: %.3e   (a sequential fp32 fma chain: %.3e) %s\n", name, worst, worst32,
           worst < 2e-6 ? "ok" : "WRONG");
  };
  int nthreads = NT;
  int launch_grid = 0;         // 0: one workgroup per tile; else a persistent grid of this many workgroups
  auto run = [&](const char* name, void (*kern)(const Args)) {
    const int grid_tiles = grid;
    const int grid = launch_grid ? launch_grid : grid_tiles;
    CHECK(hipMemset(dy, 0, hy.size() * 4));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), 0, 0, a);
    CHECK(hipDeviceSynchronize());
    check(name);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(nthreads), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(2 * grid);
    CHECK(hipMemcpy(hc.data(), dclk, hc.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, tick = 0;
    for (int i = 0; i < grid; ++i) { cyc += hc[2 * i]; tick += hc[2 * i + 1]; }
    printf("%-6s %.1f us  %.1f TFLOP/s (fp32-equivalent)   main loop of a block: %.1f us at %.0f MHz\n", name, 1e3 * ms / reps,
           flop * grid_tiles / full_grid / (ms / reps * 1e-3) / 1e12, tick / grid / 100.0, tick > 0 ? cyc / tick * 100.0 : 0.0);
  };
  run("X1", conv_x1<0>);
  run("X2", conv_x2<0>);
  run("X7", conv_x7);
  launch_grid = 256; run("X8", conv_x8); launch_grid = 0;
  run("X5", conv_x5<0>);
  run("X5d4", conv_x5<1>);
  nthreads = 256;
  run("X3", conv_x3<0>);
  if (argc > 3) { run("X3/1", conv_x3<1>); run("X3/2", conv_x3<2>); run("X3/3", conv_x3<3>); run("X3/8", conv_x3<8>); }
  nthreads = NT;
  if (argc > 3) {
    run("X2/32", conv_x2<32>); run("X2/64", conv_x2<64>); run("X2/33", conv_x2<33>); run("X2/65", conv_x2<65>);
    run("X2/1", conv_x2<1>); run("X2/2", conv_x2<2>); run("X2/3", conv_x2<3>); run("X2/4", conv_x2<4>);
    run("X2/7", conv_x2<7>); run("X2/8", conv_x2<8>); run("X2/16", conv_x2<16>); run("X2/19", conv_x2<19>);
    run("X2/23", conv_x2<23>); run("X2/24", conv_x2<24>); run("X2/31", conv_x2<31>);
  }
  return 0;
}
