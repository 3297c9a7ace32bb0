// latent.hip -- a whole stack of latent-rate convolutions in ONE launch per direction.
//
// ConditionEmbed (net.py:29-53) is five "same"-padded dilated 3-tap convs + ReLU over a (B, C, T') tensor with T' = T / 64 =
// 120 columns: 47 MFLOP each at the configs.  As five conv launches forward and fifteen backward (backward-data, weight
// gradient, its reduce) they cost 25-60 us EACH -- all of it launch and round-trip latency on a chip that is otherwise idle
// (forward) or busy with the decoder's deferred weight gradients (backward, where a small launch waits twice as long for its
// turn).  A sample's whole state is C x T' floats = 30 KB, so one workgroup per sample keeps it in LDS and walks the
// stack: only the layer's weights come from memory (requested a layer ahead), every activation is written to HBM once (the
// backward needs it) and never read back.
//
//   cstack_fwd_kernel   h_l = relu(conv_l(h_{l-1}) + b_l), l = 1..L; h_0 = x
//   cstack_bwd_kernel   g_L = gy * (h_L > 0); per layer: gb_l, gW_l (this sample's share), g_{l-1} = conv_l^T(g_l) [* (h_{l-1} > 0)]
//   cstack_reduce_kernel  gW_l, gb_l = the samples' shares summed in ascending sample order (deterministic)
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- plain fp32 products, fp32 accumulation; the contraction runs tap-major,
// channel-minor (forward / backward-data) and over time (weight gradient).  One wave owns 32 x 32 output tiles.
#include "common.h"

namespace vq {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int CS_MAXL = 8;        // layers per launch
constexpr int CS_HALO = 16;       // largest dilation served
constexpr int CS_TMAX = 128;      // columns per sample

struct CStackArgs {
  const float* x;                  // (B, C, T)   h_0
  float* h[CS_MAXL];               // (B, C, T)   h_1 .. h_L (forward: written; backward: read)
  const float* W[CS_MAXL];         // (C, C, 3)
  const float* b[CS_MAXL];         // (C,) or null
  int dil[CS_MAXL];
  int L, B, T;
  // backward
  const float* gy;                 // (B, C, T)   gradient of h_L
  float* gx;                       // (B, C, T)   gradient of x (nullable)
  float* part;                     // [B][L][C * 3C + C]: this sample's share of gW_l (as [co][j * C + ci]) then of gb_l
};

template <int C> struct CStackGeom {
  static constexpr int TP = CS_TMAX + 2 * CS_HALO + 1;      // LDS row pitch of an activation: halo of zeros either side; ODD, so a
                                                            // walk down the rows (weight-gradient operands) touches 32 banks too
  static constexpr int WP = 3 * C + 1;                      // LDS row pitch of the weights (odd)
  static constexpr int NW = (C / 32) * 4;                   // waves: one 32 x 32 tile each of the C x 128 output
  static constexpr size_t lds_fwd = (size_t)(2 * C * TP) * sizeof(float);      // (the weights are read from L2 as the MFMA's A operand: a weight
                                                                               //  image would put the workgroup at 132 KB of LDS, and it has to fit BESIDE a
                                                                               //  49 KB weight-gradient workgroup of the decoder or it waits ~0.5 ms for a free CU)
};

// ---- forward ---------------------------------------------------------------------------------------------------------
// The forward runs in the head of the step, in front of the first gate launch, on an otherwise idle chip: it may keep the
// layer's weights in LDS (132 KB per workgroup) -- the backward, which has to fit beside the decoder's deferred weight-gradient
// workgroups, reads them from L2.  Round 6's first form fetched the A operand of every MFMA straight from L2, a lane per
// weight ROW: 64 cache lines per wave instruction, 768 such gathers per layer and workgroup = ~21 us per layer for 3 us of
// MFMA work.  Now W_l (C x 3 C, 48 KB) is read in whole 16-byte runs a layer ahead (24 registers per thread), written into an
// odd-pitch LDS image behind the layer's barrier, and the A operands are ds_read_b32s like the B operands.  Same products in
// the same order as before.
template <int C>
__global__ __launch_bounds__(CStackGeom<C>::NW * 64, 2) void cstack_fwd_kernel(const CStackArgs a) {
  using G = CStackGeom<C>;
  constexpr int TP = G::TP, WP = G::WP, NT = G::NW * 64;
  extern __shared__ float lds[];
  float* act0 = lds;                     // [C][TP]
  float* act1 = lds + C * TP;
  float* wimg = lds + 2 * C * TP;        // [C][WP]: W_l as [co][ci * 3 + j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int b = blockIdx.x, T = a.T;
  const int mt = wave & (C / 32 - 1), nt = wave / (C / 32);       // this wave's tile: rows 32 mt .., columns 32 nt ..
  const int n = nt * 32 + li;
  constexpr int H = C / 4;
  constexpr int NV = (C * 3 * C / 4 + NT - 1) / NT;                // float4 of a layer's weights per thread
  float4 wv[NV];
  auto fetch_w = [&](int l) {
    const float4* w4 = reinterpret_cast<const float4*>(a.W[l]);
#pragma unroll
    for (int k = 0; k < NV; ++k) { const int i = k * NT + tid; wv[k] = i < C * 3 * C / 4 ? w4[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int i = k * NT + tid;
      if (i < C * 3 * C / 4) {
        const int e = 4 * i, co = e / (3 * C), r = e - co * (3 * C);      // (3 C is a multiple of 4: a float4 never leaves its row)
        float* d = wimg + co * WP + r;
        d[0] = wv[k].x; d[1] = wv[k].y; d[2] = wv[k].z; d[3] = wv[k].w;
      }
    }
  };
  fetch_w(0);
  // zero both activation images (halos and the columns beyond T stay zero for the whole launch)
  for (int i = tid; i < 2 * C * TP; i += NT) lds[i] = 0.f;
  __syncthreads();
  {
    const float* xb = a.x + (long)b * C * T;
    for (int i = tid; i < C * T; i += NT) { const int c = i / T, t = i - c * T; act0[c * TP + CS_HALO + t] = xb[i]; }
  }
  store_w();
  __syncthreads();
  for (int l = 0; l < a.L; ++l) {
    const float* in = (l & 1) ? act1 : act0;
    float* out = (l & 1) ? act0 : act1;
    const int dil = a.dil[l];
    const bool more = l + 1 < a.L;
    if (more) fetch_w(l + 1);           // in flight under this layer's MFMAs
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* br = in + kh * TP + CS_HALO + n;
    const float* ar = wimg + (32 * mt + li) * WP + 3 * kh;
    // step (sx = 2 j + half, q): A = W[co = 32 mt + li][ci = 2 (q + H half) + kh][j], B = in[ci][n + (j - 1) dil]
#pragma unroll
    for (int sx = 0; sx < 6; ++sx) {
      const float* aq = ar + (sx >> 1) + 6 * H * (sx & 1);
      const float* bq = br + (2 * H * (sx & 1)) * TP + ((sx >> 1) - 1) * dil;
#pragma unroll
      for (int q = 0; q < H; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[6 * q], bq[2 * q * TP], acc, 0, 0, 0);
    }
    float* hb = a.h[l] + (long)b * C * T;
    const float* bias = a.b[l];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
      float v = acc[r] + (bias ? bias[m] : 0.f);
      v = fmaxf(v, 0.f);
      if (n < T) { out[m * TP + CS_HALO + n] = v; hb[(long)m * T + n] = v; }
    }
    __syncthreads();                    // `out` is complete, every wave is done with `in` and with the weight image
    if (more) { store_w(); __syncthreads(); }
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(CStackGeom<C>::NW * 64, 4) void cstack_bwd_kernel(const CStackArgs a) {
  using G = CStackGeom<C>;
  constexpr int TP = G::TP, NT = G::NW * 64, NW = G::NW;
  extern __shared__ float lds[];
  float* bufA = lds;                     // [C][TP]
  float* bufB = lds + C * TP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int b = blockIdx.x, T = a.T, L = a.L;
  for (int i = tid; i < 2 * C * TP; i += NT) lds[i] = 0.f;
  __syncthreads();
  // g_L = gy * (h_L > 0) -> bufA;  h_{L-1} -> bufB
  float* g = bufA;
  float* hp = bufB;
  {
    const float* gyb = a.gy + (long)b * C * T;
    const float* hl = a.h[L - 1] + (long)b * C * T;
    const float* hq = (L >= 2 ? a.h[L - 2] : a.x) + (long)b * C * T;
    for (int i = tid; i < C * T; i += NT) {
      const int c = i / T, t = i - c * T;
      g[c * TP + CS_HALO + t] = hl[i] > 0.f ? gyb[i] : 0.f;
      hp[c * TP + CS_HALO + t] = hq[i];
    }
  }
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    const int dil = a.dil[l];
    float* part = a.part + ((long)b * L + l) * (C * 3 * C + C);
    // ---- bias gradient: row sums of g
    if (tid < C) {
      const float* gr = g + tid * TP + CS_HALO;
      float s = 0.f;
      for (int t = 0; t < T; ++t) s += gr[t];
      part[C * 3 * C + tid] = s;
    }
    // ---- weight gradient of this sample: gW[co][(j, ci)] = sum_t g[co][t] * h_{l-1}[ci][t + (j - 1) dil]
    //      tiles: C / 32 row tiles x 3 C / 32 column tiles, dealt round-robin to the waves; K = T (pairs of t)
    for (int tile = wave; tile < (C / 32) * (3 * C / 32); tile += NW) {
      const int mt = tile % (C / 32), nt = tile / (C / 32);
      const int ncol = nt * 32 + li, j = ncol / C, ci = ncol - j * C;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* ar = g + (32 * mt + li) * TP + CS_HALO + kh;
      const float* br = hp + ci * TP + CS_HALO + kh + (j - 1) * dil;
#pragma unroll 8
      for (int t = 0; t < CS_TMAX; t += 2)          // (columns beyond T are zero in g)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[t], br[t], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
        part[(long)m * (3 * C) + ncol] = acc[r];
      }
    }
    __syncthreads();                    // hp (h_{l-1}) has been read by every wave: its buffer becomes g_{l-1}
    // ---- backward-data: g_{l-1}[ci][t] = sum_{j, co} W[co][ci][j] g[co][t - (j - 1) dil], masked by h_{l-1} > 0 (l >= 1)
    if (l > 0 || a.gx != nullptr) {
      const int mt = wave & (C / 32 - 1), nt = wave / (C / 32);
      const int n = nt * 32 + li;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // A operand of step (j, c): W[co = c + kh][ci = 32 mt + li][j] -- from L2, in sixths of the contraction through a ring of three
      // register buffers (see cstack_fwd_kernel)
      constexpr int H = C / 4;
      const float* wr = a.W[l] + (long)kh * (3 * C) + 3 * (32 * mt + li);
      const float* br = g + kh * TP + CS_HALO + n;
      auto fetch_a = [&](int sx, float (&av)[H]) {
        const float* w2 = wr + (long)(2 * H * (sx & 1)) * (3 * C) + (sx >> 1);
#pragma unroll
        for (int q = 0; q < H; ++q) av[q] = w2[(long)(2 * q) * (3 * C)];
      };
      auto mma6 = [&](const float (&av)[H], int sx) {
        const float* bq = br + (2 * H * (sx & 1)) * TP - ((sx >> 1) - 1) * dil;
#pragma unroll
        for (int q = 0; q < H; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bq[2 * q * TP], acc, 0, 0, 0);
      };
      float r0[H], r1[H], r2[H];
      fetch_a(0, r0); fetch_a(1, r1);
      fetch_a(2, r2); mma6(r0, 0);
      fetch_a(3, r0); mma6(r1, 1);
      fetch_a(4, r1); mma6(r2, 2);
      fetch_a(5, r2); mma6(r0, 3);
      mma6(r1, 4);
      mma6(r2, 5);
      float* gxb = (l == 0) ? a.gx + (long)b * C * T : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (n < T) {
          float* q = hp + m * TP + CS_HALO + n;          // holds h_{l-1}[m][n]; becomes g_{l-1}[m][n]
          const float v = (l > 0) ? (*q > 0.f ? acc[r] : 0.f) : acc[r];
          *q = v;
          if (gxb) gxb[(long)m * T + n] = v;
        }
      }
    }
    __syncthreads();                    // g (old) and wt are free
    if (l > 0) {
      // roles swap: g <- hp (now g_{l-1}); the old g buffer receives h_{l-2}
      float* tmp = g; g = hp; hp = tmp;
      const float* hq = (l >= 2 ? a.h[l - 2] : a.x) + (long)b * C * T;
      for (int i = tid; i < C * T; i += NT) { const int c = i / T, t = i - c * T; hp[c * TP + CS_HALO + t] = hq[i]; }
      __syncthreads();
    }
  }
}

// gW_l[co][ci][j] (+)= sum_b part[b][l][co][j * C + ci];  gb_l[co] (+)= sum_b part[b][l][C * 3C + co]   (ascending b)
struct CStackGrads { float* gW[CS_MAXL]; float* gb[CS_MAXL]; };
__global__ __launch_bounds__(256) void cstack_reduce_kernel(const float* __restrict__ part, const CStackGrads gr, int B, int L, int C, int accumulate) {
  const int per = C * 3 * C + C;
  const int total = L * per;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int l = i / per, e = i - l * per;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[((long)b * L + l) * per + e];
    if (e < C * 3 * C) {
      const int co = e / (3 * C), r = e - co * 3 * C, j = r / C, ci = r - j * C;
      float* d = gr.gW[l];
      if (d) { float* q = d + ((long)co * C + ci) * 3 + j; *q = accumulate ? *q + s : s; }
    } else {
      float* d = gr.gb[l];
      if (d) { float* q = d + (e - C * 3 * C); *q = accumulate ? *q + s : s; }
    }
  }
}

template <int C>
static int cstack_launch(const CStackArgs& a, bool backward, hipStream_t st) {
  using G = CStackGeom<C>;
  const size_t lds = backward ? G::lds_fwd : G::lds_fwd + (size_t)C * G::WP * sizeof(float);      // (forward: + the layer's weight image)
  if (backward) {
    static bool attr = false;
    if (!attr) { VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cstack_bwd_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    hipLaunchKernelGGL(cstack_bwd_kernel<C>, dim3(a.B), dim3(G::NW * 64), lds, st, a);
  } else {
    static bool attr = false;
    if (!attr) { VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cstack_fwd_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    hipLaunchKernelGGL(cstack_fwd_kernel<C>, dim3(a.B), dim3(G::NW * 64), lds, st, a);
  }
  VQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace vq

using namespace vq;

extern "C" int vqvae_convstack_supported(int L, int C, int T, const int* dil) {
  if (L < 1 || L > CS_MAXL || (C != 32 && C != 64) || T < 1 || T > CS_TMAX || !dil) return 0;
  for (int l = 0; l < L; ++l) if (dil[l] < 1 || dil[l] > CS_HALO) return 0;
  return 1;
}

extern "C" size_t vqvae_convstack_workspace_bytes(int L, int B, int C) {
  return (size_t)B * L * ((size_t)C * 3 * C + C) * sizeof(float);
}

extern "C" int vqvae_convstack_fwd(int L, int B, int C, int T, const int* dil, const float* x, const float* const* W,
                                   const float* const* b, float* const* h, vqvae_stream_t s) {
  VQ_REQUIRE(dil && x && W && b && h && B > 0 && vqvae_convstack_supported(L, C, T, dil), "convstack_fwd: unsupported stack (L <= 8, C in {32, 64}, T <= 128, dilations <= 16)");
  CStackArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.L = L; a.B = B; a.T = T;
  for (int l = 0; l < L; ++l) {
    VQ_REQUIRE(W[l] && h[l], "convstack_fwd: null layer %d", l);
    a.W[l] = W[l]; a.b[l] = b[l]; a.h[l] = h[l]; a.dil[l] = dil[l];
  }
  return C == 64 ? cstack_launch<64>(a, false, (hipStream_t)s) : cstack_launch<32>(a, false, (hipStream_t)s);
}

extern "C" int vqvae_convstack_bwd(int L, int B, int C, int T, const int* dil, const float* x, const float* const* W,
                                   const float* const* h, const float* gy, float* gx, float* const* gW, float* const* gb,
                                   int accumulate, void* ws, size_t ws_bytes, vqvae_stream_t s) {
  VQ_REQUIRE(dil && x && W && h && gy && gW && gb && ws && B > 0 && vqvae_convstack_supported(L, C, T, dil), "convstack_bwd: unsupported stack");
  if (ws_bytes < vqvae_convstack_workspace_bytes(L, B, C)) { set_error("convstack_bwd: workspace too small"); return VQVAE_E_WORKSPACE; }
  CStackArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.L = L; a.B = B; a.T = T; a.gy = gy; a.gx = gx; a.part = (float*)ws;
  CStackGrads gr; memset(&gr, 0, sizeof(gr));
  for (int l = 0; l < L; ++l) {
    VQ_REQUIRE(W[l] && h[l], "convstack_bwd: null layer %d", l);
    a.W[l] = W[l]; a.h[l] = const_cast<float*>(h[l]); a.dil[l] = dil[l];
    gr.gW[l] = gW[l]; gr.gb[l] = gb[l];
  }
  if (int e = (C == 64 ? cstack_launch<64>(a, true, (hipStream_t)s) : cstack_launch<32>(a, true, (hipStream_t)s))) return e;
  const int total = L * (C * 3 * C + C);
  hipLaunchKernelGGL(cstack_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)s, (const float*)ws, gr, B, L, C, accumulate);
  VQ_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================================
// Backward of ONE encoder stage (net.py:12-17: Convolution2D(C, C, (4, 1), stride 2, pad 1)) in ONE launch + a reduce:
//     y[co][t] = b[co] + sum_{ci, j} W[co][ci][j] x[ci][2 t + j - 1]          x: (B, C, Tin), y: (B, C, Tout), Tout = Tin / 2
//   gx[ci][u] = sum_{co, j} W[co][ci][j] gy[co][(u + 1 - j) / 2]  (where that is an integer), optionally * (x > 0)
//   gW[co][ci][j] = sum_{b, t} gy[co][t] x[ci][2 t + j - 1],  gb[co] = sum_{b, t} gy[co][t]
// It used to be four launches per stage (backward-data GEMM, phase split of x, weight-gradient GEMM, its reduce), each 15-150 us
// of latency in the sweep's tail.  A workgroup owns 128 input columns [p0, p0 + 128) of one sample:
//   * backward-data by PARITY: an even column u = 2 v takes taps 1 and 3 (t = v, v - 1), an odd one taps 0 and 2 (t = v + 1,
//     v) -- two dense (C x 64) x K = 2 C products instead of one (C x 128) x K = 4 C with every other operand zero; with
//     C = 64 that is exactly one 32 x 32 MFMA tile per wave (parity x row half x column half);
//   * the weight gradient over the 64 output columns t = p0 / 2 .. + 63 the workgroup owns: (C x 4 C) x K = 64, two tiles
//     per wave, written as this workgroup's share; cstage_reduce_kernel sums the shares in ascending order.
// fp32 MFMA arithmetic (v_mfma_f32_32x32x2_f32); gy slice (17 KB) and x slice (34 KB) staged once in LDS, W read from L2.
// =====================================================================================================================
namespace vq {

struct CStageArgs {
  const float* x; const float* gy; const float* W;      // (B, C, Tin), (B, C, Tout), (C, C, 4)
  float* gx;                                            // (B, C, Tin) or null
  float* part;                                          // [B * nslice][C * 4 C + C]
  int B, Tin, Tout, nslice, mask;                       // mask: gx *= (x > 0)
};

template <int C>
__global__ __launch_bounds__(512, 4) void cstage_bwd_kernel(const CStageArgs a) {
  static_assert(C == 64, "one MFMA tile per wave for backward-data needs C == 64");
  constexpr int GP = 67, HP = 131, NT = 512;
  extern __shared__ float lds[];
  float* gl = lds;                        // [C][GP]: gy[co][t0 - 1 + i], i = 0 .. 65
  float* hl = gl + C * GP;                // [C][HP]: x[ci][p0 - 1 + i], i = 0 .. 129
  // (50 KB: the workgroup has to fit beside a 49 KB weight-gradient workgroup of the decoder, see cstack; backward-data's
  //  weights are read from L2 as the MFMA's A operand)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int b = blockIdx.x / a.nslice, sl = blockIdx.x - b * a.nslice;
  const int p0 = sl * 128, t0 = p0 / 2;
  const int Tin = a.Tin, Tout = a.Tout;
  // ---- stage everything (all loads are independent: they travel together)
  {
    const float* gyb = a.gy + (long)b * C * Tout;
    for (int i = tid; i < C * 66; i += NT) {
      const int c = i / 66, q = i - c * 66, t = t0 - 1 + q;
      gl[c * GP + q] = (t >= 0 && t < Tout) ? gyb[(long)c * Tout + t] : 0.f;
    }
    const float* xb = a.x + (long)b * C * Tin;
    for (int i = tid; i < C * 130; i += NT) {
      const int c = i / 130, q = i - c * 130, u = p0 - 1 + q;
      hl[c * HP + q] = (u >= 0 && u < Tin) ? xb[(long)c * Tin + u] : 0.f;
    }
  }
  __syncthreads();
  float* part = a.part + (long)blockIdx.x * (C * 4 * C + C);
  // ---- bias share: the owned columns t0 .. t0 + 63 are gl columns 1 .. 64
  if (tid < C) {
    float s = 0.f;
    for (int q = 1; q <= 64; ++q) s += gl[tid * GP + q];
    part[C * 4 * C + tid] = s;
  }
  // ---- weight-gradient share: (co) x (j, ci), K = the 64 owned t; tiles: 2 row x 8 column, two per wave
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int tile = wave + 8 * rep, mt = tile & 1, nt = tile >> 1;
    const int ncol = nt * 32 + li, j = ncol / C, ci = ncol - j * C;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ar = gl + (32 * mt + li) * GP + 1 + kh;                 // gy[co][t0 + t + kh]
    const float* br = hl + ci * HP + 2 * kh + j;                          // x[ci][2 (t0 + t + kh) + j - 1] = hl[ci][2 t + 2 kh + j]
#pragma unroll 8
    for (int t = 0; t < 64; t += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[t], br[2 * t], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
      part[(long)m * (4 * C) + ncol] = acc[r];
    }
  }
  // ---- backward-data by parity
  if (a.gx != nullptr) {
    const int par = wave & 1, mt = (wave >> 1) & 1, nt = wave >> 2;
    const int v = nt * 32 + li;                                            // local column pair index: u = p0 + 2 v + par
    // even u: taps (1, t = v), (3, t = v - 1); odd u: taps (0, t = v + 1), (2, t = v);  gl column of t0 + t' is t' + 1
    const int ja = par ? 0 : 1, jb = par ? 2 : 3;
    const int qa = par ? v + 2 : v + 1, qb = par ? v + 1 : v;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // A operand of step (j, c): W[co = c + kh][ci = 32 mt + li][j]: both taps' 32 + 32 values requested before the first MFMA
    const float* wr = a.W + (long)kh * (4 * C) + 4 * (32 * mt + li);
    float wa[C / 2], wb[C / 2];
#pragma unroll
    for (int q = 0; q < C / 2; ++q) { wa[q] = wr[(long)(2 * q) * (4 * C) + ja]; wb[q] = wr[(long)(2 * q) * (4 * C) + jb]; }
    const float* ba = gl + kh * GP + qa;
    const float* bb = gl + kh * GP + qb;
#pragma unroll
    for (int q = 0; q < C / 2; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[q], ba[2 * q * GP], acc, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < C / 2; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[q], bb[2 * q * GP], acc, 0, 0, 0);
    const int u = p0 + 2 * v + par;
    float* gxb = a.gx + (long)b * C * Tin;
    if (u < Tin) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * kh;
        float val = acc[r];
        if (a.mask) val = hl[m * HP + (u - p0 + 1)] > 0.f ? val : 0.f;
        gxb[(long)m * Tin + u] = val;
      }
    }
  }
}

// gW[co][ci][j] (+)= sum_w part[w][co][j * C + ci];  gb[co] (+)= sum_w part[w][C * 4C + co]   (ascending w: deterministic)
__global__ __launch_bounds__(256) void cstage_reduce_kernel(const float* __restrict__ part, int nwg, int C, float* __restrict__ gW,
                                                            float* __restrict__ gb, int accumulate) {
  const int per = C * 4 * C + C;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= per) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                 // four independent chains keep the loads in flight; the ORDER of the final
  int w = 0;                                                    // additions is fixed, so the result is reproducible
  for (; w + 3 < nwg; w += 4) {
    s0 += part[(long)w * per + e]; s1 += part[(long)(w + 1) * per + e];
    s2 += part[(long)(w + 2) * per + e]; s3 += part[(long)(w + 3) * per + e];
  }
  for (; w < nwg; ++w) s0 += part[(long)w * per + e];
  const float s = (s0 + s1) + (s2 + s3);
  if (e < C * 4 * C) {
    if (gW) { const int co = e / (4 * C), r = e - co * 4 * C, j = r / C, ci = r - j * C; float* q = gW + ((long)co * C + ci) * 4 + j; *q = accumulate ? *q + s : s; }
  } else if (gb) { float* q = gb + (e - C * 4 * C); *q = accumulate ? *q + s : s; }
}

}  // namespace vq

extern "C" int vqvae_conv_s2_bwd_supported(int Cin, int Cout, int K, int stride, int pad, int dil, int Tin, int Tout) {
  return (Cin == 64 && Cout == 64 && K == 4 && stride == 2 && pad == 1 && dil == 1 && Tin >= 2 && Tout == Tin / 2) ? 1 : 0;
}
extern "C" size_t vqvae_conv_s2_bwd_workspace_bytes(int B, int C, int Tin) {
  return (size_t)B * ((Tin + 127) / 128) * ((size_t)C * 4 * C + C) * sizeof(float);
}
extern "C" int vqvae_conv_s2_bwd(int B, int C, int Tin, int Tout, const float* x, const float* W, const float* gy, int mask_by_x,
                                 float* gx, float* gW, float* gb, int accumulate, void* ws, size_t ws_bytes, vqvae_stream_t s) {
  VQ_REQUIRE(x && W && gy && ws && B > 0 && vqvae_conv_s2_bwd_supported(C, C, 4, 2, 1, 1, Tin, Tout), "conv_s2_bwd: unsupported stage (C = 64, 4 taps, stride 2, pad 1, Tout = Tin / 2)");
  if (ws_bytes < vqvae_conv_s2_bwd_workspace_bytes(B, C, Tin)) { set_error("conv_s2_bwd: workspace too small"); return VQVAE_E_WORKSPACE; }
  CStageArgs a; memset(&a, 0, sizeof(a));
  a.x = x; a.gy = gy; a.W = W; a.gx = gx; a.part = (float*)ws; a.B = B; a.Tin = Tin; a.Tout = Tout;
  a.nslice = (Tin + 127) / 128; a.mask = mask_by_x;
  const size_t lds = (size_t)(64 * 67 + 64 * 131) * sizeof(float);
  static bool attr = false;
  if (!attr) { VQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cstage_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
  hipLaunchKernelGGL(cstage_bwd_kernel<64>, dim3(B * a.nslice), dim3(512), lds, (hipStream_t)s, a);
  VQ_LAUNCH_CHECK();
  if (gW || gb) {
    const int per = C * 4 * C + C;
    hipLaunchKernelGGL(cstage_reduce_kernel, dim3((per + 255) / 256), dim3(256), 0, (hipStream_t)s, (const float*)ws, B * a.nslice, C, gW, gb, accumulate);
    VQ_LAUNCH_CHECK();
  }
  return 0;
}
