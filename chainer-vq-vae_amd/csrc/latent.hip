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
  static constexpr size_t lds_fwd = (size_t)(2 * C * TP + C * WP) * sizeof(float);
};

// ---- forward ---------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(CStackGeom<C>::NW * 64) void cstack_fwd_kernel(const CStackArgs a) {
  using G = CStackGeom<C>;
  constexpr int TP = G::TP, WP = G::WP, NT = G::NW * 64;
  constexpr int WPT = (C * 3 * C + NT - 1) / NT;           // weight elements per thread and layer
  extern __shared__ float lds[];
  float* act0 = lds;                     // [C][TP]
  float* act1 = lds + C * TP;
  float* wl = lds + 2 * C * TP;          // [C][WP]: wl[co][j * C + ci] = W[co][ci][j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int b = blockIdx.x, T = a.T;
  // zero both activation images (halos and the columns beyond T stay zero for the whole launch)
  for (int i = tid; i < 2 * C * TP; i += NT) lds[i] = 0.f;
  float wreg[WPT];
  auto fetch_w = [&](int l) {
    const float* W = a.W[l];
#pragma unroll
    for (int i = 0; i < WPT; ++i) { const int e = tid + NT * i; wreg[i] = e < C * 3 * C ? W[e] : 0.f; }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = tid + NT * i;
      if (e < C * 3 * C) { const int co = e / (3 * C), r = e - co * 3 * C, ci = r / 3, j = r - 3 * ci; wl[co * WP + j * C + ci] = wreg[i]; }
    }
  };
  fetch_w(0);
  __syncthreads();
  {
    const float* xb = a.x + (long)b * C * T;
    for (int i = tid; i < C * T; i += NT) { const int c = i / T, t = i - c * T; act0[c * TP + CS_HALO + t] = xb[i]; }
  }
  store_w();
  __syncthreads();
  const int mt = wave & (C / 32 - 1), nt = wave / (C / 32);       // this wave's tile: rows 32 mt .., columns 32 nt ..
  const int n = nt * 32 + li;
  for (int l = 0; l < a.L; ++l) {
    const float* in = (l & 1) ? act1 : act0;
    float* out = (l & 1) ? act0 : act1;
    if (l + 1 < a.L) fetch_w(l + 1);                               // travels under this layer's MFMAs
    const int dil = a.dil[l];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* ar = wl + (32 * mt + li) * WP + kh;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float* br = in + kh * TP + CS_HALO + n + (j - 1) * dil;
#pragma unroll 8
      for (int c = 0; c < C; c += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[j * C + c], br[c * TP], acc, 0, 0, 0);
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
    __syncthreads();                    // every wave is done with `in` and with wl
    if (l + 1 < a.L) { store_w(); __syncthreads(); }
  }
}

// ---- backward --------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(CStackGeom<C>::NW * 64) void cstack_bwd_kernel(const CStackArgs a) {
  using G = CStackGeom<C>;
  constexpr int TP = G::TP, WP = G::WP, NT = G::NW * 64, NW = G::NW;
  constexpr int WPT = (C * 3 * C + NT - 1) / NT;
  extern __shared__ float lds[];
  float* bufA = lds;                     // [C][TP]
  float* bufB = lds + C * TP;
  float* wt = lds + 2 * C * TP;          // [C][WP]: wt[ci][j * C + co] = W[co][ci][j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, kh = lane >> 5;
  const int b = blockIdx.x, T = a.T, L = a.L;
  for (int i = tid; i < 2 * C * TP; i += NT) lds[i] = 0.f;
  float wreg[WPT];
  auto fetch_w = [&](int l) {
    const float* W = a.W[l];
#pragma unroll
    for (int i = 0; i < WPT; ++i) { const int e = tid + NT * i; wreg[i] = e < C * 3 * C ? W[e] : 0.f; }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = tid + NT * i;
      if (e < C * 3 * C) { const int co = e / (3 * C), r = e - co * 3 * C, ci = r / 3, j = r - 3 * ci; wt[ci * WP + j * C + co] = wreg[i]; }
    }
  };
  fetch_w(L - 1);
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
  store_w();
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    const int dil = a.dil[l];
    float* part = a.part + ((long)b * L + l) * (C * 3 * C + C);
    if (l > 0) fetch_w(l - 1);
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
      const float* ar = wt + (32 * mt + li) * WP + kh;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float* br = g + kh * TP + CS_HALO + n - (j - 1) * dil;
#pragma unroll 8
        for (int c = 0; c < C; c += 2)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[j * C + c], br[c * TP], acc, 0, 0, 0);
      }
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
      store_w();
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
  const size_t lds = G::lds_fwd;
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
