// vq.hip -- the vector quantiser: nearest-codebook lookup (StraightThrough.forward,
// utils.py:176-211) and its codebook gradient (utils.py:213-231).
//
// The reference's index is argmin_j of an fp32 distance evaluated as
//     acc = 0; for c in 0..d-1: acc = acc + (z[c]-W[j][c])**2      (each op rounded)
// with numpy.argmin's first-minimum rule.  Evaluating that form for all N*k
// pairs is 3 non-fusable VALU ops per term; instead:
//   1. vq_mfma_x3_kernel (matmul mode 2, d = 64 / 128: six bf16 MFMA products of an exact
//      three-way operand split per fp32 product) or vq_mfma_reg_kernel / vq_mfma_kernel
//      (v_mfma_f32_32x32x2_f32) -- dist'(j,n) = |w_j|^2 - 2 <w_j, z_n> on the matrix
//      cores (codes x latents tiles), per-lane running
//      (min, argmin, runner-up) and a cross-half wavefront reduce.  A row is
//      CERTAIN when runner-up - min exceeds a rigorous rounding band; otherwise it is
//      queued for
//   2. vq_exact_kernel -- all k codes re-evaluated in the reference's operation
//      order (contraction off), ties to the lowest index.  Typically 1-3 % of rows.
// The band: the reference's sequential form and the fp32 expansion form are each within
// E = (2d+3)*u*2S = (4d+6)uS of the real-number distance (u = 2^-24, S = |z|^2 + max_j |w_j|^2).
// A row is CERTAIN only if no other code can win in the REFERENCE's arithmetic: with
// ours_j = true_j - |z|^2 + e_j (|e_j| <= E_ours) and ref_j = true_j + r_j (|r_j| <= E_ref),
// ours_j - ours_i1 > band implies ref_j - ref_i1 > band - 2 E_ours - 2 E_ref, so the band must
// cover 2 (E_ours + E_ref) = (16d + 24) uS for the fp32 MFMA sweep: we use 16(d+4)uS.  (Round 2
// used 12(d+4)uS, which covered 2 E_ours + E_ref only.)  The bf16-pipe sweep has a larger E_ours
// and a wider band, see vq_mfma_x3_kernel.
#include "common.h"

namespace vq {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void vq_wnorm_kernel(const float* __restrict__ W, int k, int d, float* wn, int* wmax_bits) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= k) return;
  float s = 0.f;
  for (int c = 0; c < d; ++c) { const float w = W[(long)j * d + c]; s = fmaf(w, w, s); }
  wn[j] = s;
  atomicMax(wmax_bits, __float_as_int(s));   // s >= 0: int order == float order
}
// vq_wnorm_elt_lds_kernel: the same, plus max_ij |W_ij| in wmax_bits[1] (the two-piece fp16 sweep scales the codebook by a power of
// two from it);
// the same values in the same order (s = fma(w_c, w_c, s), c ascending), rows arriving coalesced through LDS: a thread per code
// reading its row from global memory was 64 cache lines per wave instruction and 20 us for the training shape's 512 x 64
// codebook, in front of the quantiser on the step's critical path.  64 codes per workgroup, dynamic LDS 64 (d + 1) floats.
__global__ __launch_bounds__(64) void vq_wnorm_elt_lds_kernel(const float* __restrict__ W, int k, int d, float* wn, int* wmax_bits) {
  extern __shared__ float wn_tile[];
  const int j0 = blockIdx.x * 64, pitch = d + 1;
  const int nj = min(64, k - j0);
  for (int i = threadIdx.x; i < nj * d; i += 64) {
    const int r = i / d, c = i - r * d;
    wn_tile[r * pitch + c] = W[(long)j0 * d + i];
  }
  __syncthreads();
  if ((int)threadIdx.x >= nj) return;
  const float* w_ = wn_tile + threadIdx.x * pitch;
  float s = 0.f, m = 0.f;
  for (int c = 0; c < d; ++c) { const float w = w_[c]; s = fmaf(w, w, s); m = fmaxf(m, fabsf(w)); }
  wn[j0 + threadIdx.x] = s;
  atomicMax(wmax_bits, __float_as_int(s));
  atomicMax(wmax_bits + 1, __float_as_int(m));
}

// block: NW wavefronts, each owning 32 latent columns; Zs[d][32*NW], Ws[32][d+1]
template <int NW>
__global__ __launch_bounds__(64 * NW) void vq_mfma_kernel(
    const float* __restrict__ z, const float* __restrict__ W, const float* __restrict__ wn,
    const int* __restrict__ wmax_bits, int B, int d, int T, int k, int dpad,
    int32_t* __restrict__ idx, int32_t* __restrict__ flagged, int32_t* __restrict__ nflag) {
  extern __shared__ float smem[];
  constexpr int NC = 32 * NW;
  float* Zs = smem;                       // [dpad][NC]
  float* Ws = smem + (size_t)dpad * NC;   // [32][dpad+1]
  const int WPITCH = dpad + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const long N = (long)B * T;
  const long n0 = (long)blockIdx.x * NC;

  // stage the latent columns (coalesced along t), zero padded in c and n
  for (int e = tid; e < dpad * NC; e += 64 * NW) {
    const int c = e / NC, col = e % NC;
    const long n = n0 + col;
    float v = 0.f;
    if (c < d && n < N) { const long bb = n / T, t = n % T; v = z[(bb * d + c) * T + t]; }
    Zs[c * NC + col] = v;
  }

  float m1 = INFINITY, m2 = INFINITY;
  int i1 = 0x7fffffff;
  const int ntile = (k + 31) / 32;
  for (int jt = 0; jt < ntile; ++jt) {
    __syncthreads();    // previous tile consumed (and Zs staged, first time)
    for (int e = tid; e < 32 * dpad; e += 64 * NW) {
      const int r = e / dpad, c = e % dpad;
      const int j = jt * 32 + r;
      Ws[r * WPITCH + c] = (j < k && c < d) ? W[(long)j * d + c] : 0.f;
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* wrow = Ws + li * WPITCH + lk;
    const float* zcol = Zs + lk * NC + wave * 32 + li;
#pragma unroll 8
    for (int kk = 0; kk < dpad / 2; ++kk) {
      const float av = wrow[2 * kk];
      const float bv = zcol[(2 * kk) * NC];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (j < k) {
        const float v = fmaf(-2.f, acc[r], wn[j]);
        if (v < m1) { m2 = m1; m1 = v; i1 = j; }
        else if (v < m2) { m2 = v; }
      }
    }
  }
  // merge the two half-waves (same column, disjoint code rows)
  {
    const float om1 = __shfl_xor(m1, 32, 64);
    const float om2 = __shfl_xor(m2, 32, 64);
    const int oi1 = __shfl_xor(i1, 32, 64);
    if (om1 < m1 || (om1 == m1 && oi1 < i1)) {
      m2 = fminf(m1, om2); m1 = om1; i1 = oi1;
    } else {
      m2 = fminf(m2, om1);
    }
  }
  const long n = n0 + wave * 32 + li;
  if (lk == 0 && n < N) {
    float zn = 0.f;
    for (int c = 0; c < d; ++c) { const float v = Zs[c * NC + wave * 32 + li]; zn = fmaf(v, v, zn); }
    const float wmax = __int_as_float(*wmax_bits);
    const float band = 16.f * (float)(d + 4) * 5.9604645e-8f * (zn + wmax);
    idx[n] = i1;
    if (!(m2 - m1 > band)) {           // also catches NaN
      const int slot = atomicAdd(nflag, 1);
      flagged[slot] = (int32_t)n;
    }
  }
}

// Register-resident variant for d == 2*DP (64 or 128): each wavefront keeps the B fragment of
// its 32 latent columns in DP VGPRs for the whole sweep, the codebook streams through LDS in
// double-buffered tiles of 64 codes (two independent 32x32 accumulators per wave keep the
// matrix pipe issuing back to back), one barrier per tile.
template <int DP>
__global__ __launch_bounds__(256, 2) void vq_mfma_reg_kernel(
    const float* __restrict__ z, const float* __restrict__ W, const float* __restrict__ wn,
    const int* __restrict__ wmax_bits, int B, int T, int k,
    int32_t* __restrict__ idx, int32_t* __restrict__ flagged, int32_t* __restrict__ nflag) {
  constexpr int D = 2 * DP, WP = D + 1, TILE = 64;
  extern __shared__ float smem[];          // Wt[2][TILE][WP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const long N = (long)B * T;
  const long n = (long)blockIdx.x * 128 + wave * 32 + li;      // this lane's latent column
  // B fragment: lane (k = lk, j = li) holds z[c = 2*kk + lk][column li] for every kk
  float zf[DP];
  float zn = 0.f;
  {
    const bool ok = n < N;
    const long bb = ok ? n / T : 0, t = ok ? n % T : 0;
    const float* zp = z + (bb * D) * T + t;
#pragma unroll
    for (int kk = 0; kk < DP; ++kk) {
      zf[kk] = ok ? zp[(long)(2 * kk + lk) * T] : 0.f;
      zn = fmaf(zf[kk], zf[kk], zn);
    }
    zn += __shfl_xor(zn, 32, 64);           // both halves of the column
  }
  // staging role: thread loads 8 float4 of a 64 x D tile (row-contiguous) per tile
  constexpr int F4_PER_ROW = D / 4, F4_PER_THREAD = TILE * F4_PER_ROW / 256;
  float4 st[F4_PER_THREAD];
  auto load_tile = [&](int jt) {
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
      const int f = tid + 256 * i;
      const int r = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
      const int j = jt * TILE + r;
      st[i] = j < k ? *reinterpret_cast<const float4*>(W + (long)j * D + 4 * c4)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](int buf) {
    float* wt = smem + (size_t)buf * TILE * WP;
#pragma unroll
    for (int i = 0; i < F4_PER_THREAD; ++i) {
      const int f = tid + 256 * i;
      const int r = f / F4_PER_ROW, c4 = f % F4_PER_ROW;
      float* p = wt + r * WP + 4 * c4;
      p[0] = st[i].x; p[1] = st[i].y; p[2] = st[i].z; p[3] = st[i].w;
    }
  };
  float m1 = INFINITY, m2 = INFINITY;
  int i1 = 0x7fffffff;
  const int ntile = (k + TILE - 1) / TILE;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int jt = 0; jt < ntile; ++jt) {
    const int cur = jt & 1;
    if (jt + 1 < ntile) load_tile(jt + 1);
    const float* wt = smem + (size_t)cur * TILE * WP;
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    const float* w0 = wt + li * WP + lk;
    const float* w1 = w0 + 32 * WP;
#pragma unroll
    for (int kk = 0; kk < DP; ++kk) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w0[2 * kk], zf[kk], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[2 * kk], zf[kk], a1, 0, 0, 0);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = jt * TILE + h * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (j < k) {
          const float v = fmaf(-2.f, h ? a1[r] : a0[r], wn[j]);
          if (v < m1) { m2 = m1; m1 = v; i1 = j; }
          else if (v < m2) { m2 = v; }
        }
      }
    if (jt + 1 < ntile) store_tile(cur ^ 1);
    __syncthreads();
  }
  {
    const float om1 = __shfl_xor(m1, 32, 64);
    const float om2 = __shfl_xor(m2, 32, 64);
    const int oi1 = __shfl_xor(i1, 32, 64);
    if (om1 < m1 || (om1 == m1 && oi1 < i1)) { m2 = fminf(m1, om2); m1 = om1; i1 = oi1; }
    else { m2 = fminf(m2, om1); }
  }
  if (lk == 0 && n < N) {
    const float wmax = __int_as_float(*wmax_bits);
    const float band = 16.f * (float)(D + 4) * 5.9604645e-8f * (zn + wmax);
    idx[n] = i1;
    if (!(m2 - m1 > band)) {
      const int slot = atomicAdd(nflag, 1);
      flagged[slot] = (int32_t)n;
    }
  }
}

// ---- the same sweep on the bf16 matrix pipe (matmul mode 2, d = 64 or 128) ------------------------
// <w_j, z_n> as six v_mfma_f32_32x32x16_bf16 products of an exact three-way bf16 split of both
// operands (csrc/gemm_common.h, "matmul mode 2"): 6 x 32 cycles per 16 c instead of 8 x 64.  The
// codebook is split once per call (vq_wsplit_kernel: [piece][code][d] bf16), the latent fragment of
// a wavefront's 32 columns is split once and stays in registers for the whole sweep.
// Rounding band: each kept product is exact, the three dropped ones are < 2^-25 |ab|; an MFMA adds 16
// exact products to its accumulator -- taken here as no better than 17 individually rounded
// additions -- so a dot product is within (6d/16 * 17 + d/8) u |w||z| <= 3.3 d u S of the real
// number (S = |z|^2 + max|w|^2 >= 2|w||z|) and dist' = |w|^2 - 2<w,z> within E_ours = (7.6 d + 2) u S.
// The reference's own sequential fp32 sum is within E_ref = (4 d + 6) u S of the real number, and a row
// may skip the exact re-check only if no competitor can win in the REFERENCE's arithmetic: the band
// must cover 2 (E_ours + E_ref) = (23.2 d + 16) u S -> 24(d+4)uS.  (Round 2 used 16(d+4)uS = 2 E_ours
// only.)  The accumulation order inside the MFMA is not documented; the 17-roundings-per-MFMA model is
// the pessimistic end (every product added with its own rounding), so the bound does not rest on it.
using bf16x8v = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2v = __attribute__((ext_vector_type(2))) __bf16;
__device__ __forceinline__ unsigned vq_pk(float lo, float hi) {
  bf16x2v v; v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void vq_split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = vq_pk(x0, x1);
  float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = vq_pk(r0, r1);
  r0 -= __builtin_bit_cast(float, m << 16); r1 -= __builtin_bit_cast(float, m & 0xffff0000u);
  l = vq_pk(r0, r1);
}
// ---- the sweep as THREE fp16 products (matmul mode 3, `float32x2`: csrc/gemm_common.h "matmul mode 3") ------------------
// x 2^k = hi + lo (fp16, RNE), <w, z> ~= (w_lo z_hi + w_hi z_lo + w_hi z_hi) 2^-(kw + kz): half the MFMAs of the six-product
// sweep.  The codebook takes ONE power of two (from max_ij |W_ij|: vq_wnorm_elt_lds_kernel), every latent row its OWN (from its
// d entries, which its lane pair holds anyway): 2^(14 - e) puts the largest entry in [2^14, 2^15).  Rounding band:
//   * representation: |x - (hi + lo) 2^-k| <= 2^-22 |x| + 2^-39 max|x| per entry, the dropped w_lo z_lo <= 2^-22 |w z|: over a
//     row, <= 3 * 2^-22 |w||z| + 2^-39 sqrt(d) (max|z_n| |w| + max|W| |z|) <= (6 + 2^-13 sqrt(d)) u S  (u = 2^-24, S = |z|^2 +
//     max|w|^2 >= 2 |w||z|, max|W_ij| <= max|w|, max|z_n| <= |z|);
//   * accumulation: 3 MFMAs per 16 c, each taken as 17 individually rounded additions (the pessimistic end, as for the
//     six-product sweep): 3 d / 16 * 17 u |w||z| <= 1.6 d u S;
//   so a dot product is within (1.6 d + 7) u S, dist' = |w|^2 - 2 <w,z> within E_ours = (4.2 d + 16) u S (|w|^2 itself: (d + 1) u
//   |w|^2, the final fma: u S), and the band 2 (E_ours + E_ref) = (16.4 d + 44) u S -> 17 (d + 4) u S (the six-product sweep: 24).
// fp16 range: products reach 2^30, row sums 2^37 -- far inside fp32; an all-zero row or codebook keeps a finite scale
// (exponents clamped to >= -60, so that 2^-(kw + kz) stays a normal fp32).
using f16x8v = __attribute__((ext_vector_type(8))) _Float16;
using f16x2v = __attribute__((ext_vector_type(2))) _Float16;
__device__ __forceinline__ int vq_expo(float m) {                  // exponent e with m in [2^e, 2^(e+1)), clamped to >= -60
  const int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
  return e < -60 ? -60 : e;
}
__device__ __forceinline__ void vq_split2(float x0, float x1, int k, unsigned& h, unsigned& l) {
  const float y0 = __builtin_ldexpf(x0, k), y1 = __builtin_ldexpf(x1, k);
  f16x2v hv; hv[0] = (_Float16)y0; hv[1] = (_Float16)y1;
  h = __builtin_bit_cast(unsigned, hv);
  f16x2v lv; lv[0] = (_Float16)(y0 - (float)hv[0]); lv[1] = (_Float16)(y1 - (float)hv[1]);
  l = __builtin_bit_cast(unsigned, lv);
}
// Wp[piece][code][d] fp16 of W 2^kw (two per 32-bit word)
__global__ void vq_wsplit2_kernel(const float* __restrict__ W, int k, int d, const int* __restrict__ wmax_bits, unsigned* __restrict__ Wp) {
  const long pairs = (long)k * d / 2;
  const int kw = 14 - vq_expo(__int_as_float(wmax_bits[1]));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (long)gridDim.x * blockDim.x) {
    unsigned h, l;
    vq_split2(W[2 * i], W[2 * i + 1], kw, h, l);
    Wp[i] = h; Wp[pairs + i] = l;
  }
}

// Wp[piece][code][d] bf16 (two per 32-bit word)
__global__ void vq_wsplit_kernel(const float* __restrict__ W, int k, int d, unsigned* __restrict__ Wp) {
  const long pairs = (long)k * d / 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (long)gridDim.x * blockDim.x) {
    unsigned h, m, l;
    vq_split3(W[2 * i], W[2 * i + 1], h, m, l);
    Wp[i] = h; Wp[pairs + i] = m; Wp[2 * pairs + i] = l;
  }
}

// CAND = false: the sweep over every latent row (blockIdx * 256 + ...): idx, and for an ambiguous row a slot in
//   `flagged` plus its minimum and band in fminb[2 slot], [2 slot + 1].
// CAND = true: the same sweep over the FLAGGED rows only (row ids from `flagged`, one slot per lane): instead
//   of the argmin it lists, per row, every code whose distance lies within the band of the row's minimum --
//   the only codes the reference's arithmetic can prefer -- into cand[slot][VQ_CAP] (ccount[slot] entries;
//   more than VQ_CAP: the row keeps count > VQ_CAP and takes the all-codes re-check).  The distances are
//   the first pass's to the last bit (same fragments, same MFMA order).
constexpr int VQ_CAP = 16;
// NP = 3: six bf16 products of the exact three-way split (matmul mode 2); NP = 2: three fp16 products of the scaled two-way
// split (matmul mode 3, see vq_split2) -- same tiles, same bookkeeping, half the MFMAs and a narrower band.
template <int D, bool CAND, int NP = 3>
__global__ __launch_bounds__(512, 2) void vq_mfma_x3_kernel(
    const float* __restrict__ z, const uint4* __restrict__ Wp, const float* __restrict__ wn,
    const int* __restrict__ wmax_bits, int B, int T, int k,
    int32_t* __restrict__ idx, int32_t* __restrict__ flagged, int32_t* __restrict__ nflag,
    float* __restrict__ fminb, int32_t* __restrict__ cand, int32_t* __restrict__ ccount) {
  constexpr int TILE = 64, KS = D / 16;              // K steps of 16 c
  constexpr int ROW = D / 8 + 1;                     // 16-byte words per LDS row (+1: conflict-free fragment reads)
  constexpr int WPR = D / 8;                         // 16-byte words per code row in global memory
  extern __shared__ uint4 wt[];                      // [2][NP][TILE][ROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const long N = (long)B * T;
  long n = (long)blockIdx.x * 256 + wave * 32 + li;            // this lane's latent column (CAND: its slot)
  const long slot = n;
  float thr = 0.f;                                             // CAND: minimum + band of this row
  bool ok_row = n < N;
  if constexpr (CAND) {
    ok_row = slot < (long)(*nflag);
    if (!__syncthreads_or(ok_row ? 1 : 0)) return;             // this workgroup's slots are all beyond the list
    n = ok_row ? (long)flagged[slot] : 0;
    thr = ok_row ? fminb[2 * slot] + fminb[2 * slot + 1] : -INFINITY;
  }
  // B fragment of K step s: lane (column li, half lk) holds c = 16 s + 8 lk .. + 7, split in three
  uint4 zh[KS], zm[NP == 3 ? KS : 1], zl[KS];
  float zn = 0.f;
  float cn = -2.f;                                             // distance = wn[j] + cn * accumulator (NP = 2: carries the way back from the scales)
  {
    const bool ok = ok_row;
    const long bb = ok ? n / T : 0, t = ok ? n % T : 0;
    const float* zp = z + (bb * D) * T + t;
    if constexpr (NP == 3) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = ok ? zp[(long)(16 * s + 8 * lk + e) * T] : 0.f;
          zn = fmaf(v[e], v[e], zn);
        }
        vq_split3(v[0], v[1], zh[s].x, zm[s].x, zl[s].x);
        vq_split3(v[2], v[3], zh[s].y, zm[s].y, zl[s].y);
        vq_split3(v[4], v[5], zh[s].z, zm[s].z, zl[s].z);
        vq_split3(v[6], v[7], zh[s].w, zm[s].w, zl[s].w);
      }
    } else {
      // the row's own power of two: its largest entry (both halves of the column) into [2^14, 2^15)
      float zmx = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = ok ? zp[(long)(16 * s + 8 * lk + e) * T] : 0.f;
          zn = fmaf(v[e], v[e], zn);
          zmx = fmaxf(zmx, fabsf(v[e]));
        }
        zh[s] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));   // parked: split below, once the scale is known
        zl[s] = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
      }
      zmx = fmaxf(zmx, __shfl_xor(zmx, 32, 64));
      const int kz = 14 - vq_expo(zmx), kw = 14 - vq_expo(__int_as_float(wmax_bits[1]));
      cn = -__builtin_ldexpf(2.f, -(kz + kw));
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const uint4 a = zh[s], bq = zl[s];
        vq_split2(__uint_as_float(a.x), __uint_as_float(a.y), kz, zh[s].x, zl[s].x);
        vq_split2(__uint_as_float(a.z), __uint_as_float(a.w), kz, zh[s].y, zl[s].y);
        vq_split2(__uint_as_float(bq.x), __uint_as_float(bq.y), kz, zh[s].z, zl[s].z);
        vq_split2(__uint_as_float(bq.z), __uint_as_float(bq.w), kz, zh[s].w, zl[s].w);
      }
    }
    zn += __shfl_xor(zn, 32, 64);           // both halves of the column
  }
  // staging: a tile is 3 pieces x 64 codes x WPR words; 512 threads
  constexpr int WORDS = NP * TILE * WPR, PER_THREAD = (WORDS + 511) / 512;
  uint4 st[PER_THREAD];
  const long piece_words = (long)k * WPR;
  auto load_tile = [&](int jt) {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int f = tid + 512 * i;
      const int p = f / (TILE * WPR), r = (f / WPR) % TILE, c = f % WPR;
      const int j = jt * TILE + r;
      st[i] = (f < WORDS && j < k) ? Wp[p * piece_words + (long)j * WPR + c] : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_tile = [&](int buf) {
    uint4* dst = wt + (size_t)buf * NP * TILE * ROW;
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int f = tid + 512 * i;
      if (f < WORDS) {
        const int p = f / (TILE * WPR), r = (f / WPR) % TILE, c = f % WPR;
        dst[(p * TILE + r) * ROW + c] = st[i];
      }
    }
  };
  float m1 = INFINITY, m2 = INFINITY;
  int i1 = 0x7fffffff;
  const int ntile = (k + TILE - 1) / TILE;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  // |w_j|^2 of a tile's codes, as this lane's accumulator rows want them: (half h, row group q) = the four consecutive codes
  // 64 jt + 32 h + 8 q + 4 lk .. + 3.  Requested at the TOP of a tile, in front of its MFMAs: round 5 read wn[j] where it was used,
  // behind a branch per distance (`j < k`) -- 32 branches and 32 dependent L1 round trips per tile and wave beside 48 MFMAs, the
  // matrix pipe 29 % busy at configs[3].  Codes beyond k (the last tile of a k that is no multiple of 64) get +inf: their
  // accumulator rows are zero (load_tile), their distance +inf, and they never win.  wn4: k % 4 == 0 and wn 16-byte aligned.
  const bool wn4 = (k & 3) == 0 && (reinterpret_cast<uintptr_t>(wn) & 15) == 0;
  for (int jt = 0; jt < ntile; ++jt) {
    const int cur = jt & 1;
    if (jt + 1 < ntile) load_tile(jt + 1);
    float wv[2][16];
    if (wn4 && jt * TILE + TILE <= k) {                       // (wave-uniform)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 w4 = *reinterpret_cast<const float4*>(wn + jt * TILE + h * 32 + 8 * q + 4 * lk);
          wv[h][4 * q] = w4.x; wv[h][4 * q + 1] = w4.y; wv[h][4 * q + 2] = w4.z; wv[h][4 * q + 3] = w4.w;
        }
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = jt * TILE + h * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const float w = wn[min(j, k - 1)];                  // unconditional (a load behind a branch would wait for it where it stands)
          wv[h][r] = j < k ? w : INFINITY;
        }
    }
    const uint4* base = wt + (size_t)cur * NP * TILE * ROW;
    f32x16 a0, a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
    if constexpr (NP == 2) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        f16x8v w0[2], w1[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          w0[p] = __builtin_bit_cast(f16x8v, base[(p * TILE + li) * ROW + 2 * s + lk]);
          w1[p] = __builtin_bit_cast(f16x8v, base[(p * TILE + 32 + li) * ROW + 2 * s + lk]);
        }
        const f16x8v bh = __builtin_bit_cast(f16x8v, zh[s]), bl = __builtin_bit_cast(f16x8v, zl[s]);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[1], bh, a0, 0, 0, 0);      // small products first
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[1], bh, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[0], bl, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[0], bl, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[0], bh, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[0], bh, a1, 0, 0, 0);
      }
    } else
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      bf16x8v w0[3], w1[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        w0[p] = __builtin_bit_cast(bf16x8v, base[(p * TILE + li) * ROW + 2 * s + lk]);
        w1[p] = __builtin_bit_cast(bf16x8v, base[(p * TILE + 32 + li) * ROW + 2 * s + lk]);
      }
      const bf16x8v bh = __builtin_bit_cast(bf16x8v, zh[s]), bm = __builtin_bit_cast(bf16x8v, zm[s]),
                    bl = __builtin_bit_cast(bf16x8v, zl[s]);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[2], bh, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[2], bh, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[0], bl, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[0], bl, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[1], bm, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[1], bm, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[1], bh, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[1], bh, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[0], bm, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[0], bm, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[0], bh, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[0], bh, a1, 0, 0, 0);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = jt * TILE + h * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        {
          const float v = fmaf(cn, h ? a1[r] : a0[r], wv[h][r]);
          if constexpr (CAND) {
            if (v <= thr) {                                    // (never for lanes without a row: thr = -inf)
              const int c = atomicAdd(&ccount[slot], 1);
              if (c < VQ_CAP) cand[slot * VQ_CAP + c] = j;
            }
          } else {
            // (m1 <= m2) and a new value: the index moves only on a strict improvement (j ascends within
            // a lane, so the first minimum is kept); runner-up = the median of the three -- 5 VALU per
            // distance instead of the 7 of the compare / select ladder, beside 96 MFMAs per 32 distances
            i1 = v < m1 ? j : i1;
            m2 = __builtin_amdgcn_fmed3f(m1, m2, v);
            m1 = fminf(m1, v);
          }
        }
      }
    if (jt + 1 < ntile) store_tile(cur ^ 1);
    __syncthreads();
  }
  if constexpr (CAND) return;
  {
    const float om1 = __shfl_xor(m1, 32, 64);
    const float om2 = __shfl_xor(m2, 32, 64);
    const int oi1 = __shfl_xor(i1, 32, 64);
    if (om1 < m1 || (om1 == m1 && oi1 < i1)) { m2 = fminf(m1, om2); m1 = om1; i1 = oi1; }
    else { m2 = fminf(m2, om1); }
  }
  if (lk == 0 && n < N) {
    const float wmax = __int_as_float(*wmax_bits);
    const float band = (NP == 2 ? 17.f : 24.f) * (float)(D + 4) * 5.9604645e-8f * (zn + wmax);
    idx[n] = i1;
    if (!(m2 - m1 > band)) {
      const int sl = atomicAdd(nflag, 1);
      flagged[sl] = (int32_t)n;
      fminb[2 * sl] = m1; fminb[2 * sl + 1] = band;
    }
  }
}

// Exact re-check proportional to the ambiguity: one thread per flagged row evaluates, in the reference's
// operation order (sequential over c; sub, mul, add individually rounded), only the codes the candidate
// sweep listed, and keeps the smallest distance -- ties to the lowest index, whatever order the list is in.
// Rows with more than VQ_CAP candidates (ties among many codes: a collapsed codebook, duplicate codes) go to
// `overflow` for the all-codes re-check.
__global__ __launch_bounds__(256) void vq_exact_cand_kernel(
    const float* __restrict__ z, const float* __restrict__ W, int d, int T,
    const int32_t* __restrict__ flagged, const int32_t* __restrict__ nflag, const int32_t* __restrict__ cand,
    const int32_t* __restrict__ ccount, int32_t* __restrict__ idx, int32_t* __restrict__ overflow,
    int32_t* __restrict__ noverflow) {
  const long count = (long)(*nflag);
  for (long sl = (long)blockIdx.x * blockDim.x + threadIdx.x; sl < count; sl += (long)gridDim.x * blockDim.x) {
    const long n = flagged[sl];
    const int nc = ccount[sl];
    if (nc > VQ_CAP || nc <= 0) {        // (0 cannot happen: the row's own minimum is within its band)
      overflow[atomicAdd(noverflow, 1)] = (int32_t)n;
      continue;
    }
    const float* zp = z + ((n / T) * d) * (long)T + n % T;
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int q = 0; q < nc; ++q) {
      const int j = cand[sl * VQ_CAP + q];
      const float* wr = W + (long)j * d;
      float acc = 0.f;
      for (int c = 0; c < d; ++c) {
        const float df = __fsub_rn(zp[(long)c * T], wr[c]);
        acc = __fadd_rn(acc, __fmul_rn(df, df));
      }
      if (acc < best || (acc == best && j < bi) || bi == 0x7fffffff) { best = acc; bi = j; }
    }
    idx[n] = bi;
  }
}

// Exact re-evaluation, batched: a workgroup takes R queued rows at a time and streams the
// codebook through LDS in [256 codes][32 c] chunks (coalesced loads, conflict-free reads);
// thread j owns code j of the chunk and carries the R running sums in the reference's order
// (sequential over c; sub, mul, add individually rounded).  Codebook traffic drops R-fold
// versus one row per workgroup.
template <int R>
__global__ __launch_bounds__(256) void vq_exact_batched_kernel(
    const float* __restrict__ z, const float* __restrict__ W, int B, int d, int T, int k,
    const int32_t* __restrict__ list, const int32_t* __restrict__ nlist, int32_t* __restrict__ idx) {
  constexpr int CH = 32, WPC = CH + 1;
  extern __shared__ float smem[];
  float* zs = smem;                      // [R][d]
  float* wt = zs + R * d;                // [256][WPC]
  float* rv = wt + 256 * WPC;            // [256]
  int* ri = (int*)(rv + 256);            // [256]
  const int tid = threadIdx.x;
  const long count = (long)(*nlist);
  for (long q0 = (long)blockIdx.x * R; q0 < count; q0 += (long)gridDim.x * R) {
    __syncthreads();
    for (int e = tid; e < R * d; e += 256) {
      const int r = e / d, c = e % d;
      float v = 0.f;
      if (q0 + r < count) { const long n = list[q0 + r]; v = z[((n / T) * d + c) * T + n % T]; }
      zs[e] = v;
    }
    float best[R];
    int bi[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { best[r] = INFINITY; bi[r] = 0x7fffffff; }
    for (int j0 = 0; j0 < k; j0 += 256) {
      float acc[R];
#pragma unroll
      for (int r = 0; r < R; ++r) acc[r] = 0.f;
      for (int c0 = 0; c0 < d; c0 += CH) {
        __syncthreads();
        // coalesced: 8 threads cover one code's 32-float (128-B) chunk as float4
        for (int f = tid; f < 256 * (CH / 4); f += 256) {
          const int r = f / (CH / 4), c4 = f % (CH / 4);
          const int j = j0 + r, c = c0 + 4 * c4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (j < k) {
            const float* wp = W + (long)j * d + c;
            if (c + 3 < d && (d & 3) == 0) v = *reinterpret_cast<const float4*>(wp);
            else { if (c < d) v.x = wp[0]; if (c + 1 < d) v.y = wp[1]; if (c + 2 < d) v.z = wp[2]; if (c + 3 < d) v.w = wp[3]; }
          }
          float* p = wt + r * WPC + 4 * c4;
          p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
        }
        __syncthreads();
        const float* wr = wt + tid * WPC;
        const int cmax = min(CH, d - c0);
        for (int c = 0; c < cmax; ++c) {
          const float w = wr[c];
#pragma unroll
          for (int r = 0; r < R; ++r) {
            const float df = __fsub_rn(zs[r * d + c0 + c], w);
            acc[r] = __fadd_rn(acc[r], __fmul_rn(df, df));
          }
        }
      }
      const int j = j0 + tid;
      if (j < k) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          if (acc[r] < best[r] || bi[r] == 0x7fffffff) { best[r] = acc[r]; bi[r] = j; }   // j ascends
      }
    }
    for (int r = 0; r < R; ++r) {
      if (q0 + r >= count) break;
      __syncthreads();
      rv[tid] = best[r];
      ri[tid] = bi[r];
      __syncthreads();
      for (int s = 128; s >= 1; s >>= 1) {
        if (tid < s) {
          const float ov = rv[tid + s], mv = rv[tid];
          const int oi = ri[tid + s], mi = ri[tid];
          if (oi != 0x7fffffff && (mi == 0x7fffffff || ov < mv || (ov == mv && oi < mi))) { rv[tid] = ov; ri[tid] = oi; }
        }
        __syncthreads();
      }
      if (tid == 0) idx[list[q0 + r]] = ri[0];
    }
  }
}

// exact evaluation in the reference's order for the queued rows (list != null) or
// for every row (list == null).  One 256-thread block per row.
__global__ __launch_bounds__(256) void vq_exact_kernel(
    const float* __restrict__ z, const float* __restrict__ W, int B, int d, int T, int k,
    const int32_t* __restrict__ list, const int32_t* __restrict__ nlist, int32_t* __restrict__ idx) {
  extern __shared__ float smem[];
  float* zs = smem;                         // [d]
  float* rv = smem + d;                     // [256]
  int* ri = (int*)(rv + 256);               // [256]
  const long N = (long)B * T;
  const long count = list ? (long)(*nlist) : N;
  for (long q = blockIdx.x; q < count; q += gridDim.x) {
    const long n = list ? (long)list[q] : q;
    const long bb = n / T, t = n % T;
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) zs[c] = z[(bb * d + c) * T + t];
    __syncthreads();
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < k; j += 256) {
      const float* wr = W + (long)j * d;
      float acc = 0.f;
      for (int c = 0; c < d; ++c) {
        const float df = __fsub_rn(zs[c], wr[c]);
        acc = __fadd_rn(acc, __fmul_rn(df, df));
      }
      // j ascends per thread, so strict '<' keeps this thread's first minimum
      if (acc < best || bi == 0x7fffffff) { best = acc; bi = j; }
    }
    rv[threadIdx.x] = best;
    ri[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
      if (threadIdx.x < s) {
        const float ov = rv[threadIdx.x + s];
        const int oi = ri[threadIdx.x + s];
        const float mv = rv[threadIdx.x];
        const int mi = ri[threadIdx.x];
        // numpy.argmin: first minimum; NaN never occurs for finite inputs
        if (oi != 0x7fffffff && (mi == 0x7fffffff || ov < mv || (ov == mv && oi < mi))) {
          rv[threadIdx.x] = ov; ri[threadIdx.x] = oi;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) idx[n] = ri[0];
  }
}

__global__ void vq_gather_kernel(const float* __restrict__ W, const int32_t* __restrict__ idx,
                                 int B, int d, int T, float* __restrict__ e) {
  const long total = (long)B * d * T;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const long r = i / T;
    const int c = (int)(r % d);
    const long bb = r / d;
    e[i] = W[(long)idx[bb * T + t] * d + c];
  }
}

// The same gather through LDS: e is (B, d, T) with time contiguous, a codebook row is d contiguous floats -- the kernel above
// reads a different row per lane (64 cache lines per wave instruction for 256 bytes of payload: 0.63 ms for the 1 M rows of
// configs[3]).  A workgroup takes 64 consecutive t of one b: rows arrive coalesced (lanes along the row), go to LDS at pitch
// d + 1 and leave transposed, lanes along t (both LDS passes conflict-free).  Dynamic LDS: 64 (d + 1) floats.
__global__ __launch_bounds__(256) void vq_gather_tile_kernel(const float* __restrict__ W, const int32_t* __restrict__ idx,
                                                             int B, int d, int T, float* __restrict__ e) {
  extern __shared__ float gt_tile[];
  __shared__ int rows[64];
  const int b = blockIdx.y, t0 = blockIdx.x * 64, nt = min(64, T - t0), tid = threadIdx.x;
  if (tid < 64) rows[tid] = tid < nt ? idx[(long)b * T + t0 + tid] : 0;
  __syncthreads();
  const int pitch = d + 1;
  for (int i = tid; i < nt * d; i += 256) {
    const int r = i / d, c = i - r * d;
    gt_tile[r * pitch + c] = W[(long)rows[r] * d + c];
  }
  __syncthreads();
  const int t = tid & 63;
  if (t < nt)
    for (int c = tid >> 6; c < d; c += 4) e[((long)b * d + c) * T + t0 + t] = gt_tile[t * pitch + c];
}

// ---------------------------------------------------------------------------
// Large-N codebook gradient (B*T' beyond the scan kernel's reach, e.g. the configs[3] stress
// shape): gW = onehot(idx)^T gy accumulated in float64, rounded once (utils.py:222-229).
// Deterministic by construction -- no floating-point atomics: a STABLE counting sort of the row
// ids by code (per-chunk histograms -> chunk-major offsets -> one wave per chunk places its rows
// in ascending order), then every code sums its rows in ascending row order; a long row list is
// cut into VQ_GW_SPLIT fixed pieces whose float64 partials are combined in piece order.
// ---------------------------------------------------------------------------
constexpr int VQ_GW_CHUNK = 16384;     // rows per sort chunk (one wave places a chunk)
constexpr int VQ_GW_SPLIT = 8;         // max pieces of one code's row list
constexpr int VQ_GW_PIECE = 2048;      // rows per piece before a list is cut further

__global__ __launch_bounds__(256) void vq_gw_hist_kernel(const int32_t* __restrict__ idx, long N, int k,
                                                         int32_t* __restrict__ hist) {
  // hist[chunk][j] = rows of this chunk quantised to j (integer atomics: exact, order-free)
  int32_t* h = hist + (long)blockIdx.x * k;
  const long lo = (long)blockIdx.x * VQ_GW_CHUNK;
  const long hi = lo + VQ_GW_CHUNK < N ? lo + VQ_GW_CHUNK : N;
  for (long n = lo + threadIdx.x; n < hi; n += blockDim.x) atomicAdd(&h[idx[n]], 1);
}

__global__ __launch_bounds__(1024) void vq_gw_offsets_kernel(int32_t* __restrict__ hist, int nchunk, int k,
                                                             int32_t* __restrict__ base) {
  // in place: hist[chunk][j] -> first slot of (chunk, j) in the sorted list; base[j], base[k] = N
  __shared__ int32_t part[1024];
  __shared__ int32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int j0 = 0; j0 < k; j0 += 1024) {
    const int j = j0 + threadIdx.x;
    int32_t tot = 0;
    if (j < k)
      for (int c = 0; c < nchunk; ++c) tot += hist[(long)c * k + j];
    part[threadIdx.x] = tot;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {       // inclusive scan of the 1024 totals
      const int32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
    const int32_t start = carry + part[threadIdx.x] - tot;
    if (j < k) {
      base[j] = start;
      int32_t run = start;
      for (int c = 0; c < nchunk; ++c) {
        const int32_t h = hist[(long)c * k + j];
        hist[(long)c * k + j] = run;
        run += h;
      }
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) base[k] = carry;
}

__global__ __launch_bounds__(64) void vq_gw_place_kernel(const int32_t* __restrict__ idx, long N, int k,
                                                         int32_t* __restrict__ cursor_all,
                                                         int32_t* __restrict__ sorted) {
  // one wave owns one chunk and its cursor row: rows are visited in ascending order, 64 at a
  // time; lanes holding the same code take consecutive slots in lane (= row) order
  // volatile: a wave-uniform address would otherwise be read through the scalar cache, which
  // does not see this wave's own vector stores
  volatile int32_t* cursor = cursor_all + (long)blockIdx.x * k;
  const int lane = threadIdx.x;
  const long lo = (long)blockIdx.x * VQ_GW_CHUNK;
  const long hi = lo + VQ_GW_CHUNK < N ? lo + VQ_GW_CHUNK : N;
  for (long g = lo; g < hi; g += 64) {
    const long n = g + lane;
    const bool live = n < hi;
    const int32_t c = live ? idx[n] : -1;
    unsigned long long todo = __ballot(live);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int32_t c0 = __shfl(c, leader);
      const unsigned long long m = __ballot(live && c == c0) & todo;
      const int32_t start = cursor[c0];                 // same address in every lane: one broadcast load
      if (live && c == c0 && ((todo >> lane) & 1ull))
        sorted[start + __popcll(m & ((1ull << lane) - 1ull))] = (int32_t)n;
      __builtin_amdgcn_s_waitcnt(0);                    // the load above precedes the owner's store below
      if (lane == leader) cursor[c0] = start + __popcll(m);
      __builtin_amdgcn_s_waitcnt(0);                    // ... and is complete before the next group's load
      todo &= ~m;
    }
  }
}

__global__ __launch_bounds__(256) void vq_gw_sum_kernel(const int32_t* __restrict__ sorted,
                                                        const int32_t* __restrict__ base,
                                                        const float* __restrict__ gy, int d, int T,
                                                        double* __restrict__ part64,
                                                        float* __restrict__ gW, int accumulate) {
  // block (j, s): piece s of code j's ascending row list
  const int j = blockIdx.x, s = blockIdx.y;
  const int beg = base[j], cnt = base[j + 1] - beg;
  int S = (cnt + VQ_GW_PIECE - 1) / VQ_GW_PIECE;
  if (S > VQ_GW_SPLIT) S = VQ_GW_SPLIT;
  if (S < 1) S = 1;
  if (s >= S) return;
  const int L = (cnt + S - 1) / S;
  const int lo = beg + s * L;
  int hi = lo + L;
  if (hi > beg + cnt) hi = beg + cnt;
  __shared__ double red[256];
  const int tid = threadIdx.x;
  const int dch = d < 256 ? d : 256;
  const int P = 256 / dch;
  for (int c0 = 0; c0 < d; c0 += dch) {
    const int c = c0 + tid % dch;
    const int g = tid / dch;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (g < P && c < d) {
      const float* gc = gy + (long)c * T;
      int q = lo + g;
      for (; q + 3 * P < hi; q += 4 * P) {
        const int n0 = sorted[q], n1 = sorted[q + P], n2 = sorted[q + 2 * P], n3 = sorted[q + 3 * P];
        a0 += (double)gc[(long)(n0 / T) * d * T + n0 % T];
        a1 += (double)gc[(long)(n1 / T) * d * T + n1 % T];
        a2 += (double)gc[(long)(n2 / T) * d * T + n2 % T];
        a3 += (double)gc[(long)(n3 / T) * d * T + n3 % T];
      }
      for (; q < hi; q += P) { const int n = sorted[q]; a0 += (double)gc[(long)(n / T) * d * T + n % T]; }
    }
    __syncthreads();
    red[tid] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && c < d) {
      double acc = red[tid];
      for (int gg = 1; gg < P; ++gg) acc += red[gg * dch + tid];
      if (S == 1) {
        const float v = (float)acc;
        float* dst = gW + (long)j * d + c;
        *dst = accumulate ? __fadd_rn(*dst, v) : v;
      } else {
        part64[((long)j * VQ_GW_SPLIT + s) * d + c] = acc;
      }
    }
  }
}

__global__ void vq_gw_combine_kernel(const int32_t* __restrict__ base, const double* __restrict__ part64,
                                     int k, int d, float* __restrict__ gW, int accumulate) {
  const long total = (long)k * d;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i / d), c = (int)(i % d);
    const int cnt = base[j + 1] - base[j];
    int S = (cnt + VQ_GW_PIECE - 1) / VQ_GW_PIECE;
    if (S > VQ_GW_SPLIT) S = VQ_GW_SPLIT;
    if (S <= 1) continue;                               // written by the sum kernel itself
    double acc = 0.0;
    for (int s = 0; s < S; ++s) acc += part64[((long)j * VQ_GW_SPLIT + s) * d + c];
    const float v = (float)acc;
    gW[i] = accumulate ? __fadd_rn(gW[i], v) : v;
  }
}

// small-problem form of the codebook gradient (training shape: N = B*T' ~ 2k rows): one
// workgroup per code scans the index list from LDS and sums its rows in fp64 in index
// order -- deterministic, no atomics (the scatter above serialises on hot codes).
__global__ __launch_bounds__(256) void vq_gradw_scan_kernel(const int32_t* __restrict__ idx,
                                                            const float* __restrict__ gy, int B, int d,
                                                            int T, int k, float* __restrict__ gW,
                                                            int accumulate) {
  extern __shared__ int32_t rows[];        // the rows quantised to code j, ascending
  __shared__ int wcnt[4];
  __shared__ int total;
  const int j = blockIdx.x;
  const int N = B * T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) total = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 256) {
    const int n = base + tid;
    const bool hit = n < N && idx[n] == j;          // independent loads: pipelined
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = total;
    for (int w = 0; w < wave; ++w) off += wcnt[w];
    if (hit) rows[off + __popcll(m & ((1ull << lane) - 1ull))] = n;
    __syncthreads();
    if (tid == 0) total += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  const int cnt = total;
  // a collapsed codebook sends most rows to a few codes: split the row list of one code
  // over P thread groups (4 independent accumulators each), combined in a fixed order
  __shared__ double part[256];
  const int dch = d < 256 ? d : 256;
  const int P = 256 / dch;                   // thread groups per channel
  for (int c0 = 0; c0 < d; c0 += dch) {
    const int c = c0 + tid % dch;
    const int g = tid / dch;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (g < P && c < d) {
      const float* gc = gy + (long)c * T;
      int q = g;
      for (; q + 3 * P < cnt; q += 4 * P) {
        const int n0 = rows[q], n1 = rows[q + P], n2 = rows[q + 2 * P], n3 = rows[q + 3 * P];
        a0 += (double)gc[(long)(n0 / T) * d * T + n0 % T];
        a1 += (double)gc[(long)(n1 / T) * d * T + n1 % T];
        a2 += (double)gc[(long)(n2 / T) * d * T + n2 % T];
        a3 += (double)gc[(long)(n3 / T) * d * T + n3 % T];
      }
      for (; q < cnt; q += P) { const int n = rows[q]; a0 += (double)gc[(long)(n / T) * d * T + n % T]; }
    }
    __syncthreads();
    part[tid] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && c < d) {
      double acc = part[tid];
      for (int gg = 1; gg < P; ++gg) acc += part[gg * dch + tid];
      const float v = (float)acc;
      float* dst = gW + (long)j * d + c;
      *dst = accumulate ? __fadd_rn(*dst, v) : v;
    }
  }
}

static int vq_cols_per_block(int dpad) {
  // LDS: dpad*NC + 32*(dpad+1) floats <= ~150 KB
  for (int nw = 4; nw >= 1; nw >>= 1) {
    size_t bytes = ((size_t)dpad * 32 * nw + 32 * (size_t)(dpad + 1)) * 4;
    if (bytes <= 150 * 1024) return nw;
  }
  return 0;
}

}  // namespace vq

using namespace vq;

extern "C" size_t vqvae_vq_workspace_bytes(int B, int d, int T, int k) {
  const size_t N = (size_t)B * T;
  size_t fwd = align_up((size_t)k * 4, 256) + 256 /*wmax+nflag*/ + align_up(N * 4, 256) +
               align_up((size_t)k * d * 6, 256) /*split codebook (matmul mode 2)*/ +
               // candidate re-check: (min, band) and count per flagged row, VQ_CAP candidates each, overflow list
               align_up(N * 8, 256) + align_up(N * 4, 256) + align_up(N * 4 * VQ_CAP, 256) + align_up(N * 4, 256);
  // large-N codebook gradient: part64[k][SPLIT][d] | base[k+1] | cursor[nchunk][k] | sorted[N]
  const size_t nchunk = (N + VQ_GW_CHUNK - 1) / VQ_GW_CHUNK;
  size_t bwd = align_up((size_t)k * VQ_GW_SPLIT * d * 8, 256) + align_up(((size_t)k + 1) * 4, 256) +
               align_up(nchunk * k * 4, 256) + align_up(N * 4, 256);
  return (fwd > bwd ? fwd : bwd) + 256;
}

extern "C" int vqvae_vq_nearest_fwd(const float* z, const float* W, int B, int d, int T, int k,
                                    int mode, int32_t* idx, float* e, int32_t* n_rechecked,
                                    void* ws, size_t ws_bytes, vqvae_stream_t s) {
  VQ_REQUIRE(z && W && idx && ws, "vq_nearest_fwd: null pointer");
  VQ_REQUIRE(B > 0 && d > 0 && T > 0 && k > 0, "vq_nearest_fwd: bad dims");
  if (ws_bytes < vqvae_vq_workspace_bytes(B, d, T, k)) { set_error("vq_nearest_fwd: workspace too small"); return VQVAE_E_WORKSPACE; }
  hipStream_t st = (hipStream_t)s;
  const long N = (long)B * T;
  char* wp = (char*)ws;
  float* wn = (float*)wp; wp += align_up((size_t)k * 4, 256);
  int* wmax_bits = (int*)wp;
  int32_t* nflag = (int32_t*)(wp + 64); wp += 256;
  int32_t* flagged = (int32_t*)wp; wp += align_up((size_t)N * 4, 256);
  unsigned* wsplit = (unsigned*)wp; wp += align_up((size_t)k * d * 6, 256);
  float* fminb = (float*)wp; wp += align_up((size_t)N * 8, 256);
  int32_t* ccount = (int32_t*)wp; wp += align_up((size_t)N * 4, 256);
  int32_t* cand = (int32_t*)wp; wp += align_up((size_t)N * 4 * VQ_CAP, 256);
  int32_t* overflow = (int32_t*)wp;
  int32_t* noverflow = nflag + 1;
  bool cand_path = false;
  ProfScope ps(VQVAE_PROF_VQ_NEAREST, st);
  if (mode == 0) {
    const int dpad = (d + 1) / 2 * 2;
    const int nw = vq_cols_per_block(dpad);
    VQ_REQUIRE(nw > 0, "vq_nearest_fwd: d=%d too large for the LDS-staged MFMA path (max ~1000)", d);
    VQ_CHECK_HIP(hipMemsetAsync(wmax_bits, 0, 256, st));
    // mode 3: three fp16 products (mode 2: six bf16 products)
    const bool x2 = (d == 64 || d == 128) && vqvae_get_matmul_dtype() == 3;
    if (x2) hipLaunchKernelGGL(vq_wnorm_elt_lds_kernel, dim3(cdiv(k, 64)), dim3(64), (size_t)64 * (d + 1) * 4, st, W, k, d, wn, wmax_bits);      // (x2: d is 64 or 128)
    else hipLaunchKernelGGL(vq_wnorm_kernel, dim3(cdiv(k, 256)), dim3(256), 0, st, W, k, d, wn, wmax_bits);
    VQ_LAUNCH_CHECK();
    if ((d == 64 || d == 128) && vqvae_get_matmul_dtype() >= 2) {      // modes 2 and 3: the sweep on the 16-bit matrix pipe
      int nb = (int)(((long)k * d / 2 + 255) / 256);
      if (nb > 2048) nb = 2048;
      if (x2) hipLaunchKernelGGL(vq_wsplit2_kernel, dim3(nb), dim3(256), 0, st, W, k, d, wmax_bits, wsplit);
      else hipLaunchKernelGGL(vq_wsplit_kernel, dim3(nb), dim3(256), 0, st, W, k, d, wsplit);
      VQ_LAUNCH_CHECK();
      const size_t lds = 2 * (x2 ? 2 : 3) * 64 * (size_t)(d / 8 + 1) * 16;
      const unsigned grid = (unsigned)((N + 255) / 256);
      // the candidate pass pays when re-checking a row against all k codes costs more than sweeping it again
      cand_path = k >= 1024 && N >= 1024;      // (round 5: N = 1920 at k = 8192 1.49 -> 1.32 ms with it; from N = 4096 until then)
      // the candidate sweep covers at most N/8 flagged rows per launch geometry; rows beyond are re-checked in full
      const unsigned cgrid = (unsigned)((N / 8 + 255) / 256);
      if (cand_path) VQ_CHECK_HIP(hipMemsetAsync(ccount, 0, (size_t)N * 4, st));
#define VQ_X3_LAUNCH(Dv, NPv)                                                                                   \
      {                                                                                                         \
        VQ_CHECK_HIP(hipFuncSetAttribute((const void*)vq_mfma_x3_kernel<Dv, false, NPv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((vq_mfma_x3_kernel<Dv, false, NPv>), dim3(grid), dim3(512), lds, st, z, (const uint4*)wsplit, wn, wmax_bits, B, T, k, idx, flagged, nflag, fminb, cand, ccount); \
        if (cand_path) {                                                                                        \
          VQ_CHECK_HIP(hipFuncSetAttribute((const void*)vq_mfma_x3_kernel<Dv, true, NPv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
          hipLaunchKernelGGL((vq_mfma_x3_kernel<Dv, true, NPv>), dim3(cgrid), dim3(512), lds, st, z, (const uint4*)wsplit, wn, wmax_bits, B, T, k, idx, flagged, nflag, fminb, cand, ccount); \
        }                                                                                                       \
      }
      if (d == 64) { if (x2) VQ_X3_LAUNCH(64, 2) else VQ_X3_LAUNCH(64, 3) }
      else { if (x2) VQ_X3_LAUNCH(128, 2) else VQ_X3_LAUNCH(128, 3) }
#undef VQ_X3_LAUNCH
      VQ_LAUNCH_CHECK();
    } else if (d == 64 || d == 128) {
      const size_t lds = 2 * 64 * (size_t)(d + 1) * 4;
      const unsigned grid = (unsigned)((N + 127) / 128);
      if (d == 64) {
        VQ_CHECK_HIP(hipFuncSetAttribute((const void*)vq_mfma_reg_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(vq_mfma_reg_kernel<32>, dim3(grid), dim3(256), lds, st, z, W, wn, wmax_bits, B, T, k, idx, flagged, nflag);
      } else {
        VQ_CHECK_HIP(hipFuncSetAttribute((const void*)vq_mfma_reg_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(vq_mfma_reg_kernel<64>, dim3(grid), dim3(256), lds, st, z, W, wn, wmax_bits, B, T, k, idx, flagged, nflag);
      }
      VQ_LAUNCH_CHECK();
    } else {
    const int NC = 32 * nw;
    const size_t lds = ((size_t)dpad * NC + 32 * (size_t)(dpad + 1)) * 4;
    const unsigned grid = (unsigned)((N + NC - 1) / NC);
    if (nw == 4) {
      VQ_CHECK_HIP(hipFuncSetAttribute((const void*)vq_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(vq_mfma_kernel<4>, dim3(grid), dim3(256), lds, st, z, W, wn, wmax_bits, B, d, T, k, dpad, idx, flagged, nflag);
    } else if (nw == 2) {
      VQ_CHECK_HIP(hipFuncSetAttribute((const void*)vq_mfma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(vq_mfma_kernel<2>, dim3(grid), dim3(128), lds, st, z, W, wn, wmax_bits, B, d, T, k, dpad, idx, flagged, nflag);
    } else {
      VQ_CHECK_HIP(hipFuncSetAttribute((const void*)vq_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(vq_mfma_kernel<1>, dim3(grid), dim3(64), lds, st, z, W, wn, wmax_bits, B, d, T, k, dpad, idx, flagged, nflag);
    }
    VQ_LAUNCH_CHECK();
    }
    const int32_t* list = flagged;
    const int32_t* nlist = nflag;
    if (cand_path) {
      // flagged rows: the listed candidates only; rows with too many candidates (or beyond the candidate
      // sweep's reach, which left their count at 0) fall through to the all-codes re-check below
      hipLaunchKernelGGL(vq_exact_cand_kernel, dim3(1024), dim3(256), 0, st, z, W, d, T, flagged, nflag, cand, ccount, idx, overflow, noverflow);
      VQ_LAUNCH_CHECK();
      list = overflow; nlist = noverflow;
    }
    if (N > 8192) {
      // exact re-check of the queued rows, 8 rows per workgroup pass
      constexpr int R = 8;
      const size_t lds2 = ((size_t)R * d + 256 * 33 + 512) * 4;
      VQ_CHECK_HIP(hipFuncSetAttribute((const void*)vq_exact_batched_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      const long maxb = (N + R - 1) / R;
      unsigned g2 = (unsigned)(maxb < 2048 ? maxb : 2048);
      hipLaunchKernelGGL(vq_exact_batched_kernel<R>, dim3(g2), dim3(256), lds2, st, z, W, B, d, T, k, list, nlist, idx);
      VQ_LAUNCH_CHECK();
    } else {
      // the training shapes (a few thousand rows, a few dozen of them queued): the launch is one workgroup's chain of 64 d
      // dependent roundings per row set -- two rows per workgroup spread it over four times the workgroups (38 -> ~12 us at
      // configs[1], in front of the condition embed); the per-row arithmetic, hence the index, is the same
      constexpr int R = 2;
      const size_t lds2 = ((size_t)R * d + 256 * 33 + 512) * 4;
      VQ_CHECK_HIP(hipFuncSetAttribute((const void*)vq_exact_batched_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
      const long maxb = (N + R - 1) / R;
      unsigned g2 = (unsigned)(maxb < 2048 ? maxb : 2048);
      hipLaunchKernelGGL(vq_exact_batched_kernel<R>, dim3(g2), dim3(256), lds2, st, z, W, B, d, T, k, list, nlist, idx);
      VQ_LAUNCH_CHECK();
    }
    if (n_rechecked) VQ_CHECK_HIP(hipMemcpyAsync(n_rechecked, nflag, 4, hipMemcpyDeviceToDevice, st));
  } else {
    const size_t lds2 = ((size_t)d + 512) * 4;
    unsigned g2 = (unsigned)(N < 65535 ? N : 65535);
    hipLaunchKernelGGL(vq_exact_kernel, dim3(g2), dim3(256), lds2, st, z, W, B, d, T, k,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, idx);
    VQ_LAUNCH_CHECK();
    if (n_rechecked) VQ_CHECK_HIP(hipMemsetAsync(n_rechecked, 0, 4, st));
  }
  if (e) {
    const long total = (long)B * d * T;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    // (the training shape's 1 920 rows: one launch of the plain kernel is all latency; d <= 254: the tile's 64 (d + 1) floats stay
    //  inside the 64 KB of dynamic LDS a launch gets without hipFuncSetAttribute -- wider codes take the plain kernel)
    if ((long)B * T >= 16384 && d <= 254 && B <= 65535)
      hipLaunchKernelGGL(vq_gather_tile_kernel, dim3((T + 63) / 64, B), dim3(256), (size_t)64 * (d + 1) * 4, st, W, idx, B, d, T, e);
    else
      hipLaunchKernelGGL(vq_gather_kernel, dim3(nb), dim3(256), 0, st, W, idx, B, d, T, e);
    VQ_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int vqvae_vq_grad_w(const int32_t* idx, const float* gy, int B, int d, int T, int k,
                               float* gW, int accumulate, void* ws, size_t ws_bytes,
                               vqvae_stream_t s) {
  VQ_REQUIRE(idx && gy && gW && ws, "vq_grad_w: null pointer");
  hipStream_t st = (hipStream_t)s;
  if ((long)B * T <= 8192 && (long)B * T * k <= (1L << 26)) {   // small problem: scan form
    hipLaunchKernelGGL(vq_gradw_scan_kernel, dim3(k), dim3(256), (size_t)B * T * 4, st, idx, gy, B, d, T, k, gW, accumulate);
    VQ_LAUNCH_CHECK();
    return 0;
  }
  const long N = (long)B * T;
  VQ_REQUIRE(N < (1L << 31), "vq_grad_w: B*T too large");
  if (ws_bytes < vqvae_vq_workspace_bytes(B, d, T, k)) { set_error("vq_grad_w: workspace too small"); return VQVAE_E_WORKSPACE; }
  const int nchunk = (int)((N + VQ_GW_CHUNK - 1) / VQ_GW_CHUNK);
  char* wp = (char*)ws;
  double* part64 = (double*)wp; wp += align_up((size_t)k * VQ_GW_SPLIT * d * 8, 256);
  int32_t* base = (int32_t*)wp; wp += align_up(((size_t)k + 1) * 4, 256);
  int32_t* cursor = (int32_t*)wp; wp += align_up((size_t)nchunk * k * 4, 256);
  int32_t* sorted = (int32_t*)wp;
  VQ_CHECK_HIP(hipMemsetAsync(cursor, 0, (size_t)nchunk * k * 4, st));
  hipLaunchKernelGGL(vq_gw_hist_kernel, dim3(nchunk), dim3(256), 0, st, idx, N, k, cursor);
  VQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(vq_gw_offsets_kernel, dim3(1), dim3(1024), 0, st, cursor, nchunk, k, base);
  VQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(vq_gw_place_kernel, dim3(nchunk), dim3(64), 0, st, idx, N, k, cursor, sorted);
  VQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(vq_gw_sum_kernel, dim3(k, VQ_GW_SPLIT), dim3(256), 0, st, sorted, base, gy, d, T, part64, gW, accumulate);
  VQ_LAUNCH_CHECK();
  const long n = (long)k * d;
  int nb2 = (int)((n + 255) / 256);
  if (nb2 > 4096) nb2 = 4096;
  hipLaunchKernelGGL(vq_gw_combine_kernel, dim3(nb2), dim3(256), 0, st, base, part64, k, d, gW, accumulate);
  VQ_LAUNCH_CHECK();
  return 0;
}
