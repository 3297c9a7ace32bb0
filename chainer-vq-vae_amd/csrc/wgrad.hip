// wgrad.hip -- backward-weight of every conv: contraction over the flattened (batch, time) axis, split-K into deterministic
// partial slabs (wgrad3_kernel: modes 1-3; wgrad3_dma_kernel: both operands stored in MFMA form -- pre-split in mode 3, bf16 in
// mode 1 -- and brought to LDS by LDS-DMA; wgrad2_kernel: mode 0; wgrad_kernel: strided shapes), summed in a fixed order by
// wgrad_reduce_kernel; plan_wgrad chooses the splits, launch_wgrad the kernel.
#include "gemm_common.h"

namespace vq {

// Dev aid (-DVQ_PHASE_TIMING, tools/experiments/wphases.py; never in the product build): where a step of the pre-split
// weight-gradient loops goes -- thread 0 of every workgroup adds its s_memtime differences to g_wphase[kernel][phase].
#ifdef VQ_PHASE_TIMING
__device__ unsigned long long g_wphase[2][8];
__device__ unsigned long long g_wwave[16][4];
#define W3_T(v) const unsigned long long v = __builtin_readcyclecounter()
#define W3_ACC(S, A, B) S += (B) - (A)
#else
#define W3_T(v)
#define W3_ACC(S, A, B)
#endif

template <bool BF16>
__global__ __launch_bounds__(NT, WGRAD_WAVES_PER_EU) void wgrad_kernel(const WgradArgs a) {
  __shared__ float As[BM][WP];
  __shared__ float Bs[BN][WP];
  if (a.skip_flag != nullptr && *a.skip_flag != 0) return;
  const int tile = blockIdx.x;
  const int mt = tile % a.ntile_m;
  const int ntg = tile / a.ntile_m;
  int s = 0;
#pragma unroll
  for (int i = 1; i < MAXSEG; ++i)
    if (i < a.nseg && ntg >= a.seg[i].tile0) s = i;
  const WSeg& sg = a.seg[s];
  const int n0 = (ntg - sg.tile0) * BN;
  const int m0 = mt * BM;
  const int split = blockIdx.y;
  const int g0 = split * a.steps_per_split;
  const int g1 = min(a.B * a.steps_per_b, g0 + a.steps_per_split);
  int b = g0 / a.steps_per_b;
  int tb = (g0 - b * a.steps_per_b) * WBK;
  const int tend = a.Tout;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  // staging roles.  scalar: column k = l_k, rows l_r + 8i (i < 16), one dword per load;
  // vector (rows 16-B aligned, window inside the row): 4 consecutive k = v_c4.., rows
  // v_row + 32i (i < 4), one dwordx4 per load -- 4x fewer VMEM instructions per tile.
  const int l_k = tid & 31, l_r = tid >> 5;
  const int v_row = tid >> 3, v_c4 = (tid & 7) * 4;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) bsum[i] = 0.f;

  const float* gyb = (sg.gy ? sg.gy : a.gy) + (long)b * a.gy_bstride;
  const float* xb = sg.x + (long)b * sg.x_bstride;
  auto advance = [&]() {            // next K step of the flattened (b, t) axis
    tb += WBK;
    if (tb >= tend) { tb = 0; ++b; gyb += a.gy_bstride; xb += sg.x_bstride; }
  };
  const bool do_bias = (ntg == sg.tile0) && (a.bslabs != nullptr) && (sg.gb || sg.gb2 || (s == 0 && a.ngbl > 0));
  const bool avec = a.avec != 0;
  const bool bvec = sg.vec != 0;

  float ra[16], rbv[16];
  auto load = [&](int tb) {
    if (avec) {
      const int t = tb + v_c4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + v_row + 32 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < tend && m < a.M) v = *reinterpret_cast<const float4*>(gyb + (long)m * a.Tout + t);
        ra[4 * i] = v.x; ra[4 * i + 1] = v.y; ra[4 * i + 2] = v.z; ra[4 * i + 3] = v.w;
      }
    } else {
      const int t = tb + l_k;
      const bool tok = t < tend;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = m0 + l_r + 8 * i;
        ra[i] = (tok && m < a.M) ? gyb[(long)m * a.Tout + t] : 0.f;
      }
    }
    if (bvec) {
      const int t = tb + v_c4;
      const int tin = t + sg.toff;                     // tmul == 1, tdiv == 1
      const bool inb = t < tend;
      const bool whole = inb && tin >= 0 && tin + 3 < sg.Tin;
      const bool part = inb && !whole && tin + 3 >= 0 && tin < sg.Tin;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ci = n0 + v_row + 32 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* src = xb + (long)ci * sg.x_cstride + tin;
        if (ci < sg.cin) {
          if (whole) {
            v = *reinterpret_cast<const float4*>(src);
          } else if (part) {                           // the window crosses the row's first / last sample
            if (tin >= 0 && tin < sg.Tin) v.x = src[0];
            if (tin + 1 >= 0 && tin + 1 < sg.Tin) v.y = src[1];
            if (tin + 2 >= 0 && tin + 2 < sg.Tin) v.z = src[2];
            if (tin + 3 >= 0 && tin + 3 < sg.Tin) v.w = src[3];
          }
        }
        rbv[4 * i] = v.x; rbv[4 * i + 1] = v.y; rbv[4 * i + 2] = v.z; rbv[4 * i + 3] = v.w;
      }
    } else {
      const int t = tb + l_k;
      const int tnum = t * sg.tmul + sg.toff;
      bool xok = t < tend && tnum >= 0;
      int tin = tnum;
      if (sg.tdiv > 1) { xok = xok && (tnum % sg.tdiv == 0); tin = tnum / sg.tdiv; }
      xok = xok && tin < sg.Tin;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = n0 + l_r + 8 * i;
        rbv[i] = (xok && ci < sg.cin) ? xb[(long)ci * sg.x_cstride + tin] : 0.f;
      }
    }
  };

  if (g0 < g1) load(tb);
  for (int g = g0; g < g1; ++g) {
    __syncthreads();
    if (BF16) {
      // bf16 image, k contiguous: row pitch WPB elements (80 B, 16-byte aligned rows, conflict-optimal
      // for the 16-byte fragment reads); operands are rounded (RNE) once, here, instead of per fragment
      __bf16* Ab = reinterpret_cast<__bf16*>(&As[0][0]);
      __bf16* Bb = reinterpret_cast<__bf16*>(&Bs[0][0]);
      if (avec) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bf16x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (__bf16)ra[4 * i + j];
          *reinterpret_cast<bf16x4*>(Ab + (v_row + 32 * i) * WPB + v_c4) = v;
          bsum[i] += (ra[4 * i] + ra[4 * i + 1]) + (ra[4 * i + 2] + ra[4 * i + 3]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) { Ab[(l_r + 8 * i) * WPB + l_k] = (__bf16)ra[i]; bsum[i] += ra[i]; }
      }
      if (bvec) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bf16x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = (__bf16)rbv[4 * i + j];
          *reinterpret_cast<bf16x4*>(Bb + (v_row + 32 * i) * WPB + v_c4) = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) Bb[(l_r + 8 * i) * WPB + l_k] = (__bf16)rbv[i];
      }
    } else {
    if (avec) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) As[v_row + 32 * i][v_c4 + j] = ra[4 * i + j];
        bsum[i] += (ra[4 * i] + ra[4 * i + 1]) + (ra[4 * i + 2] + ra[4 * i + 3]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) { As[l_r + 8 * i][l_k] = ra[i]; bsum[i] += ra[i]; }
    }
    if (bvec) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Bs[v_row + 32 * i][v_c4 + j] = rbv[4 * i + j];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) Bs[l_r + 8 * i][l_k] = rbv[i];
    }
    }
    __syncthreads();
    if (g + 1 < g1) { advance(); load(tb); }
    if (BF16) {
#pragma unroll
      for (int k16 = 0; k16 < WBK / 16; ++k16) {
        const __bf16* Ab = reinterpret_cast<const __bf16*>(&As[0][0]);
        const __bf16* Bb = reinterpret_cast<const __bf16*>(&Bs[0][0]);
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {        // one 16-byte LDS read per fragment
          af[h] = *reinterpret_cast<const bf16x8*>(Ab + (wm * 64 + h * 32 + li) * WPB + k16 * 16 + 8 * lk);
          bf[h] = *reinterpret_cast<const bf16x8*>(Bb + (wn * 64 + h * 32 + li) * WPB + k16 * 16 + 8 * lk);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int kk = 0; kk < WBK / 2; ++kk) {
      const float a0 = As[wm * 64 + li][kk * 2 + lk];
      const float a1 = As[wm * 64 + 32 + li][kk * 2 + lk];
      const float b0 = Bs[wn * 64 + li][kk * 2 + lk];
      const float b1 = Bs[wn * 64 + 32 + li][kk * 2 + lk];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    }
  }

  float* slab = a.slabs + (((long)split * a.ntile_m + mt) * a.ntile_n + ntg) * (BM * BN);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int col = wn * 64 + ni * 32 + li;
        slab[row * BN + col] = acc[mi][ni][r];
      }
  if (do_bias) {
    float* bs = a.bslabs + (((long)split * a.nseg + s) * a.ntile_m + mt) * BM;
    if (avec) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = bsum[i];
#pragma unroll
        for (int off = 4; off >= 1; off >>= 1) v += __shfl_xor(v, off, 8);
        if ((tid & 7) == 0) bs[v_row + 32 * i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = bsum[i];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off, 32);
        if (l_k == 0) bs[l_r + 8 * i] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// wgrad2_kernel -- the fp32 weight-gradient contraction for stride-1 segments (every conv of the
// decoder): same splits, slabs and fixed-order reduce as wgrad_kernel, rebuilt around 16-byte LDS
// traffic and 4 waves per SIMD.
//   * tile (64*WM) x 128 per 128*WM-thread workgroup (WM = 4: 256 x 128, the activation tile is
//     shared by twice the rows; 2 workgroups = 16 waves per CU), wave tile 64 x 64;
//   * both operands are K-contiguous in HBM (time is the contraction axis) and stay that way in
//     LDS: image [row][16 t] filled by dwordx4 row loads + ds_write_b128 -- no transposing scalar
//     writes.  A lane's MFMA fragment is one ds_read_b128 = 4 consecutive t of its row; the k-th
//     MFMA of a group takes component k of BOTH operands, i.e. the contraction index is visited
//     in the order the fragments deliver it (any order is valid as long as A and B agree);
//   * 16-byte chunk c of row r sits at chunk slot c ^ ((r >> 2) & 3): the 16 lanes one
//     ds_read_b128 cycle serves ({0-3,12-15,20-27}, ...) land on 16 distinct slots of the 256-B
//     bank row (conflict-free reads AND writes);
//   * double-buffered (2 x 24 KB), next step prefetched into registers, ONE barrier per 16-t step
//     (32 MFMAs per wave), <= 128 VGPRs.
// ---------------------------------------------------------------------------
constexpr int W2K = 16;                                   // t per K step
template <int WM>
__global__ __launch_bounds__(128 * WM, 4) void wgrad2_kernel(const WgradArgs a) {
  constexpr int NT2 = 128 * WM, BM2 = 64 * WM;
  constexpr int STAGE = (BM2 + BN) * 4;                   // float4 per stage
  constexpr int NA = BM2 * 4 / NT2, NB = BN * 4 / NT2;    // float4 row loads per thread: 2 and 1 (WM=4) / 2 and 2
  __shared__ float4 lds[2 * STAGE];
  if (a.skip_flag != nullptr && *a.skip_flag != 0) return;
  const int ntm = (a.ntile_m * BM + BM2 - 1) / BM2;       // a.ntile_m counts 128-row slab tiles
  // XCD-aware order (1-D grid): workgroups that run on one XCD at the same time are consecutive
  // tiles of ONE split -- the column tiles of a segment pair share their output-gradient rows and
  // K range, so that operand is fetched into the XCD's L2 once instead of once per column tile
  int logical;
  {
    const int nblk = gridDim.x, id = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int ntiles = ntm * a.ntile_n;
  const int tile = logical % ntiles;
  const int split = logical / ntiles;
  const int ntg = tile % a.ntile_n;                        // column tile fastest: neighbours share gy
  const int mt = tile / a.ntile_n;
  int s = 0;
#pragma unroll
  for (int i = 1; i < MAXSEG; ++i)
    if (i < a.nseg && ntg >= a.seg[i].tile0) s = i;
  const WSeg& sg = a.seg[s];
  const int n0 = (ntg - sg.tile0) * BN;
  const int m0 = mt * BM2;
  // K steps of 16 t: two per WBK step of the split plan
  const int spb = a.steps_per_b * (WBK / W2K);
  const int g0 = split * a.steps_per_split * (WBK / W2K);
  const int g1 = min(a.B * spb, g0 + a.steps_per_split * (WBK / W2K));
  int b = g0 / spb;
  int tb = (g0 - b * spb) * W2K;
  const int Tout = a.Tout;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int s_chunk = tid & 3, s_row = tid >> 2;          // staging role: chunk of 4 t, row (+ NT2/4 per extra load)
  constexpr int RSTEP = NT2 / 4;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) bsum[i] = 0.f;

  const float* gyb = (sg.gy ? sg.gy : a.gy) + (long)b * a.gy_bstride + (long)(m0 + s_row) * Tout + 4 * s_chunk;
  const float* xb = sg.x + (long)b * sg.x_bstride + (long)(n0 + s_row) * sg.x_cstride + 4 * s_chunk + sg.toff;
  const long a_rstep = (long)RSTEP * Tout, b_rstep = (long)RSTEP * sg.x_cstride;
  bool a_ok[NA], b_ok[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) a_ok[i] = (m0 + s_row + RSTEP * i) < a.M;
#pragma unroll
  for (int i = 0; i < NB; ++i) b_ok[i] = (n0 + s_row + RSTEP * i) < sg.cin;
  const bool do_bias = (ntg == sg.tile0) && (a.bslabs != nullptr) && (sg.gb || sg.gb2 || (s == 0 && a.ngbl > 0));

  float4 ra[NA], rb[NB];
  auto load = [&]() {
    const int t = tb + 4 * s_chunk;
    const bool tin_range = t < Tout;                      // Tout % 4 == 0: a group is in or out as a whole
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tin_range && a_ok[i]) ra[i] = *reinterpret_cast<const float4*>(gyb + i * a_rstep + tb);
    }
    const int tin = t + sg.toff;
    const bool whole = tin_range && tin >= 0 && tin + 3 < sg.Tin;
    const bool part = tin_range && !whole && tin + 3 >= 0 && tin < sg.Tin;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (b_ok[i]) {
        const float* src = xb + i * b_rstep + tb;
        if (whole) {
          rb[i] = *reinterpret_cast<const float4*>(src);   // dword-aligned dwordx4: fine on gfx950
        } else if (part) {                                 // the shifted window crosses the row's first / last sample
          if (tin >= 0 && tin < sg.Tin) rb[i].x = src[0];
          if (tin + 1 >= 0 && tin + 1 < sg.Tin) rb[i].y = src[1];
          if (tin + 2 >= 0 && tin + 2 < sg.Tin) rb[i].z = src[2];
          if (tin + 3 >= 0 && tin + 3 < sg.Tin) rb[i].w = src[3];
        }
      }
    }
  };
  auto advance = [&]() {
    tb += W2K;
    if (tb >= spb * W2K) { tb = 0; ++b; gyb += a.gy_bstride; xb += sg.x_bstride; }   // same step count per item as the plan
  };
  // staging destinations (float4 index inside a stage): row r, chunk c -> r*4 + (c ^ ((r>>2)&3)); the
  // extra rows are RSTEP (a multiple of 16) further, which leaves the swizzle term unchanged
  const int st_a = s_row * 4 + (s_chunk ^ ((s_row >> 2) & 3));
  const int st_b = BM2 * 4 + st_a;
  auto store = [&](int stage) {
    float4* base = lds + stage * STAGE;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      base[st_a + i * RSTEP * 4] = ra[i];
      bsum[i] += (ra[i].x + ra[i].y) + (ra[i].z + ra[i].w);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) base[st_b + i * RSTEP * 4] = rb[i];
  };
  // fragment addresses: row = w*64 + t*32 + li, chunk = lk + 2q
  const int swz = (li >> 2) & 3;
  const int fa0 = (wm * 64 + li) * 4 + (lk ^ swz), fa1 = (wm * 64 + li) * 4 + ((lk + 2) ^ swz);
  const int fb0 = BM2 * 4 + (wn * 64 + li) * 4 + (lk ^ swz), fb1 = BM2 * 4 + (wn * 64 + li) * 4 + ((lk + 2) ^ swz);

  if (g0 < g1) { load(); store(0); }
  __syncthreads();
  for (int g = g0; g < g1; ++g) {
    const int cur = (g - g0) & 1;
    const bool more = g + 1 < g1;
    if (more) { advance(); load(); }
    const float4* st = lds + cur * STAGE;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 a0 = st[q ? fa1 : fa0], a1 = st[(q ? fa1 : fa0) + 128];
      const float4 b0 = st[q ? fb1 : fb0], b1 = st[(q ? fb1 : fb0) + 128];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b1.x, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b1.y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b1.z, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc[1][1], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b1.w, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc[1][1], 0, 0, 0);
    }
    if (more) store(cur ^ 1);
    __syncthreads();
  }

  // partial tile -> slab(s): a 256-row tile is two 128-row slab tiles
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int rowb = wm * 64 + mi * 32;                    // wave-uniform
    const int mt_slab = (m0 + rowb) / BM;
    if (mt_slab >= a.ntile_m) continue;
    float* slab = a.slabs + (((long)split * a.ntile_m + mt_slab) * a.ntile_n + ntg) * (BM * BN);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (rowb % BM) + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int col = wn * 64 + ni * 32 + li;
        slab[row * BN + col] = acc[mi][ni][r];
      }
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 1, 4);
      v += __shfl_xor(v, 2, 4);
      const int row = m0 + s_row + RSTEP * i;              // global row
      if (s_chunk == 0 && row < a.ntile_m * BM)
        a.bslabs[(((long)split * a.nseg + s) * a.ntile_m + row / BM) * BM + row % BM] = v;
    }
  }
}

// wgrad3_kernel -- wgrad2_kernel's contraction in matmul mode 2 (fp32 products as six bf16 MFMA
// products of an exact three-way split, see conv_gemm_x3_kernel): same tiles, splits, slabs, loads
// and bias sums; both operands are activations, so both are split while they are staged, into the
// [piece][k-half][row] 16-byte-word images the 32x32x16 fragments read with one ds_read_b128.
// NC = 128-column blocks per workgroup.  NC = 2 (256 x 256 tiles, WM = 4 only): the output-gradient
// tile -- fetched, split and stored once per workgroup, and the same for every column tile of the
// launch -- serves twice the columns; each wave then owns two 64 x 64 blocks 128 columns apart.
// NP = bf16 pieces per operand: 3 (mode 2) or 1 (mode 1: operands rounded to bf16, one product).
// X16 (matmul mode 1): the x operand (the z tensors; the bf16 residual stream x_l, tap-shifted; T a multiple of 16) is
// stored as bf16 -- 8-byte loads of 4 t at any 2-byte alignment (gfx950 serves them: tools/ubench/misaligned_b64.hip),
// staged as they are.
#ifndef W3_LD_AUX
#define W3_LD_AUX 0           // cache policy of the weight-gradient kernels' operand loads (experiment: non-temporal = 2 costs 1.5 ms per step, the column tiles of a launch share their output-gradient rows through L2)
#endif
#ifndef W3_LEAN
#define W3_LEAN 1             // 256 x 128 tiles, six products: compiled for 128 VGPRs (two 8-wave workgroups per CU)
#endif
// G16 (matmul mode 1): the output-gradient operand (gh of a block, g_skip: T a multiple of 16) is stored as bf16 -- the
// same 8-byte loads; its bias sums add the stored (rounded) values.
template <int WM, int NC, int NP, bool X16 = false, bool G16 = false>
__global__ __launch_bounds__(128 * WM, (W3_LEAN && WM == 4 && NC == 1 && NP >= 2) ? 4 : 2) void wgrad3_kernel(const WgradArgs a) {
  static_assert(NC == 1 || WM == 4, "256-column tiles exist for 256-row tiles only");
  static_assert((!X16 && !G16) || NP == 1 || NP == 2, "bf16-stored operands: mode 1; pre-split operands: mode 3");
  // (matmul mode 3, NP = 2: X16 / G16 mark PRE-SPLIT operands -- fp16 hi | lo dwords at the fp32 addresses, see presplit_pair:
  // the same loads and re-alignment as fp32, staged with v_perm_b32 instead of split2; their `amax` words are scale words)
  constexpr bool XB16 = X16 && NP == 1, GB16 = G16 && NP == 1, XPRE = X16 && NP == 2, GPRE = G16 && NP == 2;
  constexpr unsigned XSZ = XB16 ? 2u : 4u, GSZ = GB16 ? 2u : 4u;
  constexpr int NT2 = 128 * WM, BM2 = 64 * WM, BNC = BN * NC;
  constexpr int PA = BM2 + 4, PB = BNC + 4;               // rows of a (piece, k-half) plane; +4: the two k-halves land on different banks
  constexpr int NA = BM2 * 4 / NT2, NB = BNC * 4 / NT2;   // float4 row loads per thread: 2 and 1 or 2 (WM=4) / 2 and 2
  __shared__ uint4 As[2][NP][2][PA];
  __shared__ uint4 Bs[2][NP][2][PB];
  if (a.skip_flag != nullptr && *a.skip_flag != 0) return;
  const int ntm = (a.ntile_m * BM + BM2 - 1) / BM2;       // a.ntile_m counts 128-row slab tiles
  // XCD-aware order (1-D grid): workgroups that run on one XCD at the same time are consecutive
  // tiles of ONE split -- the column tiles of a segment pair share their output-gradient rows and
  // K range, so that operand is fetched into the XCD's L2 once instead of once per column tile
  int logical;
  {
    const int nblk = gridDim.x, id = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int ncolt = NC == 1 ? a.ntile_n : a.ntile_p;       // column tiles of this kernel's width
  const int ntiles = ntm * ncolt;
  const int tile = logical % ntiles;
  const int split = logical / ntiles;
  const int ct = tile % ncolt;                             // column tile fastest: neighbours share gy
  const int mt = tile / ncolt;
  int s = 0;
#pragma unroll
  for (int i = 1; i < MAXSEG; ++i)
    if (i < a.nseg && ct >= (NC == 1 ? a.seg[i].tile0 : a.seg[i].ptile0)) s = i;
  const WSeg& sg = a.seg[s];
  // float32x2: this tile's operand scales 2^ka (gy), 2^kb (x) -- a tile belongs to ONE segment, so each operand takes
  // its own optimum -- and the way back, 2^ku
  [[maybe_unused]] int ka = 0, kb = 0, ku = 0;
  if constexpr (NP == 2) {
    const int eg = amax_expo(amax_load(sg.gy ? sg.amax_gy : a.amax_gy));
    const int ex = amax_expo(sg.amax_x ? amax_load(sg.amax_x) : __builtin_bit_cast(unsigned, sg.amax_x_static));
    ka = 14 - eg; kb = 14 - ex; ku = eg + ex - 28;
  }
  const int ntg = NC == 1 ? ct : sg.tile0 + NC * (ct - sg.ptile0);    // first 128-column slab tile of this workgroup
  const int n0 = (ntg - sg.tile0) * BN;
  const int m0 = mt * BM2;
  // K steps of 16 t: two per WBK step of the split plan
  const int spb = a.steps_per_b * (WBK / W2K);
  const int g0 = split * a.steps_per_split * (WBK / W2K);
  const int g1 = min(a.B * spb, g0 + a.steps_per_split * (WBK / W2K));
  int b = g0 / spb;
  int tb = (g0 - b * spb) * W2K;
  const int Tout = a.Tout;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
  const int s_chunk = tid & 3, s_row = tid >> 2;          // staging role: chunk of 4 t, row (+ NT2/4 per extra load)
  constexpr int RSTEP = NT2 / 4;

  f32x16 acc[2][2], acc2[2][2];           // acc2: the second column block (NC == 2)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }
  float bsum[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) bsum[i] = 0.f;

  // Fetches are buffer loads (see conv_gemm_x3_kernel): descriptors in SGPRs, per-thread offsets fixed
  // for the whole launch (row, 4-t chunk, tap shift), one wave-uniform SGPR offset per operand that walks
  // the flattened (b, t) axis.  An invalid row has an offset beyond the extent and reads 0.  A 16-t step
  // whose shifted window stays inside the input row needs no per-thread arithmetic at all; a step that
  // touches the row's first / last sample (two per row and tap) reads every 4-t group from the clamped
  // in-row position and re-aligns it when it is staged.  Fetches are unconditional, two K steps ahead
  // (two register sets): every wait in the loop is a counted vmcnt.
  constexpr unsigned OOB = 0x80000000u;
  const rsrc_t ra = make_rsrc(sg.gy ? sg.gy : a.gy), rbx = make_rsrc(sg.x);
  unsigned voa[NA], vrow[NB], vobk[NB];
  bool b_ok[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
    voa[i] = (m0 + s_row + RSTEP * i) < a.M ? GSZ * (unsigned)((m0 + s_row + RSTEP * i) * Tout + 4 * s_chunk) : OOB;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    b_ok[i] = (n0 + s_row + RSTEP * i) < sg.cin;
    vrow[i] = XSZ * (unsigned)((n0 + s_row + RSTEP * i) * sg.x_cstride);
    vobk[i] = b_ok[i] ? vrow[i] + 4u * XSZ * (unsigned)s_chunk : OOB;      // interior steps: the tap shift rides in the scalar offset
  }
  const bool do_bias = (ntg == sg.tile0) && (a.bslabs != nullptr) && (sg.gb || sg.gb2 || (s == 0 && a.ngbl > 0));
  const bool ragged = (Tout % WBK) != 0;                 // the last step(s) of a row hold groups beyond Tout (the plan counts WBK = 2 x W2K positions per step: with Tout % 32 == 16 the row's last 16-t step lies wholly beyond it)

  float4 pra[NA], prb[NB], qra[NA], qrb[NB];
  unsigned pvm = 0, qvm = 0;     // 1: this thread's 4-t group lies inside [0, Tout)
  int pbs = 0, qbs = 0;          // clamped start - wanted start of the B group
  int pbt = 0, qbt = 0;          // wanted start (input time) of the B group
  // (every offset handed to a load is non-negative: the scalar part carries tb + toff only on interior steps)
#define W3_FETCH(RA, RB, VM, BS, BT)                                                          \
  {                                                                                            \
    const unsigned soa = GSZ * (unsigned)((long)b * a.gy_bstride + tb);                        \
    const bool interior = tb + sg.toff >= 0 && tb + W2K + sg.toff <= sg.Tin && !ragged;        /* wave-uniform */ \
    const unsigned sob = XSZ * (unsigned)((long)b * sg.x_bstride + (interior ? tb + sg.toff : 0)); \
    VM = (!ragged || tb + 4 * s_chunk < Tout) ? 1u : 0u;                                       \
    unsigned vo_[NB];                                                                          \
    BS = 0; BT = 0;                                                                            \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) vo_[i] = vobk[i];                           \
    if (!interior) {                                   /* wave-uniform, two steps per row and tap: a real branch (the empty asm keeps hipcc from turning the ~25 VALU of the edge path into selects that every step pays) */ \
      asm volatile("");                                                           \
      const int tin = tb + 4 * s_chunk + sg.toff;                                              \
      const bool any = VM != 0u && tin + 3 >= 0 && tin < sg.Tin;                               \
      const int tc = min(max(tin, 0), sg.Tin - 4);                                             \
      BT = tin; BS = tc - tin;                                                                 \
      _Pragma("unroll") for (int i = 0; i < NB; ++i) vo_[i] = (any && b_ok[i]) ? vrow[i] + XSZ * (unsigned)tc : OOB; \
    }                                                                                          \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                           \
      if constexpr (GB16) {                      /* 4 bf16 = 8 bytes, raw, in .x / .y (host: Tout % 16 == 0) */ \
        const uint2 h_ = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(ra, VM ? voa[i] : OOB, soa, W3_LD_AUX)); \
        RA[i] = make_float4(__builtin_bit_cast(float, h_.x), __builtin_bit_cast(float, h_.y), 0.f, 0.f); \
      } else RA[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, VM ? voa[i] : OOB, soa, W3_LD_AUX)); /* a group beyond Tout (ragged last step) must not be fetched: it may lie beyond the tensor */ \
    }                                                                                          \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                           \
      if constexpr (XB16) {                      /* 4 bf16 = 8 bytes, raw, in .x / .y (host: toff == 0, Tout % 16 == 0) */ \
        const uint2 h_ = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(rbx, vo_[i], sob, W3_LD_AUX)); \
        RB[i] = make_float4(__builtin_bit_cast(float, h_.x), __builtin_bit_cast(float, h_.y), 0.f, 0.f); \
      } else RB[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rbx, vo_[i], sob, W3_LD_AUX)); \
    }                                                                                          \
  }
  auto advance = [&]() {
    tb += W2K;
    if (tb >= spb * W2K) { tb = 0; ++b; }   // same step count per item as the plan
  };
  // staging: this thread's 4 consecutive t of a row are half (s_chunk & 1) of the 8-k group
  // (s_chunk >> 1) of that row; split into the three bf16 pieces and written as 8 bytes per piece
  [[maybe_unused]] auto put_pre = [&](uint4* plane0, int prow, const float4 v) {      // four pre-split elements: the pieces are there, two v_perm_b32 per pair
    uint2* d = reinterpret_cast<uint2*>(plane0) + (s_chunk & 1);
    unsigned h0, l0, h1, l1;
    presplit_stage(v.x, v.y, h0, l0);
    presplit_stage(v.z, v.w, h1, l1);
    d[2 * (0 * 2 * prow)] = make_uint2(h0, h1);
    d[2 * (1 * 2 * prow)] = make_uint2(l0, l1);
  };
  auto put = [&](uint4* plane0, int prow, const float4 v, [[maybe_unused]] const int kx) {     // plane0 = &X[stage][0][s_chunk >> 1][row]
    uint2* d = reinterpret_cast<uint2*>(plane0) + (s_chunk & 1);
    if constexpr (NP == 2) {
      unsigned h0, l0, h1, l1;
      split2(v.x, v.y, kx, h0, l0);
      split2(v.z, v.w, kx, h1, l1);
      d[2 * (0 * 2 * prow)] = make_uint2(h0, h1);
      d[2 * (1 * 2 * prow)] = make_uint2(l0, l1);
    } else
    if constexpr (NP == 3) {
      unsigned h0, m0, l0, h1, m1, l1;
      split3(v.x, v.y, h0, m0, l0);
      split3(v.z, v.w, h1, m1, l1);
      d[2 * (0 * 2 * prow)] = make_uint2(h0, h1);
      d[2 * (1 * 2 * prow)] = make_uint2(m0, m1);
      d[2 * (2 * 2 * prow)] = make_uint2(l0, l1);
    } else {
      d[0] = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
    }
  };
  auto put_raw = [&](uint4* plane0, const float4 v) {           // X16: the 8 bytes are the staged image already
    uint2* d = reinterpret_cast<uint2*>(plane0) + (s_chunk & 1);
    d[0] = make_uint2(__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y));
  };
#define W3_STAGE(RA, RB, VM, BS, BT, STAGE, REAL)                                             \
  {                                                                                            \
    const bool real_ = (REAL);          /* evaluated here: the loops below have their own i */ \
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);                                      \
    _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                           \
      const float4 v = VM ? RA[i] : zero4;          /* invalid rows arrived as 0; VM: ragged Tout only */ \
      if constexpr (GPRE) {                                                                    \
        put_pre(&As[STAGE][0][s_chunk >> 1][s_row + i * RSTEP], PA, v);                        \
        if (real_) bsum[i] += (presplit_scaled(v.x) + presplit_scaled(v.y)) + (presplit_scaled(v.z) + presplit_scaled(v.w)); \
      } else                                                                                   \
      if constexpr (GB16) {                                                                    \
        put_raw(&As[STAGE][0][s_chunk >> 1][s_row + i * RSTEP], v);                            \
        const unsigned u0 = __builtin_bit_cast(unsigned, v.x), u1 = __builtin_bit_cast(unsigned, v.y); \
        if (real_) bsum[i] += (__builtin_bit_cast(float, u0 << 16) + __builtin_bit_cast(float, u0 & 0xffff0000u)) + \
                              (__builtin_bit_cast(float, u1 << 16) + __builtin_bit_cast(float, u1 & 0xffff0000u)); \
      } else {                                                                                 \
      put(&As[STAGE][0][s_chunk >> 1][s_row + i * RSTEP], PA, v, ka);                          \
      if (real_) bsum[i] += (v.x + v.y) + (v.z + v.w);                                         \
      }                                                                                        \
    }                                                                                          \
    _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                           \
      float4 v = RB[i];                               /* invalid rows / groups arrived as 0 */ \
      if (__builtin_amdgcn_ballot_w64(BS != 0) != 0ull) {   /* some group of this wave crosses a row end (edge steps only: wave-uniform branch): element e is loaded[e - BS] */ \
        asm volatile("");                                                         \
        float l[4] = {v.x, v.y, v.z, v.w};                                                     \
        if constexpr (XB16) {                         /* four bf16 in .x / .y: one element per register (raw bits, low half) */ \
          const unsigned u0 = __builtin_bit_cast(unsigned, v.x), u1 = __builtin_bit_cast(unsigned, v.y); \
          l[0] = __builtin_bit_cast(float, u0 & 0xffffu); l[1] = __builtin_bit_cast(float, u0 >> 16); \
          l[2] = __builtin_bit_cast(float, u1 & 0xffffu); l[3] = __builtin_bit_cast(float, u1 >> 16); \
        }                                                                                      \
        float o[4];                                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                        \
          const int src = e - BS, tt = BT + e;                                                 \
          float pick = l[0];                                                                   \
          pick = src == 1 ? l[1] : pick; pick = src == 2 ? l[2] : pick; pick = src == 3 ? l[3] : pick; \
          o[e] = (src >= 0 && src < 4 && tt >= 0 && tt < sg.Tin) ? pick : 0.f;                 \
        }                                                                                      \
        if constexpr (XB16) v = make_float4(__builtin_bit_cast(float, __builtin_bit_cast(unsigned, o[0]) | (__builtin_bit_cast(unsigned, o[1]) << 16)), \
                                           __builtin_bit_cast(float, __builtin_bit_cast(unsigned, o[2]) | (__builtin_bit_cast(unsigned, o[3]) << 16)), 0.f, 0.f); \
        else v = make_float4(o[0], o[1], o[2], o[3]);                                          \
      }                                                                                        \
      if constexpr (XB16) put_raw(&Bs[STAGE][0][s_chunk >> 1][s_row + i * RSTEP], v);         \
      else if constexpr (XPRE) put_pre(&Bs[STAGE][0][s_chunk >> 1][s_row + i * RSTEP], PB, v); \
      else put(&Bs[STAGE][0][s_chunk >> 1][s_row + i * RSTEP], PB, v, kb);                     \
    }                                                                                          \
  }
  constexpr bool LEANW = W3_LEAN && WM == 4 && NC == 1 && NP >= 2;
  auto mma = [&](auto curc) {
    constexpr int cur = decltype(curc)::value;
    if constexpr (LEANW) {                 // 128-VGPR form: the A fragments of one 32-row block at a time (same products, same order)
      uint4 bq[2][NP];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < NP; ++p) bq[j][p] = Bs[cur][p][lk][wn * 64 + j * 32 + li];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint4 ap[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) ap[p] = As[cur][p][lk][wm * 64 + i * 32 + li];
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_chain<NP>(ap, bq[j], acc[i][j]);
      }
      return;
    }
    uint4 af[2][NP], bf[2][NP];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        af[i][p] = As[cur][p][lk][wm * 64 + i * 32 + li];
        bf[i][p] = Bs[cur][p][lk][wn * 64 + i * 32 + li];
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = mfma_chain<NP>(af[i], bf[j], acc[i][j]);
    if constexpr (NC == 2) {
      uint4 bg[2][NP];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) bg[i][p] = Bs[cur][p][lk][BN + wn * 64 + i * 32 + li];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc2[i][j] = mfma_chain<NP>(af[i], bg[j], acc2[i][j]);
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  const int nsteps = g1 - g0;
  if constexpr (LEANW) {
    // one register set, fetched ONE step ahead: with two workgroups on the CU the other one's MFMAs cover the
    // wait a late load costs this one, and the second set's 15 registers are what kept the loop above 128.
    // Interior steps -- the shifted 16-t window inside the row, no ragged tail: all but the first dil / 16 steps of a
    // row -- take their own fetch / stage code behind a wave-uniform branch (the single set is waited for with
    // vmcnt(0) anyway, so a branch around its loads costs nothing): no validity selects, no re-alignment, the row
    // bases carried in two scalars.  With both kinds in one macro hipcc turned the edge path's arithmetic into
    // selects that every step paid: 130 VALU + 100 SALU per step beside 12 MFMAs (round 4: the MFMAs halved and this
    // became the loop).  Whole pairs in the loop, an odd last step behind it (single exit, see conv_gemm_x3_kernel).
    if (nsteps > 0) {
      unsigned base_a = GSZ * (unsigned)((long)b * a.gy_bstride), base_b = XSZ * (unsigned)((long)b * sg.x_bstride);
      const unsigned adv_a = GSZ * (unsigned)a.gy_bstride, adv_b = XSZ * (unsigned)sg.x_bstride;
      const int s_toff = sg.toff, s_tin = sg.Tin;
      auto adv = [&]() {
        tb += W2K;
        if (tb >= spb * W2K) { tb = 0; ++b; base_a += adv_a; base_b += adv_b; }
      };
      bool pfast = false;
#define W3L_FETCH()                                                                            \
      pfast = !ragged && tb + s_toff >= 0 && tb + W2K + s_toff <= s_tin;     /* wave-uniform */ \
      if (pfast) {                                                                             \
        const unsigned soa = base_a + GSZ * (unsigned)tb, sob = base_b + XSZ * (unsigned)(tb + s_toff); \
        _Pragma("unroll") for (int i = 0; i < NA; ++i)                                         \
          pra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa[i], soa, W3_LD_AUX)); \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                         \
          prb[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rbx, vobk[i], sob, W3_LD_AUX)); \
      } else W3_FETCH(pra, prb, pvm, pbs, pbt)
#define W3L_STAGE(STAGE, REAL)                                                                 \
      if (pfast) {                                                                             \
        const bool real_ = (REAL);                                                             \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                       \
          const float4 v = pra[i];                                                             \
          if constexpr (GPRE) {                                                                \
            put_pre(&As[STAGE][0][s_chunk >> 1][s_row + i * RSTEP], PA, v);                    \
            if (do_bias && real_) bsum[i] += (presplit_scaled(v.x) + presplit_scaled(v.y)) + (presplit_scaled(v.z) + presplit_scaled(v.w)); \
          } else {                                                                             \
          put(&As[STAGE][0][s_chunk >> 1][s_row + i * RSTEP], PA, v, ka);                      \
          if (do_bias && real_) bsum[i] += (v.x + v.y) + (v.z + v.w);                          \
          }                                                                                    \
        }                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                       \
          if constexpr (XPRE) put_pre(&Bs[STAGE][0][s_chunk >> 1][s_row + i * RSTEP], PB, prb[i]); \
          else put(&Bs[STAGE][0][s_chunk >> 1][s_row + i * RSTEP], PB, prb[i], kb);            \
        }                                                                                      \
      } else W3_STAGE(pra, prb, pvm, pbs, pbt, STAGE, REAL)
      W3L_FETCH();
      W3L_STAGE(0, true);
      __syncthreads();
#ifdef VQ_PHASE_TIMING
      unsigned long long sf = 0, sm = 0, ss = 0, sb = 0;
#endif
      for (int i = 0; i + 1 < nsteps; i += 2) {
        W3_T(q0);
        adv();
        W3L_FETCH();                                        // step i + 1
        W3_T(q1);
        mma(I0{});
        W3_T(q2);
        W3L_STAGE(1, true);
        W3_T(q3);
        __syncthreads();
        W3_T(q4);
        W3_ACC(sf, q0, q1); W3_ACC(sm, q1, q2); W3_ACC(ss, q2, q3); W3_ACC(sb, q3, q4);
        if (i + 2 < nsteps) adv();
        W3L_FETCH();                                        // step i + 2 (past the end: re-reads the last step, unused)
        mma(I1{});
        W3L_STAGE(0, i + 2 < nsteps);
        __syncthreads();
      }
#ifdef VQ_PHASE_TIMING
      if constexpr (XPRE && GPRE) if (tid == 0) {
        atomicAdd(&g_wphase[0][0], sf); atomicAdd(&g_wphase[0][1], sm); atomicAdd(&g_wphase[0][2], ss); atomicAdd(&g_wphase[0][3], sb);
        atomicAdd(&g_wphase[0][4], (unsigned long long)(nsteps / 2));
      }
#endif
      if (nsteps & 1) mma(I0{});
#undef W3L_FETCH
#undef W3L_STAGE
    }
  } else
  if (nsteps > 0) {
    W3_FETCH(pra, prb, pvm, pbs, pbt);
    if (nsteps > 1) advance();
    W3_FETCH(qra, qrb, qvm, qbs, qbt);
    W3_STAGE(pra, prb, pvm, pbs, pbt, 0, true);
    __syncthreads();
    // top of a pair (i even): LDS stage 0 holds step i, set Q holds (in flight) step i + 1
    for (int i = 0; i < nsteps; i += 2) {
      if (i + 2 < nsteps) advance();
      W3_FETCH(pra, prb, pvm, pbs, pbt);                  // step i + 2 (past the end: re-reads the last step, unused)
      mma(I0{});
      W3_STAGE(qra, qrb, qvm, qbs, qbt, 1, i + 1 < nsteps);
      __syncthreads();
      if (i + 1 >= nsteps) break;
      if (i + 3 < nsteps) advance();
      W3_FETCH(qra, qrb, qvm, qbs, qbt);                  // step i + 3
      mma(I1{});
      W3_STAGE(pra, prb, pvm, pbs, pbt, 0, i + 2 < nsteps);
      __syncthreads();
    }
  }
#undef W3_FETCH
#undef W3_STAGE

  // partial tile -> slab(s): a 256-row tile is two 128-row slab tiles, a 256-column tile two 128-column ones
  auto to_slab = [&](f32x16 (&ac)[2][2], int ntg_h) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int rowb = wm * 64 + mi * 32;                    // wave-uniform
      const int mt_slab = (m0 + rowb) / BM;
      if (mt_slab >= a.ntile_m) continue;
      float* slab = a.slabs + (((long)split * a.ntile_m + mt_slab) * a.ntile_n + ntg_h) * (BM * BN);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (rowb % BM) + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int col = wn * 64 + ni * 32 + li;
          slab[row * BN + col] = NP == 2 ? __builtin_ldexpf(ac[mi][ni][r], ku) : ac[mi][ni][r];
        }
    }
  };
  to_slab(acc, ntg);
  if constexpr (NC == 2) {
    if (n0 + BN < sg.cin) to_slab(acc2, ntg + 1);            // the segment may end in an odd 128-column tile
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 1, 4);
      v += __shfl_xor(v, 2, 4);
      if constexpr (GPRE) v = __builtin_ldexpf(v, -ka);       // the sums ran over hi + lo = gy * 2^ka
      const int row = m0 + s_row + RSTEP * i;              // global row
      if (s_chunk == 0 && row < a.ntile_m * BM)
        a.bslabs[(((long)split * a.nseg + s) * a.ntile_m + row / BM) * BM + row % BM] = v;
    }
  }
}

// wgrad3_dma_kernel -- wgrad3_kernel<4, 1, 2, true, true> (float32x2, BOTH operands pre-split: the dilated convs of ResidualNet,
// gh_l against x_l at two tap shifts) with the operands travelling global -> LDS by LDS-DMA.  Both operands are contiguous
// along the contraction axis (time) in HBM and a lane's MFMA fragment is 8 consecutive t of its row, so -- unlike the conv
// kernels' activations -- the stored dwords need no transposition: a stage is the RAW image [row][32 t] of hi | lo dwords,
// filled by `buffer_load_dwordx4 ... lds` (8 rows x one whole 128-byte line each per wave instruction; LDS[m0 + 16 lane]:
// the image is lane-linear, so the bank swizzle -- 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7): the 16 lanes one
// ds_read_b128 cycle serves land on 16 distinct slots of the 256-byte bank row -- is applied to the SOURCE offsets), and the
// pieces are separated AFTER the fragment read (two ds_read_b128 + eight v_perm_b32 per 32-row fragment pair).  No staging
// registers, no ds_write pass, no wait for the loads inside the step.
//   * ONE 16-wave workgroup per CU on a 256 x 256 tile (wave tile 64 x 64, 4 x 4 waves): a 32-t step brings 512 rows x 128 B
//     = 64 KB for 24 MFMAs per wave where two 256 x 128 workgroups brought 96 KB, in whole cache lines instead of halves;
//   * two stages (128 KB of LDS): [own DMAs of step i retired: vmcnt(0)] -> raw s_barrier -> first 16-t sub-step (12 MFMAs) ->
//     issue step i + 1 into the stage step i - 1 was read from -> second sub-step: one barrier per 24 MFMAs.  The DMAs go out
//     BEHIND the first sub-step's MFMAs: the CU's 64 DMA instructions of a step pass through the address path one after the
//     other (~50 cycles each: s_memtime stamps per wave showed the last waves of a workgroup issuing theirs 1 700 ticks after
//     the barrier, their MFMAs starting only then, and the first waves waiting that long at the next barrier);
//   * a step whose shifted window lies wholly in front of the row (tap 1, the first dil / 32 steps of a row) zero-fills its
//     B rows with a ds_write_b128; one that crosses the row's first or last sample (dil < 32 only) takes four predicated
//     dword loads per lane into the same slot (the compiler's own wait for them covers the step's DMAs as well);
//   * bias sums from the A fragments of the waves at wn == 0 (every row of the tile exactly once per sub-step).
// Same splits, slabs, scales (2^ka, 2^kb per tile; 2^ku back) and product order as wgrad3_kernel; the host requires
// Tout % 32 == 0, M % 256 == 0 and whole 256-column segments.
#ifndef W3_DMA
#define W3_DMA 1
#endif
// BF (matmul mode 1, both operands STORED as bf16: configs[4]'s chain): the same image with 2-byte elements -- a 128-byte line is
// 64 t, a step 64 t = four 16-t sub-steps of 4 MFMAs -- and a 16-byte chunk IS a lane's fragment (8 k of bf16): no permutes, no
// VALU in the loop at all; products and sums as wgrad3_kernel<4, 1, 1, true, true> (the stored values, fp32 accumulation).
constexpr int W3D_STAGE = 512 * 8;                         // uint4 per stage: [512 rows][8 chunks of 16 bytes]
template <bool BF>
__global__ __launch_bounds__(1024, 4) void wgrad3_dma_kernel(const WgradArgs a) {
  __shared__ uint4 ring[2 * W3D_STAGE];
  if (a.skip_flag != nullptr && *a.skip_flag != 0) return;
  constexpr unsigned ESZ = BF ? 2u : 4u;                   // bytes per stored element
  constexpr int TPC = 16 / (int)ESZ;                       // t per 16-byte chunk
  constexpr int BM2 = 256, W3DK = 8 * TPC, NSUB = W3DK / W2K, PPS = W3DK / WBK;   // positions per step; 16-t sub-steps; plan steps per step
  const int ntm = (a.ntile_m * BM + BM2 - 1) / BM2;
  int logical;                                             // XCD-aware order, as in wgrad3_kernel
  {
    const int nblk = gridDim.x, id = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int ntiles = ntm * a.ntile_p;
  const int tile = logical % ntiles;
  const int split = logical / ntiles;
  const int ct = tile % a.ntile_p;
  const int mt = tile / a.ntile_p;
  int s = 0;
#pragma unroll
  for (int i = 1; i < MAXSEG; ++i)
    if (i < a.nseg && ct >= a.seg[i].ptile0) s = i;
  const WSeg& sg = a.seg[s];
  [[maybe_unused]] int ka = 0, ku = 0;
  if constexpr (!BF) {
    const int eg = amax_expo(amax_load(sg.gy ? sg.amax_gy : a.amax_gy));
    const int ex = amax_expo(sg.amax_x ? amax_load(sg.amax_x) : __builtin_bit_cast(unsigned, sg.amax_x_static));
    ka = 14 - eg; ku = eg + ex - 28;
  }
  const int ntg = sg.tile0 + 2 * (ct - sg.ptile0);         // first 128-column slab tile of this workgroup
  const int n0 = (ntg - sg.tile0) * BN;
  const int m0 = mt * BM2;
  const int spb = a.steps_per_b / PPS;                     // (the host guarantees whole steps: the plan's counts are multiples of PPS)
  const int g0 = split * (a.steps_per_split / PPS);
  const int g1 = min(a.B * spb, g0 + a.steps_per_split / PPS);
  const int nsteps = g1 - g0;
  int b = g0 / spb;
  int tb = (g0 - b * spb) * W3DK;
  const int Tout = a.Tout;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, li = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float bsum = 0.f;

  // DMA roles: lane j fills slot j of a 1 KB run = row (j >> 3) of an 8-row group, physical chunk j & 7.  Wave w brings groups
  // w and w + 16 of A and of B (rows [8 w, 8 w + 8) and 128 further): two groups of one parity swizzle alike -- (row >> 1) & 7
  // = 4 (w & 1) + ((j >> 4) & 3) in both -- so one per-lane offset serves both runs, the second 128 rows up in the scalar offset.
  // (the host guarantees whole tiles: M % 256 == 0, cin % 256 == 0 -- every row of the tile exists)
  constexpr unsigned OOB = 0x80000000u;
  const i32x4_t ra4 = make_rsrc4(sg.gy ? sg.gy : a.gy), rb4 = make_rsrc4(sg.x);
  const rsrc_t rbx = make_rsrc(sg.x);
  const int chunk0 = (lane & 7) ^ (4 * (wave & 1) + ((lane >> 4) & 3));
  const unsigned voa0 = ESZ * (unsigned)((m0 + 8 * wave + (lane >> 3)) * Tout + TPC * chunk0);
  const unsigned vob0 = ESZ * (unsigned)((n0 + 8 * wave + (lane >> 3)) * sg.x_cstride + TPC * chunk0);
  const unsigned run_a = 128u * ESZ * (unsigned)Tout, run_b = 128u * ESZ * (unsigned)sg.x_cstride;     // 128 rows further
  const unsigned lds0 = lds_addr32(&ring[0]);
  // LDS image: [8-row group][stage][8 rows][8 chunks] -- a run of a stage is 1 KB, the other stage's run follows it, so a
  // fragment address reaches both stages, both row blocks and both chunks of a pair through the ds_read's immediate offset
  const unsigned dst_a = lds0 + 2048u * (unsigned)wave, dst_b = lds0 + 65536u + 2048u * (unsigned)wave;
  const int wslot_b = 4096 + 128 * wave + lane;            // run 0's slot of this lane as a uint4 index, stage 0 (edge steps; run 1: + 2048, stage 1: + 64)
  unsigned base_a = ESZ * (unsigned)((long)b * a.gy_bstride), base_b = ESZ * (unsigned)((long)b * sg.x_bstride);
  const unsigned adv_a = ESZ * (unsigned)a.gy_bstride, adv_b = ESZ * (unsigned)sg.x_bstride;
  const int s_toff = sg.toff, s_tin = sg.Tin;
  // bias sums (workgroup-uniform): the four waves that share a 64-row block take one (row block i, sub-step h) each -- wave
  // wn: i = wn >> 1, h = wn & 1 -- from the A fragments they read anyway, and the pairs meet through LDS behind the loop
  const bool do_bias = (ntg == sg.tile0) && (a.bslabs != nullptr) && (sg.gb || sg.gb2 || (s == 0 && a.ngbl > 0));
  const int b_i = wn >> 1, b_h = wn & 1;

  auto issue = [&](const int st) {                         // the step at the cursor -> stage st; the cursor moves on
    const unsigned sbytes = (unsigned)st * 1024u;
    const unsigned soa = base_a + ESZ * (unsigned)tb;
    lds_dma16(dst_a + sbytes, voa0, ra4, soa);
    lds_dma16(dst_a + sbytes + 32768u, voa0, ra4, soa + run_a);
    const int w0 = tb + s_toff;                            // the shifted window [w0, w0 + 32)
    if (w0 >= 0 && w0 + W3DK <= s_tin) {                   // wave-uniform
      const unsigned sob = base_b + ESZ * (unsigned)w0;
      lds_dma16(dst_b + sbytes, vob0, rb4, sob);
      lds_dma16(dst_b + sbytes + 32768u, vob0, rb4, sob + run_b);
    } else if (w0 + W3DK <= 0 || w0 >= s_tin) {            // wholly outside the row: zeros
      ring[st * 64 + wslot_b] = make_uint4(0u, 0u, 0u, 0u);
      ring[st * 64 + wslot_b + 2048] = make_uint4(0u, 0u, 0u, 0u);
    } else {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int tin = w0 + TPC * chunk0;
        const unsigned vrow = ESZ * (unsigned)((n0 + 128 * q + 8 * wave + (lane >> 3)) * sg.x_cstride);
        unsigned v[4];
        if constexpr (BF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int t = tin + 2 * e;
            const unsigned lo = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rbx, (t >= 0 && t < s_tin) ? vrow + 2u * (unsigned)t : OOB, base_b, 0);
            const unsigned hi = (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rbx, (t + 1 >= 0 && t + 1 < s_tin) ? vrow + 2u * (unsigned)(t + 1) : OOB, base_b, 0);
            v[e] = lo | (hi << 16);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int t = tin + e;
            v[e] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rbx, (t >= 0 && t < s_tin) ? vrow + 4u * (unsigned)t : OOB, base_b, 0);
          }
        }
        ring[st * 64 + wslot_b + 2048 * q] = make_uint4(v[0], v[1], v[2], v[3]);
      }
    }
    tb += W3DK;
    if (tb >= spb * W3DK) { tb = 0; ++b; base_a += adv_a; base_b += adv_b; }
  };
  // fragment slots (uint4 index): row R = 32-row base + li sits in group R >> 3 at row R & 7, logical chunk c at physical
  // c ^ ((R >> 1) & 7).  fp16 pairs: sub-step h wants chunks 4 h + 2 lk | + 1 (four addresses per operand: the XORs of the
  // low three bits); bf16: sub-step h wants chunk 2 h + lk, the whole fragment.  Everything else -- stage + 64, row block
  // + 512 -- is an immediate; the sub-steps' XORs (64 / 32 bytes per h) are applied where they are used (registers).
  const int fsw = (BF ? lk : 2 * lk) ^ ((li >> 1) & 7);
  const int fa = (wm * 8 + (li >> 3)) * 128 + (li & 7) * 8 + fsw, fb = 4096 + (wn * 8 + (li >> 3)) * 128 + (li & 7) * 8 + fsw;
  const unsigned ba0 = 16u * (unsigned)fa, ba1 = 16u * (unsigned)(fa ^ 1), bb0 = 16u * (unsigned)fb, bb1 = 16u * (unsigned)(fb ^ 1);   // as byte offsets
  auto frag = [&](const uint4 u0, const uint4 u1, uint4 (&p)[2]) {     // eight stored elements -> the hi and the lo fragment word
    presplit_stage(__builtin_bit_cast(float, u0.x), __builtin_bit_cast(float, u0.y), p[0].x, p[1].x);
    presplit_stage(__builtin_bit_cast(float, u0.z), __builtin_bit_cast(float, u0.w), p[0].y, p[1].y);
    presplit_stage(__builtin_bit_cast(float, u1.x), __builtin_bit_cast(float, u1.y), p[0].z, p[1].z);
    presplit_stage(__builtin_bit_cast(float, u1.z), __builtin_bit_cast(float, u1.w), p[0].w, p[1].w);
  };
  auto esum = [&](const uint4 u) {
    if constexpr (BF) {                                    // eight stored bf16
      auto two = [](const unsigned w) { return __builtin_bit_cast(float, w << 16) + __builtin_bit_cast(float, w & 0xffff0000u); };
      return (two(u.x) + two(u.y)) + (two(u.z) + two(u.w));
    } else
    return (presplit_scaled(__builtin_bit_cast(float, u.x)) + presplit_scaled(__builtin_bit_cast(float, u.y))) +
           (presplit_scaled(__builtin_bit_cast(float, u.z)) + presplit_scaled(__builtin_bit_cast(float, u.w)));
  };
  auto ld16 = [&](const unsigned off) { return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(ring) + off); };
  auto mma = [&](const int st, const int h) {
    const unsigned so = 1024u * (unsigned)st;
    if constexpr (BF) {
      unsigned xa = ba0, xb = bb0;
      if (h == 1) asm volatile("v_xor_b32 %0, 32, %0\n\tv_xor_b32 %1, 32, %1" : "+v"(xa), "+v"(xb));
      if (h == 2) asm volatile("v_xor_b32 %0, 64, %0\n\tv_xor_b32 %1, 64, %1" : "+v"(xa), "+v"(xb));
      if (h == 3) asm volatile("v_xor_b32 %0, 0x60, %0\n\tv_xor_b32 %1, 0x60, %1" : "+v"(xa), "+v"(xb));
      uint4 bq[2], aq[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bq[j] = ld16(xb + so + j * 8192u);
#pragma unroll
      for (int i = 0; i < 2; ++i) aq[i] = ld16(xa + so + i * 8192u);
      if (do_bias && (h & 1) == b_h) bsum += b_i ? esum(aq[1]) : esum(aq[0]);      // (no dynamic index: hipcc would move aq to scratch)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma16<1>(aq[i], bq[j], acc[i][j]);
    } else {
      unsigned xa0 = ba0, xa1 = ba1, xb0 = bb0, xb1 = bb1;
      if (h) {                                             // the second sub-step's four addresses are made here, not kept (registers)
        asm volatile("v_xor_b32 %0, 64, %0\n\tv_xor_b32 %1, 64, %1\n\tv_xor_b32 %2, 64, %2\n\tv_xor_b32 %3, 64, %3" : "+v"(xa0), "+v"(xa1), "+v"(xb0), "+v"(xb1));
      }
      uint4 bq[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j) frag(ld16(xb0 + so + j * 8192u), ld16(xb1 + so + j * 8192u), bq[j]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4 u0 = ld16(xa0 + so + i * 8192u), u1 = ld16(xa1 + so + i * 8192u);
        uint4 ap[2];
        frag(u0, u1, ap);
        if (do_bias && i == b_i && h == b_h) bsum += esum(u0) + esum(u1);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_chain<2>(ap, bq[j], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);                   // the next sub-step's eight reads stay behind this one's MFMAs (registers)
    }
  };
#ifdef VQ_PHASE_TIMING
  unsigned long long sw = 0, sb = 0, si = 0, sm = 0;
#endif
#define W3D_STEP(ST, I)                                                                        \
  {                                                                                            \
    W3_T(q0);                                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                               \
    W3_T(q1);                                                                                  \
    asm volatile("s_barrier" ::: "memory");                                                    \
    W3_T(q2);                                                                                  \
    mma(ST, 0);                                                                                \
    W3_T(q3);                                                                                  \
    if ((I) + 1 < nsteps) issue((ST) ^ 1);                                                     \
    W3_T(q3b);                                                                                 \
    mma(ST, 1);                                                                                \
    if constexpr (NSUB == 4) { mma(ST, 2); mma(ST, 3); }                                       \
    W3_T(q4);                                                                                  \
    W3_ACC(sm, q3b, q4);                                                                       \
    W3_ACC(sw, q0, q1); W3_ACC(sb, q1, q2); W3_ACC(sm, q2, q3); W3_ACC(si, q3, q3b);           \
  }
  if (nsteps > 0) {
    issue(0);
    int i = 0;
    for (; i + 2 <= nsteps; i += 2) {
      W3D_STEP(0, i);
      W3D_STEP(1, i + 1);
    }
    if (i < nsteps) W3D_STEP(0, i);
  }
#undef W3D_STEP
#ifdef VQ_PHASE_TIMING
  if (tid == 0) {
    atomicAdd(&g_wphase[1][0], sw); atomicAdd(&g_wphase[1][1], sb); atomicAdd(&g_wphase[1][2], si); atomicAdd(&g_wphase[1][3], sm);
    atomicAdd(&g_wphase[1][4], (unsigned long long)nsteps);
  }
  if (lane == 0) { atomicAdd(&g_wwave[wave][0], sw); atomicAdd(&g_wwave[wave][1], sb); atomicAdd(&g_wwave[wave][2], si); atomicAdd(&g_wwave[wave][3], sm); }
#endif

#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int rowb = wm * 64 + mi * 32;                    // wave-uniform
    const int mt_slab = (m0 + rowb) / BM;
    if (mt_slab >= a.ntile_m) continue;
    float* slab = a.slabs + (((long)split * a.ntile_m + mt_slab) * a.ntile_n + ntg + (wn >> 1)) * (BM * BN);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (rowb % BM) + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int col = (wn & 1) * 64 + ni * 32 + li;
        slab[row * BN + col] = BF ? acc[mi][ni][r] : __builtin_ldexpf(acc[mi][ni][r], ku);
      }
  }
  if (do_bias) {                                           // workgroup-uniform
    float* red = reinterpret_cast<float*>(ring);           // [wave][32 rows]: the ring is free behind a barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const float v = bsum + __shfl_xor(bsum, 32);           // the two k halves of the fragment
    if (lk == 0) red[wave * 32 + li] = v;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (b_h == 0 && lk == 0) {
      const float t = __builtin_ldexpf(red[wave * 32 + li] + red[(wave + 1) * 32 + li], -ka);    // the sums ran over hi + lo = gy * 2^ka (bf16: ka = 0)
      const int row = m0 + wm * 64 + b_i * 32 + li;
      if (row < a.ntile_m * BM)
        a.bslabs[(((long)split * a.nseg + s) * a.ntile_m + row / BM) * BM + row % BM] = t;
    }
  }
}

// block = (64 outputs) x (4 split groups): each thread sums every 4th split with
// 4 independent accumulators, then the 4 groups combine through LDS in fixed order.
// Outputs [0,total) are weight-gradient entries, [total, total + nseg*Mpad) bias entries.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradArgs a, int nsplit) {
  __shared__ float red[4][64];
  if (a.skip_flag != nullptr && *a.skip_flag != 0) return;
  const long ncol = (long)a.ntile_n * BN;
  const long total = (long)a.ntile_m * BM * ncol;
  const long mpad = (long)a.ntile_m * BM;
  const long total_ext = total + (a.bslabs ? (long)a.nseg * mpad : 0);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (long base = (long)blockIdx.x * 64; base < total_ext; base += (long)gridDim.x * 64) {
    const long i = base + tx;
    bool ok = i < total_ext;
    const bool is_bias = i >= total;
    int row = 0, s = 0, ci = 0;
    const float* p = a.slabs;
    long sstride = (long)a.ntile_m * a.ntile_n * (BM * BN);
    if (ok && !is_bias) {
      const int colg = (int)(i % ncol);
      row = (int)(i / ncol);
      const int ntg = colg / BN, col = colg % BN;
#pragma unroll
      for (int k = 1; k < MAXSEG; ++k)
        if (k < a.nseg && ntg >= a.seg[k].tile0) s = k;
      ci = (ntg - a.seg[s].tile0) * BN + col;
      ok = row < a.M && ci < a.seg[s].cin && a.seg[s].gw != nullptr;
      const int mt = row / BM, r = row % BM;
      p = a.slabs + ((long)mt * a.ntile_n + ntg) * (BM * BN) + r * BN + col;
    } else if (ok) {
      const long bi = i - total;
      s = (int)(bi / mpad);
      row = (int)(bi % mpad);
      ok = row < a.M && (a.seg[s].gb || a.seg[s].gb2 || (s == 0 && a.ngbl > 0));
      p = a.bslabs + (long)s * mpad + row;
      sstride = (long)a.nseg * mpad;
    }
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (ok) {
      int sp = ty;
      for (; sp + 12 < nsplit; sp += 16) {
        v0 += p[(long)sp * sstride];
        v1 += p[(long)(sp + 4) * sstride];
        v2 += p[(long)(sp + 8) * sstride];
        v3 += p[(long)(sp + 12) * sstride];
      }
      for (; sp < nsplit; sp += 4) v0 += p[(long)sp * sstride];
    }
    __syncthreads();
    red[ty][tx] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (ty == 0 && ok) {
      const float v = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
      const WSeg& sg = a.seg[s];
      if (!is_bias) {
        float* dst = sg.gw + (long)row * sg.gw_co_stride + (long)ci * sg.gw_ci_stride;
        *dst = a.accumulate ? *dst + v : v;
      } else {
        if (sg.gb) sg.gb[row] = a.accumulate ? sg.gb[row] + v : v;
        if (sg.gb2) sg.gb2[row] = a.accumulate ? sg.gb2[row] + v : v;
        if (s == 0)
          for (int l = 0; l < a.ngbl; ++l)
            if (a.gbl[l]) a.gbl[l][row] = a.accumulate ? a.gbl[l][row] + v : v;
      }
    }
  }
}


// wgrad_kernel runs 2 workgroups per CU (205 VGPR): 512 resident slots on 256 CUs.  The K axis is
// the flattened (batch, time) axis cut into WBK-wide steps; choose the number of K splits so that
// tiles x splits is just under a whole number of residency rounds, with as few splits as that
// allows (every split costs one 64 KB partial slab per tile, written and re-read by the reduce).
#ifndef W3_SLOTS_256
#define W3_SLOTS_256 512
#endif
bool wgrad_dma_shape(int M, int Tout, const int* cins, int nseg) {
  bool ok = W3_DMA != 0 && (g_matmul_dtype == 3 || g_matmul_dtype == 1) && M % 256 == 0 && Tout % WBK == 0;
  for (int i = 0; i < nseg; ++i) ok = ok && cins[i] % (2 * BN) == 0;
  return ok;
}
WgradPlan plan_wgrad(int M, int B, int Tout, const int* cins, int nseg, bool wide) {
  WgradPlan p;
  p.nseg = nseg;
  p.wide = wide ? 1 : 0;
  p.ntile_m = cdiv(M, BM);
  p.ntile_n = 0;
  for (int i = 0; i < nseg; ++i) p.ntile_n += cdiv(cins[i], BN);
  const long tiles = (long)p.ntile_m * p.ntile_n;
  p.steps_per_b = cdiv(Tout, WBK);
  const long total_steps = (long)B * p.steps_per_b;
  long maxs = total_steps / 4;                 // at least 4 K steps (128 positions) per split
  if (maxs < 1) maxs = 1;
  if (maxs > 256) maxs = 256;
  // 128-row tile units per residency round: 512 = one 256-row workgroup (two units) per CU.  The 256-row
  // six-product kernel would admit two per CU since round 3 (128 VGPRs), i.e. 1024 units: twice the splits
  // (and slab traffic) for half the K range each measured 22.02 against 21.97 ms per step, 768 units 22.24:
  // the plan stays.
  // wide (wgrad3_dma_kernel): a workgroup covers four units and owns its CU: 1024 units per round, and the first split count
  // that fills 92 % of ONE round is taken (a second round would pay the 256 KB slab store of every workgroup again)
  const long slots = wide ? 1024 : (M % 256 == 0 && g_matmul_dtype != 0) ? W3_SLOTS_256 : 512;
  long want = 1;
  double best = -1.0;
  const bool even = wide && g_matmul_dtype == 1;          // wgrad3_dma_kernel<BF> takes two plan steps (64 positions) at a time
  for (long w = 1; w <= maxs; ++w) {
    long sps = (total_steps + w - 1) / w;
    if (even) sps += sps & 1;
    const long ns = (total_steps + sps - 1) / sps;        // splits actually produced
    const long blocks = tiles * ns;
    const long rounds = (blocks + slots - 1) / slots;
    const double eff = (double)blocks / (double)(rounds * slots);
    if (eff > best + 1e-9) { best = eff; want = w; }
    if (eff >= 0.92 && (blocks >= slots || wide)) { want = w; break; }
  }
  p.steps_per_split = (int)((total_steps + want - 1) / want);
  if (even) p.steps_per_split += p.steps_per_split & 1;
  p.nsplit = (int)((total_steps + p.steps_per_split - 1) / p.steps_per_split);
  p.slab_floats = (size_t)p.nsplit * p.ntile_m * p.ntile_n * BM * BN;
  p.bslab_floats = (size_t)p.nsplit * nseg * p.ntile_m * BM;
  return p;
}

int launch_wgrad(WgradArgs& w, const WgradPlan& p, float* ws, int tag, hipStream_t st) {
  w.ntile_m = p.ntile_m; w.ntile_n = p.ntile_n;
  w.steps_per_b = p.steps_per_b; w.steps_per_split = p.steps_per_split; w.nsplit = p.nsplit;
  w.slabs = ws;
  bool any_b = w.ngbl > 0;
  for (int i = 0; i < w.nseg; ++i) any_b = any_b || w.seg[i].gb || w.seg[i].gb2;
  w.bslabs = any_b ? ws + p.slab_floats : nullptr;
  int t0 = 0;
  int p0 = 0;
  for (int i = 0; i < w.nseg; ++i) {
    w.seg[i].tile0 = t0; t0 += cdiv(w.seg[i].cin, BN);
    w.seg[i].ptile0 = p0; p0 += cdiv(w.seg[i].cin, 2 * BN);
  }
  w.ntile_p = p0;
  // 16-B row loads: every row start and every chunk start must be 16-B aligned and no float4
  // may straddle a row end
  bool av = (w.Tout % 4 == 0) && (w.gy_bstride % 4 == 0);
  for (int i = 0; i < w.nseg; ++i) {
    const float* g = w.seg[i].gy ? w.seg[i].gy : w.gy;
    av = av && (((uintptr_t)g) % 16 == 0);
  }
  w.avec = av ? 1 : 0;
  for (int i = 0; i < w.nseg; ++i) {
    WSeg& sg = w.seg[i];
    // dwordx4 row loads: global loads only need dword alignment on gfx950, so a shifted window
    // (toff % 4 != 0: dilations 1 and 2) keeps them; a group that straddles the row's valid range is
    // fetched element by element inside the kernel
    sg.vec = (sg.tmul == 1 && sg.tdiv == 1 && w.Tout % 4 == 0) ? 1 : 0;
  }
  // fp32, stride-1 segments, 16-B aligned output-gradient rows: the 16-byte-LDS kernel
  bool fast = av && g_wgrad_impl != 1;
  for (int i = 0; i < w.nseg; ++i) fast = fast && w.seg[i].tmul == 1 && w.seg[i].tdiv == 1;
  // wgrad3_kernel addresses both operands with 32-bit buffer offsets from the tensor base
  fast = fast && (g_matmul_dtype == 0 || (long)w.B * w.gy_bstride * 4 < (1L << 31));
  for (int i = 0; i < w.nseg; ++i) fast = fast && (g_matmul_dtype == 0 || (long)w.B * w.seg[i].x_bstride * 4 < (1L << 31));
  // the arithmetic of this launch (see launch_gemm): float32x2 when the caller gave every segment its maxima and the
  // shape runs on wgrad3_kernel; both operands are activations, so falling back to mode 2 needs nothing re-packed
  VQ_REQUIRE(!w.f16x2 || g_matmul_dtype == 3, "wgrad: float32x2 launch outside matmul mode 3");
  if (w.f16x2)
    for (int i = 0; i < w.nseg; ++i)
      VQ_REQUIRE((w.seg[i].amax_x || w.seg[i].amax_x_static > 0.f) && (w.seg[i].gy ? w.seg[i].amax_gy != nullptr : w.amax_gy != nullptr),
                 "wgrad: float32x2 segment %d without its maxima", i);
  const int mode = g_matmul_dtype == 3 ? ((w.f16x2 && fast) ? 3 : 2) : g_matmul_dtype;
  ProfScope ps(tag, st);
  if (g_matmul_dtype == 3 && (w.x16 || w.g16)) {        // pre-split operands (see presplit_pair): their `amax` words are scale words
    VQ_REQUIRE(fast && mode == 3 && w.M % 256 == 0, "wgrad: pre-split operands need a float32x2 launch on 256-row tiles");
    const dim3 grid((p.ntile_m / 2) * p.ntile_n * p.nsplit);
    int cins_[MAXSEG];
    for (int i = 0; i < w.nseg; ++i) cins_[i] = w.seg[i].cin;
    const bool dma = w.x16 && w.g16 && wgrad_dma_shape(w.M, w.Tout, cins_, w.nseg);     // (with any plan; the caller asks for a wide one when it knows)
    if (dma) hipLaunchKernelGGL(wgrad3_dma_kernel<false>, dim3((p.ntile_m / 2) * w.ntile_p * p.nsplit), dim3(1024), 0, st, w);
    else if (w.x16 && w.g16) hipLaunchKernelGGL((wgrad3_kernel<4, 1, 2, true, true>), grid, dim3(512), 0, st, w);
    else if (w.g16) hipLaunchKernelGGL((wgrad3_kernel<4, 1, 2, false, true>), grid, dim3(512), 0, st, w);
    else hipLaunchKernelGGL((wgrad3_kernel<4, 1, 2, true, false>), grid, dim3(512), 0, st, w);
  } else if (fast && mode == 3 && w.M % 256 == 0) {
    hipLaunchKernelGGL((wgrad3_kernel<4, 1, 2>), dim3((p.ntile_m / 2) * p.ntile_n * p.nsplit), dim3(512), 0, st, w);
  } else if (fast && mode == 3) {
    hipLaunchKernelGGL((wgrad3_kernel<2, 1, 2>), dim3(p.ntile_m * p.ntile_n * p.nsplit), dim3(256), 0, st, w);
  } else if (fast && mode == 2 && w.M % 256 == 0) {
    hipLaunchKernelGGL((wgrad3_kernel<4, 1, 3>), dim3((p.ntile_m / 2) * p.ntile_n * p.nsplit), dim3(512), 0, st, w);
  } else if (fast && mode == 2) {
    hipLaunchKernelGGL((wgrad3_kernel<2, 1, 3>), dim3(p.ntile_m * p.ntile_n * p.nsplit), dim3(256), 0, st, w);
  } else if (w.x16 || w.g16) {
    bool ok16 = fast && mode == 1 && w.M % 256 == 0 && w.Tout % W2K == 0;
    if (w.x16) for (int i = 0; i < w.nseg; ++i) ok16 = ok16 && w.seg[i].Tin == w.Tout;
    VQ_REQUIRE(ok16, "wgrad: bf16-stored operands need matmul mode 1, stride-1 segments, 256-row tiles and T %% 16 == 0");
    const dim3 grid((p.ntile_m / 2) * p.ntile_n * p.nsplit);
    int cins_[MAXSEG];
    for (int i = 0; i < w.nseg; ++i) cins_[i] = w.seg[i].cin;
    if (w.x16 && w.g16 && wgrad_dma_shape(w.M, w.Tout, cins_, w.nseg) && w.Tout % 64 == 0 && p.steps_per_b % 2 == 0 && p.steps_per_split % 2 == 0)
      hipLaunchKernelGGL(wgrad3_dma_kernel<true>, dim3((p.ntile_m / 2) * w.ntile_p * p.nsplit), dim3(1024), 0, st, w);
    else if (w.x16 && w.g16) hipLaunchKernelGGL((wgrad3_kernel<4, 1, 1, true, true>), grid, dim3(512), 0, st, w);
    else if (w.g16) hipLaunchKernelGGL((wgrad3_kernel<4, 1, 1, false, true>), grid, dim3(512), 0, st, w);
    else hipLaunchKernelGGL((wgrad3_kernel<4, 1, 1, true>), grid, dim3(512), 0, st, w);
  } else if (fast && mode == 1 && w.M % 256 == 0) {
    hipLaunchKernelGGL((wgrad3_kernel<4, 1, 1>), dim3((p.ntile_m / 2) * p.ntile_n * p.nsplit), dim3(512), 0, st, w);
  } else if (fast && mode == 1) {
    hipLaunchKernelGGL((wgrad3_kernel<2, 1, 1>), dim3(p.ntile_m * p.ntile_n * p.nsplit), dim3(256), 0, st, w);
  } else if (fast && w.M % 256 == 0) {
    hipLaunchKernelGGL(wgrad2_kernel<4>, dim3((p.ntile_m / 2) * p.ntile_n * p.nsplit), dim3(512), 0, st, w);
  } else if (fast) {
    hipLaunchKernelGGL(wgrad2_kernel<2>, dim3(p.ntile_m * p.ntile_n * p.nsplit), dim3(256), 0, st, w);
  } else if (mode == 1) hipLaunchKernelGGL(wgrad_kernel<true>, dim3(p.ntile_m * p.ntile_n, p.nsplit), dim3(NT), 0, st, w);
  else hipLaunchKernelGGL(wgrad_kernel<false>, dim3(p.ntile_m * p.ntile_n, p.nsplit), dim3(NT), 0, st, w);
  VQ_LAUNCH_CHECK();
  const long total = (long)p.ntile_m * BM * p.ntile_n * BN + (long)p.nseg * p.ntile_m * BM;
  int nb = (int)((total + 63) / 64);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb), dim3(256), 0, st, w, p.nsplit);
  VQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace vq

#ifdef VQ_PHASE_TIMING
extern "C" int vqvae_debug_wphases(unsigned long long* out, int reset) {       // dev aid, see g_wphase
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(vq::g_wphase), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(vq::g_wwave), sizeof(unsigned long long) * 64) != hipSuccess) return 1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(vq::g_wphase), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif
