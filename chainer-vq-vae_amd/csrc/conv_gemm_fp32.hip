// conv_gemm_fp32.hip -- matmul mode 0: the conv GEMMs on v_mfma_f32_32x32x2_f32 (fp32 operands, no split).
#include "gemm_common.h"

namespace vq {

template <int EPI, int WM, bool BF16>
// (the linear 128-row instantiation carries the split-K path and needs 171 registers: three waves per
// SIMD; at four it spilled 118 of them)
__global__ __launch_bounds__(128 * WM, (WM == 2 ? (EPI == EPI_LINEAR ? 3 : 4) : 2)) void conv_gemm_kernel(const GemmArgs a) {
  static_assert(!BF16, "matmul mode 1 runs on conv_gemm_x3_kernel<..., NP = 1>");
  constexpr int BM = 64 * WM, NT = 128 * WM;
  __shared__ float As[2][BK][BM];
  __shared__ float Bs[2][BK][BN];
  if (a.skip_flag != nullptr && *a.skip_flag != 0) return;

  // ---- XCD-aware tile order: consecutive logical tiles (which share the same
  // activation window across their M tiles) land on the same XCD / L2. -------
  const int nblk = gridDim.x;
  int logical;
  {
    const int id = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int ntiles_all = a.ntile_m * a.ntile_n * a.B;
  const int ksp = (EPI == EPI_LINEAR && WM == 2 && !BF16 && a.ksplit > 1) ? logical / ntiles_all : 0;
  const int tile_id = (EPI == EPI_LINEAR && WM == 2 && !BF16 && a.ksplit > 1) ? logical % ntiles_all : logical;
  const int mt = tile_id % a.ntile_m;
  const int rest = tile_id / a.ntile_m;
  const int nt = rest % a.ntile_n;
  const int b = rest / a.ntile_n;
  const int m0 = mt * BM, t0 = nt * BN;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int nk = 0;
  for (int s = 0; s < a.nseg; ++s) nk += (a.seg[s].cin + BK - 1) / BK;

  if constexpr (WM == 4) {
    float4 ra0, ra1;
    float4 rb0, rb1;
    bool rvec = false;

    // thread roles for the staging loads (per K step: A = 16 x BM, B = 16 x 128 floats)
    constexpr int ACOLS4 = BM / 4;                        // float4 per A row
    const int a_k = tid / ACOLS4, a_col = (tid % ACOLS4) * 4;    // A: rows a_k, a_k + 8
    const int v_k = tid >> 5, v_col = (tid & 31) * 4;     // vector B: row v_k (+8 when WM == 2)
    constexpr int BROWS = NT / 128;                        // scalar B: rows b_k + BROWS*i
    const int b_n = tid & 127, b_k = tid >> 7;

    // Staging state of the NEXT K step, advanced incrementally: a VALU instruction issued beside
    // the MFMA stream costs ~4 % of an MFMA slot (tools/ubench/mfma_coexec.hip), so the per-step
    // address arithmetic is two pointer bumps; everything else is set up once per segment.
    int s_n = 0, c_n = 0, cin_n = 0;
    const float* wp = nullptr;       // A: row a_k of the chunk (+ wrow8 for row a_k + 8)
    const float* xp = nullptr;       // B: vector path row v_k at the window start, scalar path row b_k at tin
    long wrow8 = 0, wadv = 0, xadv = 0, xrow = 0;
    bool rvec_n = false, ok_n = false;
    auto seg_setup = [&](int s) {
      const Seg& sg = a.seg[s];
      cin_n = sg.cin; c_n = 0;
      wp = sg.w + (long)a_k * sg.ldw + m0 + a_col;
      wrow8 = 8L * sg.ldw; wadv = (long)BK * sg.ldw; xadv = (long)BK * sg.x_cstride;
      const float* xb = sg.x + (long)b * sg.x_bstride;
      const int tw = t0 * sg.tmul + sg.toff;   // window start (when tmul==1,tdiv==1)
      rvec_n = sg.vec && ((tw & 3) == 0) && tw >= 0 && (tw + BN) <= sg.Tin;
      if (rvec_n) {
        xp = xb + (long)v_k * sg.x_cstride + tw + v_col;
        xrow = 8L * sg.x_cstride;
      } else {
        const int tnum = (t0 + b_n) * sg.tmul + sg.toff;
        bool ok = tnum >= 0;
        int tin = tnum;
        if (sg.tdiv > 1) { ok = ok && (tnum % sg.tdiv == 0); tin = tnum / sg.tdiv; }
        ok_n = ok && tin < sg.Tin;
        xp = xb + (long)b_k * sg.x_cstride + (ok_n ? tin : 0);
        xrow = (long)BROWS * sg.x_cstride;
      }
    };
    auto load_next = [&]() {          // chunk (s_n, c_n) -> ra*, rb*; then step to the following chunk
      ra0 = *reinterpret_cast<const float4*>(wp);                 // packed slabs are zero padded to 16 rows
      ra1 = *reinterpret_cast<const float4*>(wp + wrow8);
      rvec = rvec_n;
      const bool full = c_n + BK <= cin_n;
      if (rvec_n) {
        if (full) {
          rb0 = *reinterpret_cast<const float4*>(xp);
          if (WM == 2) rb1 = *reinterpret_cast<const float4*>(xp + xrow);
        } else {
          rb0 = make_float4(0.f, 0.f, 0.f, 0.f);
          rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c_n + v_k < cin_n) rb0 = *reinterpret_cast<const float4*>(xp);
          if (WM == 2 && c_n + v_k + 8 < cin_n) rb1 = *reinterpret_cast<const float4*>(xp + xrow);
        }
      } else {
        const int cb = c_n + b_k;
        rb0.x = (ok_n && (full || cb + 0 * BROWS < cin_n)) ? xp[0 * xrow] : 0.f;
        rb0.y = (ok_n && (full || cb + 1 * BROWS < cin_n)) ? xp[1 * xrow] : 0.f;
        rb0.z = (ok_n && (full || cb + 2 * BROWS < cin_n)) ? xp[2 * xrow] : 0.f;
        rb0.w = (ok_n && (full || cb + 3 * BROWS < cin_n)) ? xp[3 * xrow] : 0.f;
        if (WM == 2) {
          rb1.x = (ok_n && (full || cb + 4 * BROWS < cin_n)) ? xp[4 * xrow] : 0.f;
          rb1.y = (ok_n && (full || cb + 5 * BROWS < cin_n)) ? xp[5 * xrow] : 0.f;
          rb1.z = (ok_n && (full || cb + 6 * BROWS < cin_n)) ? xp[6 * xrow] : 0.f;
          rb1.w = (ok_n && (full || cb + 7 * BROWS < cin_n)) ? xp[7 * xrow] : 0.f;
        }
      }
      c_n += BK;
      if (c_n >= cin_n) {
        if (++s_n < a.nseg) seg_setup(s_n);
      } else {
        wp += wadv; xp += xadv;
      }
    };
    auto store_tiles = [&](auto bufc) {
      constexpr int buf = decltype(bufc)::value;
      *reinterpret_cast<float4*>(&As[buf][a_k][a_col]) = ra0;
      *reinterpret_cast<float4*>(&As[buf][a_k + 8][a_col]) = ra1;
      if (rvec) {
        *reinterpret_cast<float4*>(&Bs[buf][v_k][v_col]) = rb0;
        if (WM == 2) *reinterpret_cast<float4*>(&Bs[buf][v_k + 8][v_col]) = rb1;
      } else {
        Bs[buf][b_k + 0 * BROWS][b_n] = rb0.x;  Bs[buf][b_k + 1 * BROWS][b_n] = rb0.y;
        Bs[buf][b_k + 2 * BROWS][b_n] = rb0.z;  Bs[buf][b_k + 3 * BROWS][b_n] = rb0.w;
        if (WM == 2) {
          Bs[buf][b_k + 4 * BROWS][b_n] = rb1.x;  Bs[buf][b_k + 5 * BROWS][b_n] = rb1.y;
          Bs[buf][b_k + 6 * BROWS][b_n] = rb1.z;  Bs[buf][b_k + 7 * BROWS][b_n] = rb1.w;
        }
      }
    };
    // one K step on LDS buffer `cur` (compile-time: every LDS address is base + immediate)
    auto k_step = [&](auto curc, bool more) {
      constexpr int cur = decltype(curc)::value;
      if (more) load_next();
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
      if (more) store_tiles(std::integral_constant<int, cur ^ 1>{});
      __syncthreads();
    };

    seg_setup(0);
    load_next();
    store_tiles(std::integral_constant<int, 0>{});
    __syncthreads();

    for (int it = 0; it < nk; it += 2) {
      k_step(std::integral_constant<int, 0>{}, it + 1 < nk);
      if (it + 1 < nk) k_step(std::integral_constant<int, 1>{}, it + 2 < nk);
    }

  } else {
    // 128-row tiles (4 workgroups per CU): the plain per-step staging measured faster here
    float4 ra0, ra1;
    float4 rb0, rb1;
    bool rvec = false;

    // thread roles for the staging loads (per K step: A = 16 x BM, B = 16 x 128 floats)
    constexpr int ACOLS4 = BM / 4;                        // float4 per A row
    const int a_k = tid / ACOLS4, a_col = (tid % ACOLS4) * 4;    // A: rows a_k, a_k + 8
    const int v_k = tid >> 5, v_col = (tid & 31) * 4;     // vector B: row v_k (+8 when WM == 2)
    constexpr int BROWS = NT / 128;                        // scalar B: rows b_k + BROWS*i
    const int b_n = tid & 127, b_k = tid >> 7;

    auto load_tiles = [&](int s, int c0) {
      const Seg& sg = a.seg[s];
      const float* wp = sg.w + (long)(c0 + a_k) * sg.ldw + m0 + a_col;
      ra0 = *reinterpret_cast<const float4*>(wp);
      ra1 = *reinterpret_cast<const float4*>(wp + 8L * sg.ldw);
      const float* xb = sg.x + (long)b * sg.x_bstride;
      const int tw = t0 * sg.tmul + sg.toff;   // window start (when tmul==1,tdiv==1)
      rvec = sg.vec && ((tw & 3) == 0) && tw >= 0 && (tw + BN) <= sg.Tin;
      if (rvec) {
        const int ci0 = c0 + v_k, ci1 = c0 + v_k + 8;
        rb0 = make_float4(0.f, 0.f, 0.f, 0.f);
        rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ci0 < sg.cin) rb0 = *reinterpret_cast<const float4*>(xb + (long)ci0 * sg.x_cstride + tw + v_col);
        if (WM == 2 && ci1 < sg.cin) rb1 = *reinterpret_cast<const float4*>(xb + (long)ci1 * sg.x_cstride + tw + v_col);
      } else {
        const int tnum = (t0 + b_n) * sg.tmul + sg.toff;
        bool ok = tnum >= 0;
        int tin = tnum;
        if (sg.tdiv > 1) { ok = ok && (tnum % sg.tdiv == 0); tin = tnum / sg.tdiv; }
        ok = ok && tin < sg.Tin;
        const float* xp = xb + (long)(c0 + b_k) * sg.x_cstride + tin;
        const long csr = (long)BROWS * sg.x_cstride;
        const int cb = c0 + b_k;
        rb0.x = (ok && cb + 0 * BROWS < sg.cin) ? xp[0 * csr] : 0.f;
        rb0.y = (ok && cb + 1 * BROWS < sg.cin) ? xp[1 * csr] : 0.f;
        rb0.z = (ok && cb + 2 * BROWS < sg.cin) ? xp[2 * csr] : 0.f;
        rb0.w = (ok && cb + 3 * BROWS < sg.cin) ? xp[3 * csr] : 0.f;
        if (WM == 2) {
          rb1.x = (ok && cb + 4 * BROWS < sg.cin) ? xp[4 * csr] : 0.f;
          rb1.y = (ok && cb + 5 * BROWS < sg.cin) ? xp[5 * csr] : 0.f;
          rb1.z = (ok && cb + 6 * BROWS < sg.cin) ? xp[6 * csr] : 0.f;
          rb1.w = (ok && cb + 7 * BROWS < sg.cin) ? xp[7 * csr] : 0.f;
        }
      }
    };
    auto store_tiles = [&](int buf) {
      *reinterpret_cast<float4*>(&As[buf][a_k][a_col]) = ra0;
      *reinterpret_cast<float4*>(&As[buf][a_k + 8][a_col]) = ra1;
      if (rvec) {
        *reinterpret_cast<float4*>(&Bs[buf][v_k][v_col]) = rb0;
        if (WM == 2) *reinterpret_cast<float4*>(&Bs[buf][v_k + 8][v_col]) = rb1;
      } else {
        Bs[buf][b_k + 0 * BROWS][b_n] = rb0.x;  Bs[buf][b_k + 1 * BROWS][b_n] = rb0.y;
        Bs[buf][b_k + 2 * BROWS][b_n] = rb0.z;  Bs[buf][b_k + 3 * BROWS][b_n] = rb0.w;
        if (WM == 2) {
          Bs[buf][b_k + 4 * BROWS][b_n] = rb1.x;  Bs[buf][b_k + 5 * BROWS][b_n] = rb1.y;
          Bs[buf][b_k + 6 * BROWS][b_n] = rb1.z;  Bs[buf][b_k + 7 * BROWS][b_n] = rb1.w;
        }
      }
    };

    int s = 0, c0 = 0;
    int it_beg = 0, it_end = nk;
    if (EPI == EPI_LINEAR && a.ksplit > 1) {          // this workgroup's share of the K steps
      it_beg = ksp * a.ksteps_per_split;
      it_end = min(nk, it_beg + a.ksteps_per_split);
      int skip = it_beg;
      while (s < a.nseg) {
        const int steps = (a.seg[s].cin + BK - 1) / BK;
        if (skip < steps) { c0 = skip * BK; break; }
        skip -= steps; ++s;
      }
    }
    if (it_beg < it_end) { load_tiles(s, c0); store_tiles(0); }
    __syncthreads();

    for (int it = it_beg; it < it_end; ++it) {
      const int cur = (it - it_beg) & 1;
      const bool more = (it + 1) < it_end;
      if (more) {
        c0 += BK;
        if (c0 >= a.seg[s].cin) { c0 = 0; ++s; }
        load_tiles(s, c0);
      }
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
      if (more) store_tiles(cur ^ 1);
      __syncthreads();
    }
  }

  gemm_epilogue<EPI, WM, (EPI == EPI_LINEAR && WM == 2 && !BF16)>(a, acc, m0, t0, b, wm, wn, li, lk, ksp, tile_id, ntiles_all);
}


template <int EPI>
int launch_gemm_fp32(const GemmArgs& g, int wm, unsigned grid, int tag, hipStream_t st) {
  hipEvent_t pe0 = nullptr, pe1 = nullptr;
  const bool attach = tag != 0 && prof_attach(tag, &pe0, &pe1);
  if (wm == 4) {
    if (attach) hipExtLaunchKernelGGL((conv_gemm_kernel<EPI, 4, false>), dim3(grid), dim3(512), 0, st, pe0, pe1, 0, g);
    else hipLaunchKernelGGL((conv_gemm_kernel<EPI, 4, false>), dim3(grid), dim3(512), 0, st, g);
  } else {
    if (attach) hipExtLaunchKernelGGL((conv_gemm_kernel<EPI, 2, false>), dim3(grid), dim3(256), 0, st, pe0, pe1, 0, g);
    else hipLaunchKernelGGL((conv_gemm_kernel<EPI, 2, false>), dim3(grid), dim3(256), 0, st, g);
  }
  VQ_LAUNCH_CHECK();
  return 0;
}
template int launch_gemm_fp32<EPI_LINEAR>(const GemmArgs&, int, unsigned, int, hipStream_t);
template int launch_gemm_fp32<EPI_GATE>(const GemmArgs&, int, unsigned, int, hipStream_t);
template int launch_gemm_fp32<EPI_GATE_BWD>(const GemmArgs&, int, unsigned, int, hipStream_t);

}  // namespace vq
