// gemm_common.h -- what the MFMA contraction kernels of the hot path share (gfx950): the launch structures, the
// float32x2 scale bookkeeping, buffer / LDS-DMA plumbing, the cache-policy choices, the operand splits and the epilogues.
// Kernel families (one translation unit each, so that they build in parallel):
//   conv_gemm_x3.hip     conv_gemm_x3_kernel (modes 1-3: fwd and bwd-data of every conv, gate / gate-derivative GEMMs),
//                        lin128_stream_kernel, the split-K reduce, launch_gemm
//   conv_gemm_fp32.hip   conv_gemm_kernel (mode 0: v_mfma_f32_32x32x2_f32)
//   wgrad.hip            wgrad3_kernel / wgrad2_kernel / wgrad_kernel, their reduce, plan_wgrad, launch_wgrad
//   conv_api.hip         weight packing, maxima and norms (pack / wamax / absmax / wl1 kernels) and the C ABI of the conv,
//                        ResidualBlock and ResidualNet entry points
#pragma once
#include "common.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <map>
#include <mutex>
#include <type_traits>

namespace vq {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

// 0: fp32 operands on v_mfma_f32_32x32x2_f32; 1: operands rounded to bf16 (RNE, v_cvt_pk_bf16_f32),
// v_mfma_f32_32x32x16_bf16, fp32 accumulate; 2 (default): fp32 products as six bf16 MFMA products of an
// exact three-way split of both operands (conv_gemm_x3_kernel / wgrad3_kernel below).
// 3: fp32 products as THREE fp16 MFMA products of a two-way split of both operands scaled by a power of two
// per tensor (`float32x2`, section "matmul mode 3" below) wherever the caller provides the operands' absolute
// maxima (the ResidualNet chain, the large generic convs); mode 2's kernels everywhere else.
// HBM tensors, epilogues and accumulators stay fp32 in every mode.
extern int g_matmul_dtype;       // (conv_api.hip)
extern int g_wgrad_impl;         // 0: auto; 1: force the generic wgrad_kernel (tests / A-B timing)

constexpr int BM = 128, BN = 128, BK = 16, NT = 256;   // wgrad tiles; conv_gemm derives BM/NT from WM

// Dev aid (-DVQ_PHASE_TIMING, tools/experiments/phases.py; never in the product build): s_memtime stamps at the phase
// boundaries of the two-tap float32x2 kernels, summed per epilogue kind over one workgroup thread -- how round 5 found
// the gate epilogue's 40 k cycles of dependent condition loads.  [EPI][0..3] = sums of prologue / K loop / condition step /
// epilogue ticks, [4] = workgroups, [5] / [6] (gate) = epilogue phase 1 (loads) / phase 2 issue.
#ifdef VQ_PHASE_TIMING
__device__ unsigned long long g_phase[3][8];
#define VQ_STAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#define VQ_PHASE_ADD(EP_, I_, V_) do { if (threadIdx.x == 0) atomicAdd(&g_phase[EP_][I_], (unsigned long long)(V_)); } while (0)
#else
#define VQ_STAMP(v)
#define VQ_PHASE_ADD(EP_, I_, V_)
#endif
constexpr int MAXSEG = 24;   // a whole ResidualNet's blocks can feed one contraction
constexpr int MAXTAPS = 4;

struct Seg {
  const float* x;      // activations, channel 0 of this segment
  long x_bstride;      // elements between batch items
  int x_cstride;       // elements between channels (= time length of x)
  int cin;             // contraction length of this segment
  int Tin;             // valid input times [0, Tin)
  int tmul, toff, tdiv;  // t_in = (t_out*tmul + toff) / tdiv  (must divide exactly)
  int vec;             // host says: strides/pointer allow aligned float4 rows
  const float* w;      // packed A^T slab [cin_pad16][ldw]
  int ldw;
  // float32x2 (NP = 2) only: where the absolute maximum of the activation tensor is (device, float bits; any
  // upper bound will do) or, with amax == nullptr, a host-known bound; and the absolute maximum pack_kernel
  // scaled this slab's weights by
  const unsigned* amax; float amax_static;
  const unsigned* wamax;
};

struct OutR {          // one row range of M
  float* y; long y_bstride;
  const float* add; long add_bstride;   // residual add / gates input
  const float* bias; const float* bias2;
  int rows; int accumulate; int relu;
  unsigned* amax_out;  // range 0 only, nullable: atomicMax of |y| over everything this launch stores (float bits)
  // add_is_mask != 0: `add` is not added but gates the result -- y = add > 0 ? value : 0: the backward of a ReLU whose OUTPUT
  // is the tensor this GEMM's result is the gradient of (conv1d_bwd_data_relu), applied where the gradient is produced
  // instead of in a pass of its own (three passes over 126 MB per step at the configs)
  int add_is_mask;
};
__device__ __forceinline__ float lin_combine(float acc, float p, int is_mask) { return is_mask ? (p > 0.f ? acc : 0.f) : acc + p; }

enum { EPI_LINEAR = 0, EPI_GATE = 1, EPI_GATE_BWD = 2 };

struct Lerp {           // epilogue add of an up-sampled latent-rate tensor (align-corners lerp)
  const float* P; long p_bstride; int Tl;
  const int* v0; const float* w0; const float* w1;
  // fold != 0 (the two-tap 256 x 128-tile gate kernels, modes 2 / 3): the lerp runs on the matrix pipe as ONE more K step
  // instead of 128 dependent loads per lane in the epilogue -- see "the condition as a K step" in conv_gemm_x3_kernel;
  // amax: max |P| (float32x2: it joins the launch's product scale)
  int fold; const unsigned* amax;
};

struct GemmArgs {
  Lerp lerp;
  Seg seg[MAXSEG];
  int nseg;
  int M;         // logical rows (packed rows for EPI_GATE)
  int Tout;
  int B;
  int ntile_m, ntile_n;
  OutR out[2];
  // split-K (EPI_LINEAR, 128-row tiles, fp32): a latent-rate GEMM has a handful of output tiles
  // and a long contraction (the condition gradient: 32 tiles, K = 5120), so ksplit > 1 workgroups
  // share a tile, each over ksteps_per_split K steps, writing raw partial tiles to `partial`
  // ([split][tile][128][128]); gemm_splitk_reduce_kernel sums them in split order and applies
  // the epilogue.
  int ksplit, ksteps_per_split;
  float* partial;
  // device-side conditional launch: when non-null and *skip_flag != 0 every workgroup returns at
  // once (the one-hot embed conv launches its gather form and this dense form; a flag computed on
  // the device picks one of them without a host round trip)
  const int32_t* skip_flag;
  int x_nt;      // the activations of every segment are read once by this launch and by nothing soon after (the skip sum over all blocks' z): non-temporal loads
  int f16x2;     // matmul mode 3: every segment carries its maxima and a format-3 slab -> the float32x2 kernels (NP = 2); otherwise mode 3 runs mode 2's
  int g16;       // matmul mode 1 only: the gate values (EPI_GATE: out[0]; EPI_GATE_BWD: out[0].add) are stored as bf16, the pair (tanh, sigmoid) of a (channel, t) as one dword in tanh's fp32 position
  int x16;       // matmul mode 1 only: activations STORED as bf16 (conv_gemm_x3_kernel's X16 mask: bit 0 = segment 0 of a two-tap launch / every segment otherwise, bit 1 = the second segment of a two-tap launch)
  int add16, y16; // matmul mode 1 only, the streaming residual 1x1 (lin128_stream_kernel): out[0].add is read / out[0].y is stored as bf16 (the residual stream x_l, vqvae_resblock_desc::storage)
  int h16;       // matmul mode 1 only: EPI_GATE_BWD stores gh (out[0].y) as bf16 (same element strides, 2-byte elements)
  int z16;       // matmul mode 1 only: EPI_GATE writes z (out[1]) as bf16; a linear GEMM reads the activations of EVERY segment as bf16 (the z tensors)
  // matmul mode 3 (float32x2 launches), PRE-SPLIT storage (see presplit_pair): x16 = the same segment mask, here "stored
  // as fp16 hi | lo dwords" (same addresses as fp32; the segment's `amax` words are the scale words its producer wrote);
  // h16 = EPI_GATE_BWD stores gh that way under the bound sum_seg bound_l1[seg] * max|x_seg|, published to scale_out
  const float* bound_l1; unsigned* scale_out;
  // ... and the streaming residual 1x1 (lin128_stream_kernel): add16 / y16 = x_l read / x_{l+1} stored pre-split; add_scale =
  // the scale words of x_l, add_amax = its ACTUAL maximum, bound_l1[0] = max_r (sum_c |Wr[r][c]| + |br[r]|)
  const unsigned* add_scale; const unsigned* add_amax;
  // ... whose scale also leaves room for the NEXT block's condition step (see "the condition as a K step"): the exponent
  // of x_{l+1}'s scale is at least e(max |P|) - e(max |Wd_{l+1}|) - 1 (floor_p / floor_w: those maxima; NULL = no floor)
  const unsigned* floor_w; const unsigned* floor_p;
  // EPI_GATE_BWD, float32x2 (OUT bit 1): the pull-back of gh to the latent rate (the adjoint of the gate kernels' condition
  // lerp, net.py:54-55) in this epilogue -- every workgroup leaves the sums of its 128 columns for the four latent positions
  // under them in pb_part[b][column tile][2 Ch][4]; lerp.v0 / w0 / w1 are the resize tables (pullback_reduce_kernel finishes)
  float* pb_part;
  // gsig (vqvae_resblock_desc::storage & VQVAE_STORE_GATES_SIG): the saved gate values are sigmoid and z = tanh * sigmoid only.
  // EPI_GATE does not store the tanh half of out[0] (a third of its 189 MB of stores: gate launch 106 -> 88 us); EPI_GATE_BWD
  // reads z from zsrc (B, Ch, T) where it used to read tanh, and takes tanh = z / sigmoid (z = fl(tanh * sigmoid): tanh to
  // 2^-23 relative; sigmoid == 0 => z == 0 and both derivatives vanish whatever tanh is taken to be)
  int gsig; const float* zsrc;
};

// Gate non-linearities on the hardware exp/rcp units (v_exp_f32, v_rcp_f32): absolute error
// ~1e-7, far inside the 1e-4 parity tolerance, and ~8x fewer VALU instructions than libm's
// tanhf in the epilogue of the hottest kernel.
__device__ __forceinline__ float fast_tanhf_(float x) {
  const float e = __expf(-2.f * fabsf(x));
  const float r = (1.f - e) * __builtin_amdgcn_rcpf(1.f + e);     // v_rcp_f32: 1 ulp; __fdividef expands to a full division here
  return copysignf(r, x);
}
__device__ __forceinline__ float sigmoidf_(float x) {
  const float e = __expf(-fabsf(x));
  const float r = __builtin_amdgcn_rcpf(1.f + e);      // sigmoid(|x|)
  return x >= 0.f ? r : 1.f - r;
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {      // RNE, v_cvt_pk_bf16_f32
  bf16x2 v;
  v[0] = (__bf16)lo; v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}

// float32x2 scale bookkeeping.  A tensor's absolute maximum travels as the bit pattern of a non-negative float
// (unsigned compare == float compare), spread over AMAX_SLOTS words: a producer raises ONE of them per workgroup
// (atomicMax on slot blockIdx & 15 -- thousands of same-address atomics per launch cost 50 us, measured), a consumer
// takes the maximum of all.  Scales are powers of two taken from its exponent: 2^(14 - e) puts a tensor with amax in
// [2^e, 2^(e+1)) into [2^14, 2^15) < 65504.
constexpr int AMAX_SLOTS = 16;
__device__ __forceinline__ int amax_expo(unsigned bits) {        // unbiased exponent; zero / denormal amax -> -126
  const int e = (int)((bits >> 23) & 0xffu);
  return (e < 1 ? 1 : e) - 127;
}
__device__ __forceinline__ unsigned amax_load(const unsigned* p) {      // max over the slots, wave-uniform
  unsigned v = p[threadIdx.x & (AMAX_SLOTS - 1)];
#pragma unroll
  for (int o = AMAX_SLOTS / 2; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}
// m >= 0: this thread's max |value stored|.  EVERY thread of the workgroup must call (two barriers).
__device__ __forceinline__ void amax_commit(float m, unsigned* dst) {
  __shared__ float red[16];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (int)(blockDim.x >> 6);
    for (int i = 1; i < nw; ++i) m = fmaxf(m, red[i]);
    atomicMax(dst + (blockIdx.x & (AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, m));
  }
  __syncthreads();
}

// Buffer-resource addressing for the epilogues: base in four SGPRs, one 32-bit VGPR byte offset per
// lane and a wave-uniform SGPR offset per row -- no per-element 64-bit address arithmetic on the VALU
// (flat global_load/store cost a v_lshl_add_u64 and friends per access, ~40 % of the epilogue VALU).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
// LDS-DMA plumbing (conv_gemm_x3_kernel's ADMA): a resource descriptor as four SGPRs for inline asm, an LDS byte address, and one
// buffer_load_dwordx4 ... lds = LDS[m0 + 16 lane] <- buffer[voff + soff] (16 bytes per lane, 1 KB per wave)
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t make_rsrc4(const void* p) {
  const unsigned long long a = (unsigned long long)p;
  i32x4_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r[2] = 0x7fffffff;
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ unsigned lds_addr32(const void* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)p;
}
__device__ __forceinline__ void lds_dma16(unsigned lds_dst, unsigned voff, i32x4_t rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ float buf_ld(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st(float v, rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0);
}
// Cache policy of the epilogues' stores (2 = non-temporal), A/B'd per epilogue at configs[1] on one box:
#ifndef X3_LIN_ST_AUX
#define X3_LIN_ST_AUX 2       // linear epilogue (interior tiles): -0.1 ms per step
#endif
#ifndef X3_LIN_ADD_AUX
#define X3_LIN_ADD_AUX 2       // linear epilogue: the added operand (g_res in the backward-data launches: its last use before the res weight gradients) is read non-temporally: step -0.1 ms
#endif
#ifndef X3_SKIP_X_NT
#define X3_SKIP_X_NT 0         // skip sum: every z read non-temporally -- neutral (21.24 vs 21.26 ms)
#endif
#ifndef X3_GBWD_B0_AUX
#define X3_GBWD_B0_AUX 0       // gate-derivative GEMM: g_res fetched non-temporally -- +0.1 ms
#endif
#ifndef X3_GBWD_ST_AUX
#define X3_GBWD_ST_AUX 0      // gate-derivative epilogue (gh, read by the next three launches): non-temporal +0.07 ms
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
#ifndef L128_Z_NT
#define L128_Z_NT 1           // z is streamed non-temporally (it was stored that way by the gate kernel; next reads: the skip sum after the stack, the backward): step -0.09 ms, and the gate kernel beside it -1.4 %
#endif
#ifndef L128_X_AUX
#define L128_X_AUX 2           // the residual operand x_l is read non-temporally (after this launch nobody needs it before the backward): the NEXT gate launch then finds its own operand still cached -- gate kernel 193 -> 181 us, step -0.1 ms
#endif
#ifndef L128_ST_AUX
#define L128_ST_AUX 0         // streaming residual 1x1 (the next block's input): non-temporal +0.12 ms
#endif
#ifndef X3_GBWD_LD_AUX
#define X3_GBWD_LD_AUX 2      // cache policy of the gate-derivative epilogue's loads of tanh / sigmoid (their last use): non-temporal, step -0.08 ms
#endif
#ifndef X3_GBWD_LD16_AUX
#define X3_GBWD_LD16_AUX 0
#endif
#ifndef X3_GATE_ST_AUX
#define X3_GATE_ST_AUX 2      // cache policy of the gate epilogue's three stores: non-temporal (tanh / sigmoid are next read in the backward pass; z by the next launch, which measured no slower for it).  Gate kernel 199.5 -> 195.5-196 us, step -0.1 ms
#endif
__device__ __forceinline__ void buf_st_gate(float v, rsrc_t r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, X3_GATE_ST_AUX);
}

__device__ __forceinline__ void presplit_pair(float x0, float x1, int k, unsigned& d0, unsigned& d1);   // (float32x2 pre-split storage: defined beside split2)

// WM = wavefronts along M: block tile (64*WM) x 128 with 128*WM threads.  WM = 4 (256 rows)
// halves the activation-tile loads per FLOP and is used whenever M >= 256.
// Epilogue of the conv GEMM kernels: acc[mi][ni] is the wave's 2 x 2 block of 32 x 32 accumulator tiles
// (rows m0 + wm*64 + mi*32, columns t0 + wn*64 + ni*32) of batch item b.  SPLITK: this instantiation
// may have been launched with ksplit > 1 (raw partial tiles out, gemm_splitk_reduce_kernel finishes).
// DEEP: the linear epilogue requests a whole block's operands up front (needs 64 more registers).
// ST16: the instantiation may be asked for bf16-stored tensors (GemmArgs::g16 / h16 / z16: matmul mode 1's kernels only --
// the other modes' kernels do not carry those paths: they cost the float32x2 gate-derivative kernel 24 spilled registers).
// OUT = 1 (EPI_GATE_BWD, float32x2): gh is stored PRE-SPLIT under 2^kout (presplit_pair; `am` still collects the actual maximum).
template <int EPI, int WM, bool SPLITK, bool DEEP = false, bool ST16 = false, int OUT = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16 (&acc)[2][2], const int m0, const int t0,
                                              const int b, const int wm, const int wn, const int li, const int lk,
                                              const int ksp, const int tile_id, const int ntiles_all, [[maybe_unused]] const int kout = 0,
                                              [[maybe_unused]] const bool folded = false) {      // folded (EPI_GATE): the K loop added the condition term
  // ---- epilogue ----------------------------------------------------------
  // C/D layout of 32x32x2: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int T = a.Tout;
  float am = 0.f;                 // max |value stored| (published through out[0].amax_out when the caller asked for it)
  if (SPLITK && a.ksplit > 1) {
    float* pt = a.partial + ((long)ksp * ntiles_all + tile_id) * (128 * 128);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          pt[(wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk) * 128 + wn * 64 + ni * 32 + li] = acc[mi][ni][r];
    return;
  }
  if (EPI == EPI_LINEAR) {
    // ---- interior tiles: software-pipelined epilogue --------------------------------------------
    // VMEM operations retire through one in-order counter, so "load sub-tile q+1, then store
    // sub-tile q" lets the next operands travel while the previous results drain; the plain
    // load/store/load/store order exposed one load AND one store latency per sub-tile, which is
    // what bounded the K = 128 projections (315 MB of traffic per launch, 8 GFLOP).
    if constexpr (ST16) {
      // matmul mode 1, the residual GRADIENT stream kept as bf16 (GemmArgs::add16 / y16: the backward-data GEMM of the
      // packed chain; whole tiles, one output range, no bias / relu / accumulate -- the host guarantees all of it):
      // y = bf16(acc + add).  2-byte elements at the fp32 element strides; a lane pair (columns t, t + 1) shares the dwords
      // of a row pair and swaps halves by DPP, as in lin128_stream_kernel.  Same one-sub-tile look-ahead as below.
      if (a.add16 || a.y16) {
        const OutR& od = a.out[0];
        auto run16 = [&](auto add16c, auto y16c) {
          constexpr bool ADD16 = decltype(add16c)::value, Y16 = decltype(y16c)::value;
          const rsrc_t rs = make_rsrc(reinterpret_cast<const char*>(od.add) + (long)b * od.add_bstride * (ADD16 ? 2 : 4));
          const rsrc_t ry = make_rsrc(reinterpret_cast<char*>(od.y) + (long)b * od.y_bstride * (Y16 ? 2 : 4));
          float pv[2][16];
#pragma unroll
          for (int q = 0; q <= 4; ++q) {
            if (q < 4) {
              const int mi = q >> 1, ni = q & 1;
              const unsigned voff = 4u * (unsigned)(4 * lk * T + wn * 64 + ni * 32 + li);
              const unsigned voff16 = 2u * (unsigned)((4 * lk + (li & 1)) * T + wn * 64 + ni * 32 + (li & ~1));
              const unsigned sbase = 4u * (unsigned)((m0 + wm * 64 + mi * 32) * T + t0);
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const unsigned so = sbase + 4u * (unsigned)(((r & 3) + 8 * (r >> 2)) * T);
                if (od.add == nullptr) pv[q & 1][r] = 0.f;
                else if constexpr (ADD16) { if ((r & 1) == 0) pv[q & 1][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff16, so >> 1, X3_LIN_ADD_AUX)); }
                else pv[q & 1][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, so, X3_LIN_ADD_AUX));
              }
            }
            if (q > 0) {
              const int p = q - 1, mi = p >> 1, ni = p & 1;
              const unsigned voff = 4u * (unsigned)(4 * lk * T + wn * 64 + ni * 32 + li);
              const unsigned voff16 = 2u * (unsigned)((4 * lk + (li & 1)) * T + wn * 64 + ni * 32 + (li & ~1));
              const unsigned sbase = 4u * (unsigned)((m0 + wm * 64 + mi * 32) * T + t0);
#pragma unroll
              for (int r = 0; r < 16; r += 2) {            // rows R = ... + (r & 3) + 8 (r >> 2) and R + 1
                const unsigned so = sbase + 4u * (unsigned)(((r & 3) + 8 * (r >> 2)) * T);
                float va = acc[mi][ni][r], vb = acc[mi][ni][r + 1];
                if (ADD16 && od.add != nullptr) {
                  const unsigned own = __builtin_bit_cast(unsigned, pv[p & 1][r]);
                  const unsigned got = (unsigned)__shfl_xor((int)own, 1);
                  const unsigned ra_ = (li & 1) ? got : own, rb_ = (li & 1) ? own : got;     // (row R, row R + 1) x columns (t, t + 1) of the pair
                  va += __builtin_bit_cast(float, (li & 1) ? (ra_ & 0xffff0000u) : (ra_ << 16));
                  vb += __builtin_bit_cast(float, (li & 1) ? (rb_ & 0xffff0000u) : (rb_ << 16));
                } else { va += pv[p & 1][r]; vb += pv[p & 1][r + 1]; }
                if constexpr (Y16) {
                  const unsigned h = pack_bf16x2(va, vb);
                  const unsigned send = (li & 1) ? (h & 0xffffu) : (h >> 16);
                  const unsigned got = (unsigned)__shfl_xor((int)send, 1);
                  const unsigned pr = (li & 1) ? (got | (h & 0xffff0000u)) : ((h & 0xffffu) | (got << 16));
                  __builtin_amdgcn_raw_buffer_store_b32((int)pr, ry, voff16, so >> 1, X3_LIN_ST_AUX);
                } else {
                  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, va), ry, voff, so, X3_LIN_ST_AUX);
                  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vb), ry, voff, so + 4u * (unsigned)T, X3_LIN_ST_AUX);
                }
              }
            }
          }
        };
        if (a.add16 && a.y16) run16(std::true_type{}, std::true_type{});
        else if (a.y16) run16(std::false_type{}, std::true_type{});
        else run16(std::true_type{}, std::false_type{});
        return;
      }
    }
    bool fast = (t0 + BN <= T);
    {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int mb = m0 + wm * 64 + mi * 32;
        const int o = (mb < a.out[0].rows) ? 0 : 1;
        const int rows_left = (o ? a.M - a.out[0].rows : min(a.M, a.out[0].rows));
        const int mr0 = o ? mb - a.out[0].rows : mb;
        fast = fast && (mr0 + 32 <= rows_left) && !(a.out[o].add && a.out[o].accumulate);
      }
    }
    if (__builtin_amdgcn_readfirstlane(fast ? 1 : 0)) {
      // bias first, one row group at a time (the registers are needed for the operand pipeline)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int mb = m0 + wm * 64 + mi * 32;
        const int o = __builtin_amdgcn_readfirstlane((mb < a.out[0].rows) ? 0 : 1);
        const OutR& od = a.out[o];
        if (od.bias) {
          const int mrb = (o ? mb - a.out[0].rows : mb) + 4 * lk;
          float bias[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) bias[r] = od.bias[mrb + (r & 3) + 8 * (r >> 2)];
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][n2][r] += bias[r];
        }
      }
      if constexpr (DEEP) {
        // every operand of the block is requested before the first store (64 loads in flight per lane:
        // the registers of the main loop's staging are free now); a store then only waits for ITS
        // sub-tile's loads (counted vmcnt).  With one sub-tile of look-ahead the K = 128 residual
        // projection spent as long in this epilogue as in the rest of the kernel (71 of 142 us).
        float pv[4][16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int mi = q >> 1, ni = q & 1;
          const int mb = m0 + wm * 64 + mi * 32;
          const int o = __builtin_amdgcn_readfirstlane((mb < a.out[0].rows) ? 0 : 1);
          const OutR& od = a.out[o];
          const unsigned voff = 4u * (unsigned)(4 * lk * T + wn * 64 + ni * 32 + li);
          const unsigned sbase = 4u * (unsigned)((o ? mb - a.out[0].rows : mb) * T + t0);
          const float* src = od.add ? od.add + (long)b * od.add_bstride
                                    : (od.accumulate ? od.y + (long)b * od.y_bstride : nullptr);
          if (src) {
            const rsrc_t rs = make_rsrc(src);
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[q][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, sbase + 4u * (unsigned)(((r & 3) + 8 * (r >> 2)) * T), X3_LIN_ADD_AUX));
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[q][r] = 0.f;
          }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int mi = p >> 1, ni = p & 1;
          const int mb = m0 + wm * 64 + mi * 32;
          const int o = __builtin_amdgcn_readfirstlane((mb < a.out[0].rows) ? 0 : 1);
          const OutR& od = a.out[o];
          const unsigned voff = 4u * (unsigned)(4 * lk * T + wn * 64 + ni * 32 + li);
          const unsigned sbase = 4u * (unsigned)((o ? mb - a.out[0].rows : mb) * T + t0);
          const rsrc_t ry = make_rsrc(od.y + (long)b * od.y_bstride);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = lin_combine(acc[mi][ni][r], pv[p][r], od.add_is_mask);
            if (od.relu) v = fmaxf(v, 0.f);
            am = fmaxf(am, fabsf(v));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (float)(v)), ry, voff, sbase + 4u * (unsigned)(((r & 3) + 8 * (r >> 2)) * T), X3_LIN_ST_AUX);
          }
        }
      } else {
        // LOOK sub-tiles of operands in flight ahead of the one being finished
        constexpr int LOOK = 2;      // (three in flight for the 256 x 128-tile kernels: epilogue 32 k -> 41 k cycles, measured: the phase is bound by the CU's ~10 B / cycle memory path, not by latency)
        float pv[LOOK][16];
        auto request = [&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const int mi = q >> 1, ni = q & 1;
          const int mb = m0 + wm * 64 + mi * 32;
          const int o = __builtin_amdgcn_readfirstlane((mb < a.out[0].rows) ? 0 : 1);
          const OutR& od = a.out[o];
          const unsigned voff = 4u * (unsigned)(4 * lk * T + wn * 64 + ni * 32 + li);
          const unsigned sbase = 4u * (unsigned)((o ? mb - a.out[0].rows : mb) * T + t0);
          const float* src = od.add ? od.add + (long)b * od.add_bstride
                                    : (od.accumulate ? od.y + (long)b * od.y_bstride : nullptr);
          if (src) {
            const rsrc_t rs = make_rsrc(src);
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[q % LOOK][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, sbase + 4u * (unsigned)(((r & 3) + 8 * (r >> 2)) * T), X3_LIN_ADD_AUX));
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[q % LOOK][r] = 0.f;
          }
        };
        auto finish = [&](auto pc) {
          constexpr int p = decltype(pc)::value;
          const int mi = p >> 1, ni = p & 1;
          const int mb = m0 + wm * 64 + mi * 32;
          const int o = __builtin_amdgcn_readfirstlane((mb < a.out[0].rows) ? 0 : 1);
          const OutR& od = a.out[o];
          const unsigned voff = 4u * (unsigned)(4 * lk * T + wn * 64 + ni * 32 + li);
          const unsigned sbase = 4u * (unsigned)((o ? mb - a.out[0].rows : mb) * T + t0);
          const rsrc_t ry = make_rsrc(od.y + (long)b * od.y_bstride);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = lin_combine(acc[mi][ni][r], pv[p % LOOK][r], od.add_is_mask);
            if (od.relu) v = fmaxf(v, 0.f);
            am = fmaxf(am, fabsf(v));
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (float)(v)), ry, voff, sbase + 4u * (unsigned)(((r & 3) + 8 * (r >> 2)) * T), X3_LIN_ST_AUX);
          }
        };
        using Q0 = std::integral_constant<int, 0>; using Q1 = std::integral_constant<int, 1>;
        using Q2 = std::integral_constant<int, 2>; using Q3 = std::integral_constant<int, 3>;
        request(Q0{}); request(Q1{});
        if constexpr (LOOK == 3) { request(Q2{}); finish(Q0{}); request(Q3{}); finish(Q1{}); }
        else { finish(Q0{}); request(Q2{}); finish(Q1{}); request(Q3{}); }
        finish(Q2{}); finish(Q3{});
      }
    } else {
    // ---- edge tiles: fully predicated ---------------------------------------------------------
      // All loads of a 32x32 sub-tile (bias, residual, old value) are issued before its
      // first store, so they overlap instead of serialising behind may-alias stores.
  #pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int mb = m0 + wm * 64 + mi * 32;
        // the host guarantees out[0].rows % 32 == 0 when two ranges exist: wave-uniform
        const int o = __builtin_amdgcn_readfirstlane((mb < a.out[0].rows) ? 0 : 1);
        const OutR& od = a.out[o];
        const int mrb = (o ? mb - a.out[0].rows : mb) + 4 * lk;
        const int rows_left = (o ? a.M - a.out[0].rows : min(a.M, a.out[0].rows));
        float bias[16];
  #pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mr = mrb + (r & 3) + 8 * (r >> 2);
          bias[r] = (od.bias && mr < rows_left) ? od.bias[mr] : 0.f;
        }
  #pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int t = t0 + wn * 64 + ni * 32 + li;
          const bool tok = t < T;
          float addv[16], oldv[16];
          const long boff = (long)mrb * T + t;
          if (od.add) {
            const float* ap = od.add + (long)b * od.add_bstride + boff;
  #pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int dr = (r & 3) + 8 * (r >> 2);
              addv[r] = (tok && mrb + dr < rows_left) ? ap[(long)dr * T] : 0.f;
            }
          } else {
  #pragma unroll
            for (int r = 0; r < 16; ++r) addv[r] = 0.f;
          }
          float* yp = od.y + (long)b * od.y_bstride + boff;
          if (od.accumulate) {
  #pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int dr = (r & 3) + 8 * (r >> 2);
              oldv[r] = (tok && mrb + dr < rows_left) ? yp[(long)dr * T] : 0.f;
            }
          } else {
  #pragma unroll
            for (int r = 0; r < 16; ++r) oldv[r] = 0.f;
          }
  #pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (tok && mrb + dr < rows_left) {
              float v = lin_combine(acc[mi][ni][r] + bias[r], addv[r], od.add_is_mask) + oldv[r];
              if (od.relu) v = fmaxf(v, 0.f);
              am = fmaxf(am, fabsf(v));
              yp[(long)dr * T] = v;
            }
          }
        }
      }
    }
    if (a.out[0].amax_out != nullptr) amax_commit(am, a.out[0].amax_out);
  } else if (EPI == EPI_GATE) {
    // packed rows: each 64-row wave tile = 32 tanh rows (mi=0) + the matching 32
    // sigmoid rows (mi=1) of channel group g.
    const int Ch = a.M >> 1;
    const int g = (m0 + wm * 64) >> 6;
    const OutR& og = a.out[0];   // gates (B, 2Ch, T)
    const OutR& oz = a.out[1];   // z (B, Ch, T)
    // Phase 1 -- pre-activations completed in place in the accumulators: biases and the lerp of the
    // latent-rate condition projection.  No store has been issued yet, so all of these loads overlap
    // (a load behind a may-alias store would wait for the store's acknowledgement: one in-order counter).
    const float* Pb = (a.lerp.P && !folded) ? a.lerp.P + (long)b * a.lerp.p_bstride : nullptr;      // (folded: the K loop added it)
    int tt[2], vv[2];
    float w0v[2], w1v[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      tt[ni] = t0 + wn * 64 + ni * 32 + li;
      const bool tok = tt[ni] < T;
      vv[ni] = (tok && Pb) ? a.lerp.v0[tt[ni]] : 0;
      w0v[ni] = (tok && Pb) ? a.lerp.w0[tt[ni]] : 0.f;
      w1v[ni] = (tok && Pb) ? a.lerp.w1[tt[ni]] : 0.f;
    }
    const rsrc_t rP = make_rsrc(Pb);
    const int chl = 32 * g + 4 * lk;            // this lane's first channel; row r adds (r&3) + 8*(r>>2)
    unsigned vP[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) vP[ni] = 4u * (unsigned)(chl * a.lerp.Tl + vv[ni]);
    const unsigned sPq = 4u * (unsigned)(Ch * a.lerp.Tl);
    if (Pb || og.bias || og.bias2)       // (wave-uniform; nothing to add when the K loop folded the condition and P carries the biases)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = (r & 3) + 8 * (r >> 2);
      const int ch = chl + dr;
      if (ch >= Ch) continue;
      float ba = 0.f, bb = 0.f;
      if (og.bias) { ba += og.bias[ch]; bb += og.bias[Ch + ch]; }
      if (og.bias2) { ba += og.bias2[ch]; bb += og.bias2[Ch + ch]; }
      const unsigned sP = 4u * (unsigned)(dr * a.lerp.Tl);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        float pa = 0.f, pb = 0.f;
        if (Pb) {      // h += upsample(P)[t]: condition projected at latent rate
          pa = w0v[ni] * buf_ld(rP, vP[ni], sP) + w1v[ni] * buf_ld(rP, vP[ni] + 4u, sP);
          pb = w0v[ni] * buf_ld(rP, vP[ni], sP + sPq) + w1v[ni] * buf_ld(rP, vP[ni] + 4u, sP + sPq);
        }
        acc[0][ni][r] = (acc[0][ni][r] + ba) + pa;
        acc[1][ni][r] = (acc[1][ni][r] + bb) + pb;
      }
    }
#ifdef VQ_PHASE_TIMING
    VQ_STAMP(te0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VQ_STAMP(te1);
    VQ_PHASE_ADD(1, 5, te1 - te0);
#endif
    // Phase 2 -- gate and the three stores per element
    const rsrc_t rG = make_rsrc(og.y + (long)b * og.y_bstride);
    // z is read only through GEMM staging; in matmul mode 1 that staging rounds it to bf16 anyway, so it is
    // STORED as bf16 there (a.z16; same element strides, 2-byte elements): identical results, half the bytes
    const rsrc_t rZ = make_rsrc(reinterpret_cast<const char*>(oz.y) + (long)b * oz.y_bstride * (a.z16 ? 2 : 4));
    unsigned vT[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) vT[ni] = 4u * (unsigned)(chl * T + tt[ni]);
    const unsigned sGq = 4u * (unsigned)(Ch * T);
    auto gate_store = [&](auto sigc) {            // (ONE wave-uniform branch around the store loop: GemmArgs::gsig)
    constexpr bool SIG = decltype(sigc)::value;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dr = (r & 3) + 8 * (r >> 2);
      if (chl + dr >= Ch) continue;
      const unsigned sT = 4u * (unsigned)(dr * T);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        if (tt[ni] >= T) continue;
        const float ta = fast_tanhf_(acc[0][ni][r]);
        const float sb = sigmoidf_(acc[1][ni][r]);
        if constexpr (SIG) {       // sigmoid and z only: the backward takes tanh = z / sigmoid
          buf_st_gate(sb, rG, vT[ni], sT + sGq);
        } else
        if (ST16 && a.g16) {     // BASELINE configs[4] precision: the saved gate values are bf16 (the backward pass reads exactly these): the pair (tanh, sigmoid) of one (channel, t) as ONE dword in tanh's fp32 position -- one store, and one load in the backward, instead of two
          __builtin_amdgcn_raw_buffer_store_b32((int)pack_bf16x2(ta, sb), rG, vT[ni], sT, X3_GATE_ST_AUX);
        } else {
          buf_st_gate(ta, rG, vT[ni], sT);
          buf_st_gate(sb, rG, vT[ni], sT + sGq);
        }
        if (ST16 && a.z16) __builtin_amdgcn_raw_buffer_store_b16((short)(pack_bf16x2(ta * sb, 0.f) & 0xffffu), rZ, vT[ni] >> 1, sT >> 1, 0);
        else buf_st_gate(ta * sb, rZ, vT[ni], sT);
      }
    }
    };
    if (a.gsig) gate_store(std::true_type{}); else gate_store(std::false_type{});
  } else {  // EPI_GATE_BWD: rows are gz channels; add = gates (B,2Ch,T); y = gh (B,2Ch,T)
    const int Ch = a.M;
    const OutR& od = a.out[0];
    const rsrc_t rGt = make_rsrc(od.add + (long)b * od.add_bstride);
    const rsrc_t rGh = make_rsrc(od.y + (long)b * od.y_bstride);
    const unsigned sQ = 4u * (unsigned)(Ch * T);
    if constexpr (DEEP) {       // (the x3 kernels: 256 VGPRs allowed)
      // ALL gate values of the wave's four 32 x 32 sub-tiles are requested before the first store: loads and
      // stores retire through one in-order counter, so a sub-tile's loads issued behind the previous sub-tile's
      // stores waited for those stores' acknowledgements -- four load + store round trips per tile, now one.
      float ta[2][2][16], sb[2][2][16];
      // (ONE wave-uniform branch around all of the loads: a branch per load makes hipcc drain vmcnt(0) at each)
      const rsrc_t rZs = make_rsrc(a.zsrc ? a.zsrc + (long)b * Ch * T : od.add);      // gsig: z (B, Ch, T) in tanh's place
      auto load_gates = [&](auto packedc) {
        constexpr int LMODE = decltype(packedc)::value;      // 0: tanh | sigmoid as two fp32, 1: one packed bf16 pair, 2: z | sigmoid (gsig)
        constexpr bool PACKED = LMODE == 1;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int t = t0 + wn * 64 + ni * 32 + li;
            const int mb = m0 + wm * 64 + mi * 32 + 4 * lk;
            const bool tok = t < T;
            const unsigned voff = 4u * (unsigned)(mb * T + t);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int dr = (r & 3) + 8 * (r >> 2);
              const bool ok = tok && mb + dr < Ch;
              const unsigned so = 4u * (unsigned)(dr * T);
              if constexpr (PACKED) {      // one dword = (bf16 tanh | bf16 sigmoid << 16), decoded below once ALL
                // are requested.  Unconditional loads (an out-of-range lane reads element 0; its value is never
                // used): a branch per load made hipcc wait for each load before the next was issued
                ta[mi][ni][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rGt, ok ? voff + so : 0u, 0, X3_GBWD_LD16_AUX));
              } else if constexpr (LMODE == 2) {
                const unsigned vo = ok ? voff + so : 0u;
                ta[mi][ni][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rZs, vo, 0, X3_GBWD_LD_AUX));
                sb[mi][ni][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rGt, vo, sQ, X3_GBWD_LD_AUX));
              } else {
                const unsigned vo = ok ? voff + so : 0u;
                ta[mi][ni][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rGt, vo, 0, X3_GBWD_LD_AUX));
                sb[mi][ni][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rGt, vo, sQ, X3_GBWD_LD_AUX));
              }
            }
          }
        if constexpr (PACKED) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const unsigned pr = __builtin_bit_cast(unsigned, ta[mi][ni][r]);
                ta[mi][ni][r] = __builtin_bit_cast(float, pr << 16);
                sb[mi][ni][r] = __builtin_bit_cast(float, pr & 0xffff0000u);
              }
        }
      };
      if (ST16 && a.g16) load_gates(std::integral_constant<int, 1>{});
      else if (a.gsig) load_gates(std::integral_constant<int, 2>{});
      else load_gates(std::integral_constant<int, 0>{});
      // ---- fused latent pull-back (OUT bit 1).  gP[c, v] = sum_t gh[c, t] W(t, v), W(t, v0[t]) = w0[t], W(t, v0[t] + 1) = w1[t]: the
      // 32 columns of a sub-tile touch the latent positions vs, vs + 1, vs + 2 (vs = v0 of its first column; T >= 64 Tl: the host
      // checks), the 128 columns of the tile vb .. vb + 3.  The accumulator layout has a lane per COLUMN; the sums run over columns,
      // so each wave transposes a sub-tile through LDS (64 gh rows x 32 t, row pitch 36 floats: conflict-free both ways) and
      // lane L then owns row L: 8 ds_read_b128 + 32 broadcast reads of the columns' coefficient triples + 96 FMAs.  The stand-alone
      // kernel (upsample_bwd_seg_kernel) re-read all of gh, 126 MB per block, for the same sums: 36 us per block, 0.7 ms per step.
      constexpr bool PB = (OUT & 2) != 0;
      __shared__ __attribute__((aligned(16))) float pbG[PB ? 4 * 64 * 36 : 4];
      __shared__ float4 pbC[PB ? 128 : 1];
      __shared__ float pbT[PB ? 2 * 256 * 4 : 1];              // [column half wn][gh channel][position - vb]
      [[maybe_unused]] int pb_vb = 0;
      if constexpr (PB) {
        pb_vb = a.lerp.v0[t0];
        const int tid_ = (wm * 2 + wn) * 64 + lk * 32 + li;      // 0 .. 255
        if (tid_ < 128) {
          const int tc = min(t0 + tid_, T - 1);
          const int dv = a.lerp.v0[tc] - a.lerp.v0[min(t0 + (tid_ & ~31), T - 1)];      // 0 or 1
          const float c0 = t0 + tid_ < T ? a.lerp.w0[tc] : 0.f, c1 = t0 + tid_ < T ? a.lerp.w1[tc] : 0.f;
          pbC[tid_] = dv == 0 ? make_float4(c0, c1, 0.f, 0.f) : make_float4(0.f, c0, c1, 0.f);
        }
        for (int i = tid_; i < 2 * 256 * 4; i += 256) pbT[i] = 0.f;
        __syncthreads();
      }
      const bool gsig_ = a.gsig != 0;
      auto store_gh = [&](auto h16c) {         // (ONE wave-uniform branch around the whole store loop, as for the loads)
        constexpr bool H16 = decltype(h16c)::value;
        // H16 (GemmArgs::h16): gh is read back only as an MFMA operand (backward-data, weight gradient), i.e. rounded
        // to bf16 -- and by the bias sum and the latent pull-back, which then see the rounded values (the oracle's bf16
        // mode mirrors that): stored as bf16, same element strides
        const rsrc_t rGh16 = make_rsrc(reinterpret_cast<const char*>(od.y) + (long)b * od.y_bstride * 2);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            const int t = t0 + wn * 64 + ni * 32 + li;
            const int mb = m0 + wm * 64 + mi * 32 + 4 * lk;
            const bool tok = t < T;
            const unsigned voff = 4u * (unsigned)(mb * T + t);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int dr = (r & 3) + 8 * (r >> 2);
              if (tok && mb + dr < Ch) {
                const float gz = acc[mi][ni][r];
                const float sv = sb[mi][ni][r];
                // gsig: ta holds z = tanh * sigmoid; tanh = z / sigmoid, taken where it is used (a pass over the 64 values in
                // front of the stores cost 13 more registers than the kernel has)
                const float tv = gsig_ ? ta[mi][ni][r] * __builtin_amdgcn_rcpf(fmaxf(sv, 1e-30f)) : ta[mi][ni][r];
                const unsigned so = 4u * (unsigned)(dr * T);
                const float ga = gz * sv * (1.f - tv * tv), gb = gz * tv * sv * (1.f - sv);
                if constexpr (PB) {
                  float* gw = pbG + (wm * 2 + wn) * (64 * 36);
                  gw[(4 * lk + dr) * 36 + li] = ga;
                  gw[(32 + 4 * lk + dr) * 36 + li] = gb;
                }
                if constexpr (H16) {
                  const unsigned pr = pack_bf16x2(ga, gb);
                  __builtin_amdgcn_raw_buffer_store_b16((short)(pr & 0xffffu), rGh16, voff >> 1, so >> 1, X3_GBWD_ST_AUX);
                  __builtin_amdgcn_raw_buffer_store_b16((short)(pr >> 16), rGh16, voff >> 1, (so + sQ) >> 1, X3_GBWD_ST_AUX);
                } else if constexpr ((OUT & 1) != 0) {
                  am = fmaxf(am, fmaxf(fabsf(ga), fabsf(gb)));
                  unsigned da, db;
                  presplit_pair(ga, gb, kout, da, db);
                  __builtin_amdgcn_raw_buffer_store_b32((int)da, rGh, voff, so, X3_GBWD_ST_AUX);
                  __builtin_amdgcn_raw_buffer_store_b32((int)db, rGh, voff, so + sQ, X3_GBWD_ST_AUX);
                } else {
                  am = fmaxf(am, fmaxf(fabsf(ga), fabsf(gb)));
                  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, ga), rGh, voff, so, X3_GBWD_ST_AUX);
                  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, gb), rGh, voff, so + sQ, X3_GBWD_ST_AUX);
                }
              } else if constexpr (PB) {           // rows / columns beyond the tensor contribute nothing
                float* gw = pbG + (wm * 2 + wn) * (64 * 36);
                gw[(4 * lk + dr) * 36 + li] = 0.f;
                gw[(32 + 4 * lk + dr) * 36 + li] = 0.f;
              }
            }
            if constexpr (PB) {
              // lane L = 32 lk + li owns gh row L of this sub-tile (rows 0..31: ga of z channels mb0 .., rows 32..63: gb)
              const int L = 32 * lk + li;
              const float* gr = pbG + (wm * 2 + wn) * (64 * 36) + L * 36;
              const float4* cc = pbC + (wn * 2 + ni) * 32;
              float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                const float4 g4 = *reinterpret_cast<const float4*>(gr + 4 * q);
                const float4 ca = cc[4 * q], cb = cc[4 * q + 1], cd = cc[4 * q + 2], ce = cc[4 * q + 3];
                s0 = fmaf(g4.x, ca.x, s0); s1 = fmaf(g4.x, ca.y, s1); s2 = fmaf(g4.x, ca.z, s2);
                s0 = fmaf(g4.y, cb.x, s0); s1 = fmaf(g4.y, cb.y, s1); s2 = fmaf(g4.y, cb.z, s2);
                s0 = fmaf(g4.z, cd.x, s0); s1 = fmaf(g4.z, cd.y, s1); s2 = fmaf(g4.z, cd.z, s2);
                s0 = fmaf(g4.w, ce.x, s0); s1 = fmaf(g4.w, ce.y, s1); s2 = fmaf(g4.w, ce.z, s2);
              }
              // gh channel of row L: ga rows -> z channel, gb rows -> Ch + z channel; position offset of this sub-tile in the tile
              const int zc = m0 + wm * 64 + mi * 32 + (L & 31);
              const int ghc = (L >> 5) * Ch + zc;
              const int off = a.lerp.v0[min(t0 + wn * 64 + ni * 32, T - 1)] - pb_vb;      // 0 .. 2 (wave-uniform)
              if (zc < Ch) {
                float* tp = pbT + (wn * 2 * Ch + ghc) * 4;      // (2 Ch = 256 gh channels per column half)
                tp[off] += s0;
                tp[off + 1] += s1;
                if (off + 2 < 4) tp[off + 2] += s2;             // (s2 is exactly 0 when the sub-tile starts at vb + 2)
              }
            }
          }
      };
      if (ST16 && a.h16) store_gh(std::true_type{}); else store_gh(std::false_type{});
      if constexpr (PB) {
        __syncthreads();
        const int tid_ = (wm * 2 + wn) * 64 + lk * 32 + li;
        const int nt_ = t0 / BN;
        float* dst = a.pb_part + (((long)b * a.ntile_n + nt_) * (2 * Ch)) * 4;
        for (int i = tid_; i < 2 * Ch * 4; i += 256) dst[i] = pbT[i] + pbT[2 * Ch * 4 + i];      // column halves in a fixed order
      }
    } else {                    // the fp32 MFMA kernel runs four waves per SIMD (128 VGPRs): one sub-tile's gate values at a time
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int t = t0 + wn * 64 + ni * 32 + li;
          const int mb = m0 + wm * 64 + mi * 32 + 4 * lk;
          const bool tok = t < T;
          const unsigned voff = 4u * (unsigned)(mb * T + t);
          // all gate loads of the 32x32 sub-tile first, then the stores (may-alias ordering)
          float ta[16], sb[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            const bool ok = tok && mb + dr < Ch;
            const unsigned so = 4u * (unsigned)(dr * T);
            ta[r] = ok ? buf_ld(rGt, voff, so) : 0.f;
            sb[r] = ok ? buf_ld(rGt, voff, so + sQ) : 0.f;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            if (tok && mb + dr < Ch) {
              const float gz = acc[mi][ni][r];
              const unsigned so = 4u * (unsigned)(dr * T);
              const float ga = gz * sb[r] * (1.f - ta[r] * ta[r]), gb = gz * ta[r] * sb[r] * (1.f - sb[r]);
              am = fmaxf(am, fmaxf(fabsf(ga), fabsf(gb)));
              buf_st(ga, rGh, voff, so);
              buf_st(gb, rGh, voff, so + sQ);
            }
          }
        }
    }
    if (od.amax_out != nullptr) amax_commit(am, od.amax_out);
  }
}

// ---------------------------------------------------------------------------
// matmul mode 2: fp32 products on the bf16 matrix pipe.
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the rate of v_mfma_f32_32x32x16_bf16 on gfx950, so an fp32
// product is cheaper as six bf16 products of an EXACT three-way split of both operands:
//     x = x_h + x_m + x_l     (three bf16, each RNE of the remainder: 3 x 8 significand bits and
//                              the signs of the remainders cover all 24 bits of an fp32)
//     a*b ~= a_l*b_h + a_h*b_l + a_m*b_m + a_m*b_h + a_h*b_m + a_h*b_h        (fp32 accumulate)
// A product of two bf16 is exact in fp32, and the three dropped products (m*l, l*m, l*l) are below
// 2^-25 |a*b| -- less than the rounding of ONE fp32 multiply -- so the result is as accurate as the
// fp32 MFMA path (tests/test_gpu_kernels.py compares both with float64); it is not a reduced-
// precision mode like mode 1.  6 x 32 cycles per 16 k against 8 x 64: 0.375 of the MFMA time.
//
// Weights arrive already split from pack_kernel, in the order the LDS image wants: per 16-k step,
// [piece 3][k-half 2][m] 16-byte words (8 consecutive k of one piece).  Activations are split while
// they are staged: every thread owns one column of the tile and 16/NQ channels of the step, which
// is also what makes every window (stride, dilation shift, transposed-conv gaps) the same path.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = pack_bf16x2(x0, x1);
  asm("" : "+v"(h));        // opaque (not volatile: free to move): hipcc otherwise re-derives `h << 16` as a second conversion of x0 alone
  float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  m = pack_bf16x2(r0, r1);
  asm("" : "+v"(m));
  r0 -= __builtin_bit_cast(float, m << 16); r1 -= __builtin_bit_cast(float, m & 0xffff0000u);
  l = pack_bf16x2(r0, r1);
}

// ---------------------------------------------------------------------------
// matmul mode 3 (`float32x2`): fp32 products on the fp16 matrix pipe, THREE MFMAs per product.
//
//     x * 2^k = hi + lo          hi = fp16(x 2^k) (RNE), lo = fp16(x 2^k - hi)           (k: one power of two per tensor)
//     a*b ~= (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi) * 2^-(ka+kb)                            (fp32 accumulate in the MFMA)
//
// 2 x 11 significand bits plus the sign of the remainder: hi + lo is x to within 2^-24 |x| -- the size of one fp32
// rounding -- for every element within 2^-15 of the tensor's absolute maximum, and to within 2^-39 of that maximum for
// smaller ones (lo then lies in fp16's subnormal range, which v_mfma_f32_32x32x16_f16 honours: tools/ubench/
// f16x2_probe.hip); a product of two fp16 is exact in fp32; the dropped a_lo*b_lo is below 2^-22 |a*b|.  Against
// float64 the result is at or below the error of the fp32 MFMA path AND of mode 2's six bf16 products (same probe,
// K = 128 ... 2560, also with 6 decades of dynamic range inside a tensor and with 1e-7-sized gradients), because a
// K step of 16 products is rounded once where the fp32 MFMA rounds eight times.  The power of two needs the tensor's
// absolute maximum BEFORE the kernel runs: producers on this path publish it from their epilogues (OutR::amax_out),
// entry points whose operand comes from elsewhere run absmax_kernel first.  Several segments in one accumulator
// (taps, the g_res | g_skip pair, the skip sum over all blocks) share ONE product scale 2^(28 - emax),
// emax = max_j (e_w_j + e_x_j): segment j's activations are scaled by 2^(14 - emax + e_w_j) <= their own optimum, so
// nothing overflows and every segment's error stays below 2^-39 of the largest product any segment can contribute.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void split2(float x0, float x1, int k, unsigned& h, unsigned& l) {
  const float y0 = __builtin_ldexpf(x0, k), y1 = __builtin_ldexpf(x1, k);
  f16x2 hv;
  hv[0] = (_Float16)y0; hv[1] = (_Float16)y1;
  h = __builtin_bit_cast(unsigned, hv);
  asm("" : "+v"(h));
  const f16x2 hb = __builtin_bit_cast(f16x2, h);
  f16x2 lv;
  lv[0] = (_Float16)(y0 - (float)hb[0]); lv[1] = (_Float16)(y1 - (float)hb[1]);
  l = __builtin_bit_cast(unsigned, lv);
}
// ---------------------------------------------------------------------------
// `float32x2`, PRE-SPLIT storage (vqvae_resblock_desc::storage & VQVAE_STORE_*_F16X2).
//
// A tensor of ResidualNet's chain that is read back only as a float32x2 MFMA operand -- the residual stream x_l (gate
// GEMM, dilated weight gradient) and gh_l (backward-data GEMM, dilated weight gradient) -- reaches THREE consumers, each
// of which split every element it staged (split2: 4 VALU per element, a third of the K loops' instruction stream once
// the MFMAs were halved).  Its producer now writes it split, ONCE: one dword per element at the fp32 element's
// address = fp16 hi | fp16 lo << 16 of x * 2^k, so every consumer keeps its addressing and stages two elements with
// two v_perm_b32 (presplit_stage).  k must be known BEFORE the producer runs, so it comes from a rigorous a-priori
// BOUND on the tensor's absolute maximum instead of the maximum itself:
//     |x_{l+1}| = |x_l + Wr z + br| <= max|x_l| + max_r (sum_c |Wr[r][c]| + |br[r]|)          (|z| = |tanh * sigmoid| <= 1)
//     |gh_l| <= |gz| = |Wr^T g_res + Ws^T g_skip| <= max_c sum_r |Wr[r][c]| * max|g_res| + max_c sum_s |Ws[s][c]| * max|g_skip|
// (the maxima on the right are the ACTUAL ones, published by the producers' epilogues as before; the weight norms are
// found once per step by wl1_kernel).  The producer writes the bound into the tensor's SCALE words -- the group of
// AMAX_SLOTS words its consumers are handed in place of the maximum -- so consumers derive the very k it used.  A bound
// that is 2^m above the true maximum costs m of the 2^-39 absolute precision bits (an element within 2^-(15-m) of
// the maximum still carries a full fp32 significand): see DESIGN.md 3a for the numbers (m = 1-2 for x, 4-6 for gh).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void presplit_pair(float x0, float x1, int k, unsigned& d0, unsigned& d1) {   // the stored dwords of two elements
  unsigned h, l;
  split2(x0, x1, k, h, l);
  d0 = __builtin_amdgcn_perm(l, h, 0x05040100u);        // hi(x0) | lo(x0) << 16
  d1 = __builtin_amdgcn_perm(l, h, 0x07060302u);        // hi(x1) | lo(x1) << 16
}
__device__ __forceinline__ void presplit_stage(float d0, float d1, unsigned& h, unsigned& l) {            // two stored elements -> the fp16 pair of each piece
  h = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, d1), __builtin_bit_cast(unsigned, d0), 0x05040100u);
  l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, d1), __builtin_bit_cast(unsigned, d0), 0x07060302u);
}
__device__ __forceinline__ float presplit_scaled(float d) {       // hi + lo = x * 2^k (exact in fp32: 22 significant bits)
  const f16x2 v = __builtin_bit_cast(f16x2, d);
  return (float)v[0] + (float)v[1];
}
__device__ __forceinline__ float presplit_value(float d, int kinv) { return __builtin_ldexpf(presplit_scaled(d), kinv); }
// a bound enters the scale words with a margin for the roundings of what it bounds (fp32 accumulation over <= 2560
// terms: relative 2^-12 at worst) and never as zero (an all-zero tensor keeps a finite scale)
__device__ __forceinline__ float bound_margin(float b) { return fmaxf(b * 1.001f, 1e-30f); }
__device__ __forceinline__ void scale_publish(unsigned* scale_out, float bound) {     // word 0; the caller zeroed the group
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = __builtin_bit_cast(unsigned, bound);
}

// one 32 x 32 x 16 MFMA on 16-byte fragment words: fp16 (NP == 2) or bf16 operands
template <int NP>
__device__ __forceinline__ f32x16 mfma16(const uint4 a, const uint4 b, const f32x16 c) {
  if constexpr (NP == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// the product chain of one accumulator tile and K step, small products first (NP pieces per operand: 3 -> six bf16
// products, 2 -> three fp16 products, 1 -> one bf16 product)
template <int NP>
__device__ __forceinline__ f32x16 mfma_chain(const uint4 (&a)[NP], const uint4 (&b)[NP], f32x16 c) {
  if constexpr (NP == 3) {
    c = mfma16<3>(a[2], b[0], c);
    c = mfma16<3>(a[0], b[2], c);
    c = mfma16<3>(a[1], b[1], c);
  }
  if constexpr (NP >= 2) {
    c = mfma16<NP>(a[1], b[0], c);
    c = mfma16<NP>(a[0], b[1], c);
  }
  return mfma16<NP>(a[0], b[0], c);
}

// NB = 128-column blocks per workgroup (a.ntile_n counts NB*128-column tiles).  NB = 2 (256 x 256
// tiles, 256-row tiles only): every weight word staged serves twice the columns -- the weight
// stream from L2 is the largest non-MFMA consumer of the power budget the chip runs into (DESIGN.md
// section 8) -- and a barrier covers 48 MFMAs per wave; each wave then owns two 64 x 64 blocks, 128
// columns apart, and runs the unchanged epilogue on each.
// NP = bf16 pieces per operand: 3 (mode 2, six products) or 1 (mode 1: operands rounded to bf16, one product).
// TAP2: the contraction is exactly two segments of the same shape (channel count a multiple of 16, same
// row pitch / extent / stride) -- the two taps of a dilated conv, forward and backward-data, or the
// g_res | g_skip pair of the gate-derivative GEMM: the K loop alternates the two segments channel group
// by channel group instead of running segment 0 to the end first.  For two taps of ONE tensor the second
// fetch of a group's rows then follows the first by one step and is served by L1 / L2 instead of HBM /
// MALL (round 1's gate kernel read x 2.09 times from the fabric); for any pair, both segments advance
// by the same scalar offsets and the loop never re-runs the per-segment setup.
#ifndef X3_ADMA
#define X3_ADMA 1             // the 256 x 128-tile two-tap loop brings its weights into LDS by LDS-DMA (see ADMA in the kernel)
#endif
#ifndef X3_LEAN
#define X3_LEAN 1             // 256 x 128 tiles, two taps, NP >= 2: the 128-VGPR loop below (two 8-wave workgroups per CU)
#endif
// ---------------------------------------------------------------------------
// weight packing: dst[(tap*Rpad + k)*ldw + m_off + mp] = src[k*s_k + m*s_m + tap*s_tap]
// where m = unpermute(mp) (gate interleave) ; zero for k >= R or m >= Cm.
// ---------------------------------------------------------------------------
struct PackJob {
  float* dst; const float* src;
  int R, Cm, K;          // k extent, m extent, taps
  long s_k, s_m, s_tap;  // source strides
  int gate_half;         // 0, or Ch: interleave 32-row groups of [0,Ch) and [Ch,2Ch)
  int Rpad, ldw, m_off;
  int mspan;             // columns of dst this job owns (multiple of 4, zero filled)
  unsigned* amax;        // format 3: where wamax_kernel leaves max |src| (device, float bits); the slab holds src * 2^(14 - e)
};
struct PackArgs { PackJob job[MAXSEG]; int njob; int bf16; };     // bf16 = slab format: 0 fp32, 1 bf16, 2 three bf16 pieces, 3 two scaled fp16 pieces

// ---------------------------------------------------------------------------
// bwd-weight: gW[co, (seg,ci)] = sum_{b,t} gy[b,co,t] * x_seg[b,ci,tin(t)]
// ---------------------------------------------------------------------------
#ifndef WGRAD_WAVES_PER_EU
#define WGRAD_WAVES_PER_EU 2
#endif
constexpr int WBK = 32, WP = WBK + 1;
constexpr int WPB = 40;     // bf16 image of the wgrad tiles: 32 k + 8 pad = 80 B per row

struct WSeg {
  const float* x; long x_bstride; int x_cstride; int cin; int Tin;
  int tmul, toff, tdiv;
  int vec;            // host: 16-B row loads of x are legal for this segment
  const float* gy;    // this segment's own output-gradient tensor (nullptr: WgradArgs.gy)
  float* gw; long gw_co_stride, gw_ci_stride;
  float* gb; float* gb2;   // bias-gradient destinations fed by this segment's gy (nullable)
  int tile0;          // first global n-tile of this segment
  int ptile0;         // first global 256-column tile of this segment (wgrad3_kernel<4, 2>)
  // float32x2 (NP = 2): absolute maxima (device, float bits) of this segment's x and of its own gy (nullptr with
  // gy == nullptr: WgradArgs::amax_gy); amax_x == nullptr: the host-known bound amax_x_static
  const unsigned* amax_x; float amax_x_static; const unsigned* amax_gy;
};
struct WgradArgs {
  const float* gy; long gy_bstride;
  int M, Tout, B;
  WSeg seg[MAXSEG]; int nseg;
  int ntile_m, ntile_n;      // ntile_n = total over segments
  int ntile_p;               // 256-column tiles, total over segments (every segment starts a new one)
  // split-K over the FLATTENED (batch, time) axis in units of WBK-wide K steps: split s owns global
  // steps [s*steps_per_split, (s+1)*steps_per_split); a step never straddles two batch items
  int steps_per_b, steps_per_split, nsplit;
  int avec;                  // host: 16-B row loads of gy are legal
  float* slabs;              // [nsplit][ntile_m][ntile_n][128][128]
  float* bslabs;             // [nsplit][nseg][ntile_m*128]
  float* gbl[MAXSEG]; int ngbl;   // further copies of segment 0's bias grad (shared gy, many layers)
  int accumulate;
  const int32_t* skip_flag;       // see GemmArgs::skip_flag
  int x16;                        // matmul mode 1 only: the x operand of every segment (the z tensors) is stored as bf16
  int g16;                        // matmul mode 1 only: the output-gradient operand (every segment's gy) is stored as bf16
  const unsigned* amax_gy;        // float32x2: absolute maximum of the common gy
  int f16x2;                      // host: run the float32x2 kernel (every segment carries its maxima)
};

struct L1Job { const float* Wr; const float* br; const float* Ws; float* out; };
struct L1Args { L1Job job[MAXSEG]; int Cr, Cs, Ch; };

// ---------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------
static inline int pad16(int v) { return (v + 15) / 16 * 16; }
static inline int pad128(int v) { return (v + 127) / 128 * 128; }
// rows of `ldw` floats one tap's packed slab occupies: the contraction length padded to whole K
// steps, and half as much again in mode 2 (three bf16 pieces = 6 bytes per weight instead of 4)
// (mode 3's two-fp16-piece slabs need only r, but keep mode 2's stride: one workspace layout serves both kinds of launch)
static inline int slab_rows(int c) { const int r = pad16(c); return g_matmul_dtype >= 2 ? r + r / 2 : r; }

static bool seg_vec_ok(const Seg& s) {
  return s.tmul == 1 && s.tdiv == 1 && (s.x_cstride % 4 == 0) && (s.x_bstride % 4 == 0) &&
         (((uintptr_t)s.x) % 16 == 0);
}

struct WgradPlan { int ntile_m, ntile_n, steps_per_b, steps_per_split, nsplit, nseg; size_t slab_floats, bslab_floats; int wide; };
// wide: the splits are chosen for wgrad3_dma_kernel (one 256 x 256-tile workgroup per CU) -- see wgrad_dma_shape

// ---- what the translation units call in one another -------------------------------------------------------------------
// conv_gemm_x3.hip
template <int EPI> int launch_gemm(GemmArgs& g, int tag, hipStream_t st);
int plan_ksplit(int M, int Tout, int B, int nk);
size_t ksplit_partial_floats(int M, int Tout, int B, int nk);
// conv_gemm_fp32.hip: the mode-0 kernel of a launch_gemm<EPI> call (wm: 4 = 256-row tiles, 2 = 128-row); tag != 0: timed by
// the dispatch's own events
template <int EPI> int launch_gemm_fp32(const GemmArgs& g, int wm, unsigned grid, int tag, hipStream_t st);
// wgrad.hip
WgradPlan plan_wgrad(int M, int B, int Tout, const int* cins, int nseg, bool wide = false);
bool wgrad_dma_shape(int M, int Tout, const int* cins, int nseg);     // float32x2 with both operands pre-split: does this shape run on wgrad3_dma_kernel?
int launch_wgrad(WgradArgs& w, const WgradPlan& p, float* ws, int tag, hipStream_t st);

}  // namespace vq

