// conv_gemm.hip -- the MFMA contraction kernels of the hot path (gfx950; fp32 tensors and results).
//
// Every convolution on the path is a (K,1) filter over the time axis of a
// (B, C, T) tensor with T contiguous, i.e. for one batch element
//     Y[co, t] = sum_{tap, ci} W[co, ci, tap] * X[ci, t*stride + tap*dil - pad]
// which is a GEMM  Y[M=co, N=t] = A[M, K=(tap,ci)] * B[K, N]  whose B operand is
// a time-shifted window of X.  Time is the contiguous axis, so the MFMA B
// fragment (lane -> 32 consecutive t) is read straight out of coalesced rows.
//
//   conv_gemm_x3_kernel     fwd and bwd-data of every conv in the default matmul mode 2 (fp32
//                           products as six bf16 MFMA products of an exact three-way operand
//                           split) and, with one piece, in mode 1 (operands rounded to bf16).
//                           The K dimension is a list of "segments" = (input tensor, tap shift,
//                           packed weight slab); 256 x 256, 256 x 128 or 128 x 128 output tiles,
//                           wavefronts own 64 x 64 blocks of v_mfma_f32_32x32x16_bf16 tiles; LDS
//                           double buffer, fetches two K steps ahead, one barrier per step;
//                           XCD-aware tile order.
//   conv_gemm_kernel        the same GEMMs in mode 0 on v_mfma_f32_32x32x2_f32.
//   gemm_epilogue           shared epilogues: linear (+bias, +residual, +=, relu), gated
//                           tanh*sigmoid (ResidualBlock fwd), gate derivative (ResidualBlock bwd).
//   wgrad3_kernel / wgrad2_kernel / wgrad_kernel
//                           bwd-weight (modes 2 and 1 / mode 0 / strided shapes): contraction over
//                           the flattened (b, t) axis, split-K into deterministic partial slabs,
//                           reduced in fixed order by wgrad_reduce_kernel.
//   pack_kernel             re-lays Chainer (Cout,Cin,K,1) weights as the A slabs the GEMMs want
//                           (mode 0: [k][m] fp32; modes 1, 2: pre-split 16-byte fragment words).
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
static int g_matmul_dtype = 2;
static int g_wgrad_impl = 0;      // 0: auto; 1: force the generic wgrad_kernel (tests / A-B timing)

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
// X16 (matmul mode 1): activations that are STORED as bf16 (GemmArgs::z16 / x16) are fetched with 2-byte loads and
// staged without a conversion.  Bit 0: segment 0 of a TAP2 launch / every segment of any other launch; bit 1: the
// second segment of a TAP2 launch (the two may differ: g_res fp32 | g_skip bf16 in the gate-derivative GEMM).
// (matmul mode 3, NP = 2: the same X16 mask marks PRE-SPLIT segments -- fp16 hi | lo dwords at the fp32 addresses, staged
// with two v_perm_b32 per element pair instead of split2; OUT = 1: EPI_GATE_BWD stores gh that way.)
template <int EPI, int WM, int NB, int NP, bool TAP2 = false, int X16 = 0, int OUT = 0>
__global__ __launch_bounds__(128 * WM, (WM == 2 && EPI == EPI_LINEAR) ? 3 : ((WM == 4 && NB == 1 && TAP2 && X3_LEAN) ? 4 : 2)) void conv_gemm_x3_kernel(const GemmArgs a) {
  static_assert(NB == 1 || WM == 4, "256-column tiles exist for 256-row tiles only");
  static_assert(X16 == 0 || NP == 1 || NP == 2, "bf16-stored activations: mode 1; pre-split activations: mode 3");
  static_assert(OUT == 0 || (NP == 2 && EPI == EPI_GATE_BWD), "pre-split output (bit 0) / fused latent pull-back (bit 1): the float32x2 gate-derivative GEMM");
  static_assert(X16 >= 0 && X16 <= (TAP2 ? 3 : 1), "X16: one bit per TAP2 segment, one bit otherwise");
  constexpr bool SEL0 = (X16 & 1) != 0, SEL1 = TAP2 ? (X16 & 2) != 0 : SEL0;
  constexpr bool RAW0 = SEL0 && NP == 1, RAW1 = SEL1 && NP == 1;      // stored as bf16 (2-byte elements, staged as they are)
  constexpr bool PRE0 = SEL0 && NP == 2, PRE1 = SEL1 && NP == 2;      // stored pre-split (4-byte elements, staged by presplit_stage)
  constexpr unsigned ESZ = RAW0 ? 2u : 4u, ESZ1 = RAW1 ? 2u : 4u;     // bytes per activation element (segment 0 / TAP2's segment 1)
  [[maybe_unused]] auto of_seg1 = [](unsigned v) -> unsigned { return ESZ1 == ESZ ? v : (ESZ1 > ESZ ? v << 1 : v >> 1); };   // a byte offset of segment 0 -> the same element of segment 1
  static_assert(NP >= 1 && NP <= 3, "one piece (bf16 operands), two (fp16 hi + lo, scaled) or three (exact bf16 split)");
  constexpr int SCHED = (WM == 4 && NB == 1 && NP == 3) ? 3 : 0;   // MFMA : VALU interleave of the main loop (A/B at configs[1]: 256-row tiles -3 %, 128-row tiles +2 %)
  constexpr int BM = 64 * WM, NT = 128 * WM, BNW = BN * NB;
  constexpr int NQ = NT / BNW;            // staging threads per tile column
  constexpr int CPT = BK / NQ;            // channels per staging thread and K step: 8 or 4
  constexpr bool SPLITK = (EPI == EPI_LINEAR && WM == 2);
  // (the two-piece gate kernel of the 256 x 128-tile two-tap loop keeps a THIRD image: the condition step's operands, staged in
  // the prologue -- see "the condition as a K step"; 73 KB per workgroup, still two per CU)
  constexpr int NBUF = (EPI == EPI_GATE && NB == 1 && NP == 2 && TAP2 && WM == 4 && X3_LEAN) ? 3 : 2;
  __shared__ uint4 As[NBUF][NP][2][BM];
  __shared__ uint4 Bs[NBUF][NP][2][BNW];
  if (a.skip_flag != nullptr && *a.skip_flag != 0) return;
  VQ_STAMP(tp0);

  const int nblk = gridDim.x;
  int logical;
  {
    const int id = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int ntiles_all = a.ntile_m * a.ntile_n * a.B;
  const int ksp = (SPLITK && a.ksplit > 1) ? logical / ntiles_all : 0;
  const int tile_id = (SPLITK && a.ksplit > 1) ? logical % ntiles_all : logical;
  const int mt = tile_id % a.ntile_m;
  const int rest = tile_id / a.ntile_m;
  const int nt = rest % a.ntile_n;
  const int b = rest / a.ntile_n;
  const int m0 = mt * BM, t0 = nt * BNW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2], acc2[2][2];           // acc2: the second column block (NB == 2), 128 columns to the right
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acc2[i][j][r] = 0.f; }

  int nk = 0;
  for (int s = 0; s < a.nseg; ++s) nk += (a.seg[s].cin + BK - 1) / BK;
  int it_beg = 0, it_end = nk;
  if (SPLITK && a.ksplit > 1) {
    it_beg = ksp * a.ksteps_per_split;
    it_end = min(nk, it_beg + a.ksteps_per_split);
  }
  const int nsteps = it_end - it_beg;

  // ---- float32x2 (NP == 2): the launch's common product scale 2^(28 - emax) and each segment's activation scale
  // 2^(14 - emax + e_w) (see split2); all wave-uniform, read once per workgroup / per segment switch
  [[maybe_unused]] int emax = 0;
  // (two-tap launches read their four maxima -- and max |P| -- in straight-line code, once: the loads travel together; a loop
  // over a runtime segment count, and a second read per use, made the prologue a chain of eight dependent L2 round trips)
  [[maybe_unused]] unsigned axb[2] = {0u, 0u}, awb[2] = {0u, 0u}, apb = 0u;
  [[maybe_unused]] bool fold = EPI == EPI_GATE && a.lerp.fold != 0;
  [[maybe_unused]] int kp = 0, kc = 0;
  [[maybe_unused]] int kout = 0;
  [[maybe_unused]] auto seg_kx = [&](int s) -> int {
    if constexpr (TAP2) return 14 - emax + amax_expo(awb[s]);
    else return 14 - emax + amax_expo(amax_load(a.seg[s].wamax));
  };
  // (the 256 x 128-tile two-tap loop calls this BEHIND its first fetches: the maxima are L2 hits, but a round trip of
  // their own in front of the first operand loads was ~3 % of a workgroup's life -- now they travel together)
  constexpr bool SCALES_LATE = (NB == 1 && NP == 2 && TAP2 && WM == 4 && X3_LEAN);
  auto read_scales = [&]() {
  if constexpr (NP == 2 && TAP2) {
    const unsigned x0 = a.seg[0].amax ? amax_load(a.seg[0].amax) : __builtin_bit_cast(unsigned, a.seg[0].amax_static);
    const unsigned w0 = amax_load(a.seg[0].wamax);
    const unsigned x1 = a.seg[1].amax ? amax_load(a.seg[1].amax) : __builtin_bit_cast(unsigned, a.seg[1].amax_static);
    const unsigned w1 = amax_load(a.seg[1].wamax);
    if (EPI == EPI_GATE && a.lerp.fold) apb = amax_load(a.lerp.amax);
    axb[0] = x0; axb[1] = x1; awb[0] = w0; awb[1] = w1;
  }
  if constexpr (NP == 2) {
    int em = -100000;
    if constexpr (TAP2) em = max(amax_expo(awb[0]) + amax_expo(axb[0]), amax_expo(awb[1]) + amax_expo(axb[1]));
    else
    for (int s = 0; s < a.nseg; ++s) {
      const Seg& sg = a.seg[s];
      const int eb = amax_expo(sg.amax ? amax_load(sg.amax) : __builtin_bit_cast(unsigned, sg.amax_static));
      em = max(em, amax_expo(amax_load(sg.wamax)) + eb);
    }
    // the condition step's products P * c, c <= 1, join the scale -- unless the activations are PRE-SPLIT: their scale,
    // hence the launch's, was fixed by their producer (which saw max |P| too: lin128_stream_kernel's floor)
    if (EPI == EPI_GATE && a.lerp.fold && !PRE0) em = max(em, amax_expo(TAP2 ? apb : amax_load(a.lerp.amax)));
    emax = em;
  }
  // the condition as a K step (see behind the two-tap loop): P is scaled by 2^kp, its lerp coefficients by 2^kc, kp + kc = the
  // launch's product scale 28 - emax.  A pre-split x pins emax; should max |P| then need kc > 15 (the coefficients would leave
  // fp16's range: lin128_stream_kernel's floor on x's scale rules it out inside ResidualNet's chain) the epilogue lerps as before.
  if constexpr (NP == 2 && EPI == EPI_GATE) {
    if (fold) {
      const int ep = amax_expo(TAP2 ? apb : amax_load(a.lerp.amax));
      kp = 14 - ep; kc = 14 - emax + ep;
      if (kc > 15) fold = false;
    }
  }
  // pre-split output (OUT): the power of two gh is stored under, from the a-priori bound sum_seg l1[seg] * max|x_seg|
  if constexpr ((OUT & 1) != 0) {
    float bound = 0.f;
    if constexpr (TAP2) bound = a.bound_l1[0] * __builtin_bit_cast(float, axb[0]) + a.bound_l1[1] * __builtin_bit_cast(float, axb[1]);
    else
    for (int s = 0; s < a.nseg; ++s) {
      const Seg& sg = a.seg[s];
      bound += a.bound_l1[s] * __builtin_bit_cast(float, sg.amax ? amax_load(sg.amax) : __builtin_bit_cast(unsigned, sg.amax_static));
    }
    bound = bound_margin(bound);
    kout = 14 - amax_expo(__builtin_bit_cast(unsigned, bound));
    scale_publish(a.scale_out, bound);
  }
  };
  if constexpr (!SCALES_LATE) read_scales();
  [[maybe_unused]] int kcur = 0, k1 = 0;    // scale exponent of the segment the fetch cursor is in (TAP2: of segment 0 / segment 1)

  // ---- staging state of the next step to fetch (advanced once per fetch) ----------------------
  // Every fetch is a buffer load: descriptor (base, extent) in SGPRs, a per-thread 32-bit offset that is
  // fixed for a whole segment, and a wave-uniform SGPR offset that walks the K steps -- the per-step
  // address arithmetic runs on the scalar unit.  Round 2 fetched through per-thread 64-bit pointers with a
  // compare + two selects + a 64-bit add per load and kept a validity mask for the staging: ~60 VALU
  // instructions per wave and step beside 48 MFMAs, and an instruction issued beside the MFMA stream costs
  // matrix-pipe time whichever wave issues it (tools/pp_prof.py).  Out-of-range elements need no mask: a
  // column outside [0, Tin) gets an offset beyond the descriptor's extent, a channel beyond the segment's
  // last one lies beyond it by construction (extent = cin rows), and such loads return 0.
  const int s_n = tid % BNW, s_c = (tid / BNW) * CPT;      // this thread's column and first channel of a step
  const int a_hi = tid / BM, a_m = tid % BM;               // A: 16-byte words (2j + a_hi) * BM + a_m, j = 0..NP-1
  int seg_i = 0, c_n = 0, cin_n = 0, left = nsteps;
  rsrc_t rw = make_rsrc(a.seg[0].w), rx = make_rsrc(a.seg[0].x), rx1 = rx;   // rx1: TAP2, the second segment's tensor
  unsigned va = 0, vb = 0, vb1 = 0;          // per-thread byte offsets: A word, B column (tap 0 / TAP2: tap 1)
  unsigned sw = 0, sx = 0, sw1 = 0;          // wave-uniform byte offsets of the next step (TAP2: sw1 = tap 1's slab)
  unsigned wl2b = 0, wadvb = 0, xcsb = 0, xadvb = 0;
  constexpr unsigned OOB = 0x80000000u;      // beyond any extent: the load returns 0
  auto col_offset = [&](const Seg& sg, const unsigned esz) -> unsigned {       // byte offset of this thread's column in channel s_c of a step
    const int tnum = (t0 + s_n) * sg.tmul + sg.toff;
    bool ok = tnum >= 0;
    int tin = tnum;
    if (sg.tdiv > 1) { ok = ok && (tnum % sg.tdiv == 0); tin = tnum / sg.tdiv; }
    ok = ok && tin < sg.Tin;
    return ok ? esz * (unsigned)(s_c * sg.x_cstride + tin) : OOB;
  };
  auto seg_setup = [&](int s, int skip) {
    const Seg& sg = a.seg[s];
    cin_n = sg.cin; c_n = skip * BK;
    wl2b = 32u * (unsigned)sg.ldw; wadvb = 32u * NP * (unsigned)sg.ldw;          // 2 ldw / 2 NP ldw 16-byte words
    xcsb = ESZ * (unsigned)sg.x_cstride; xadvb = (unsigned)BK * xcsb;
    rw = make_rsrc(sg.w);
    rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(sg.x) + (long)b * sg.x_bstride * ESZ), 0,
                                           (int)(ESZ * (unsigned)sg.cin * (unsigned)sg.x_cstride), 0x00020000);
    va = 16u * (unsigned)(a_hi * sg.ldw + m0 + a_m);
    vb = col_offset(sg, ESZ);
    sw = (unsigned)skip * wadvb; sx = (unsigned)skip * xadvb;
    if constexpr (NP == 2) kcur = seg_kx(s);
  };
  {
    int s = 0, skip = it_beg;
    while (s + 1 < a.nseg) {
      const int steps = (a.seg[s].cin + BK - 1) / BK;
      if (skip < steps) break;
      skip -= steps; ++s;
    }
    seg_i = s;
    seg_setup(s, skip);
    if constexpr (TAP2) {                    // both segments start at channel 0 and advance together
      const Seg& s1 = a.seg[1];
      vb1 = col_offset(s1, ESZ1);
      sw1 = (unsigned)(reinterpret_cast<const char*>(s1.w) - reinterpret_cast<const char*>(a.seg[0].w));
      rx1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(s1.x) + (long)b * s1.x_bstride * ESZ1), 0,
                                              (int)(ESZ1 * (unsigned)s1.cin * (unsigned)s1.x_cstride), 0x00020000);
      if constexpr (NP == 2) k1 = seg_kx(1);
    }
  }
  auto advance2 = [&]() {                    // TAP2: both taps of a channel group have been fetched
    left -= 2;
    const bool more = left > 0;              // nothing further: later fetches re-read this step (never used)
    sw += more ? wadvb : 0u; sx += more ? xadvb : 0u;
  };
  auto advance = [&]() {
    if (--left <= 0) return;                 // nothing further: later fetches re-read this step (never used)
    c_n += BK;
    if (c_n >= cin_n) seg_setup(++seg_i, 0);
    else { sw += wadvb; sx += xadvb; }
  };

  // staging of one K step's activations: split (or round) this thread's CPT channels of its column, one 8- or 16-byte
  // LDS write per piece.  KX: the segment's scale exponent (NP == 2)
  auto stage_b = [&](auto rawc, const float (&bv)[CPT], const int kx, const int buf) {
    constexpr bool RAW = decltype(rawc)::value;                // the elements arrived as bf16 bits (NP == 1) / as pre-split dwords (NP == 2)
    unsigned pc[NP][CPT / 2];                                  // [piece][channel pair]
#pragma unroll
    for (int e = 0; e < CPT; e += 2) {
      const float v0 = bv[e], v1 = bv[e + 1];                  // out-of-range elements arrived as 0
      if constexpr (NP == 3) split3(v0, v1, pc[0][e / 2], pc[1][e / 2], pc[2][e / 2]);
      else if constexpr (NP == 2 && RAW) presplit_stage(v0, v1, pc[0][e / 2], pc[1][e / 2]);
      else if constexpr (NP == 2) split2(v0, v1, kx, pc[0][e / 2], pc[1][e / 2]);
      else if constexpr (RAW) pc[0][e / 2] = __builtin_bit_cast(unsigned, v0) | (__builtin_bit_cast(unsigned, v1) << 16);   // already bf16
      else pc[0][e / 2] = pack_bf16x2(v0, v1);
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if constexpr (CPT == 8) {
        Bs[buf][p][tid / BNW][s_n] = make_uint4(pc[p][0], pc[p][1], pc[p][2], pc[p][3]);
      } else {
        uint2* bd = reinterpret_cast<uint2*>(&Bs[buf][p][tid >> 8][s_n]) + ((tid >> 7) & 1);
        *bd = make_uint2(pc[p][0], pc[p][1]);
      }
    }
  };

  // two register sets (P: even steps, Q: odd steps) so that the fetch of step i+2 is in flight while
  // step i+1 is split and stored: every wait in the loop is then a counted vmcnt.  The fetches are
  // unconditional (a branch around them makes hipcc drain to vmcnt(0)).
  [[maybe_unused]] uint4 pa0, pa1, pa2, qa0, qa1, qa2;      // scalars, not arrays: hipcc leaves uint4[NP] in scratch / LDS here
  [[maybe_unused]] int pkx = 0, qkx = 0;                    // the scale exponent that goes with each set's activations
#define X3_FETCH(A0, A1, A2, BV, KX) X3_FETCH_(A0, A1, A2, BV, KX, kcur, sw, vb, rx, !TAP2, ((EPI == EPI_GATE_BWD && TAP2) ? X3_GBWD_B0_AUX : (a.x_nt ? 2 : 0)), RAW0, sx, xcsb)
#define X3_FETCH1(A0, A1, A2, BV, KX) X3_FETCH_(A0, A1, A2, BV, KX, k1, sw + sw1, vb1, rx1, false, 0, RAW1, of_seg1(sx), of_seg1(xcsb))
#define X3_FETCH_(A0, A1, A2, BV, KX, KV, SW, VB, RX, ADV, BAUX, RAW, SX, XCS)               \
  {                                                                                          \
    A0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, va, (SW), 0));  \
    if constexpr (NP >= 2) A1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, va, (SW) + wl2b, 0)); \
    if constexpr (NP == 3) A2 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, va, (SW) + 2u * wl2b, 0)); \
    if constexpr (RAW) {                      /* raw bf16 bits, kept in the low half of a register */ \
      _Pragma("unroll") for (int e = 0; e < CPT; ++e)                                          \
        BV[e] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(RX, (VB), (SX) + (unsigned)e * (XCS), 0)); \
    } else                                                                                     \
    _Pragma("unroll") for (int e = 0; e < CPT; ++e) BV[e] = ((BAUX) == 2 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(RX, (VB), (SX) + (unsigned)e * (XCS), 2)) : buf_ld(RX, (VB), (SX) + (unsigned)e * (XCS))); \
    KX = (KV);                                                                               \
    if (!SCHED && (ADV)) advance();                                                          \
  }
#define X3_STAGE(A0, A1, A2, BV, KX, BUF, RAW)                                               \
  {                                                                                          \
    uint4* ad = &As[BUF][0][0][0];                                                           \
    ad[tid] = A0;                                                                            \
    if constexpr (NP >= 2) ad[NT + tid] = A1;                                                \
    if constexpr (NP == 3) ad[2 * NT + tid] = A2;                                            \
    stage_b(std::integral_constant<bool, (RAW)>{}, BV, KX, BUF);                             \
  }
  auto mma = [&](auto curc) {
    constexpr int cur = decltype(curc)::value;
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
    if constexpr (NB == 2) {
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
    if (SCHED) {
      // one MFMA (32 pipe cycles), then a few of the step's other instructions (the split of the next
      // step, the addresses of the one after): hipcc otherwise issues 16 of the 24 MFMAs back to back
      // behind the barrier and everything else after them, with the matrix pipe idle
#pragma unroll
      for (int q = 0; q < 24; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, SCHED, 0);
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // LEAN (256 x 128 tiles, two taps, NP >= 2): the loop in 128 VGPRs, so that TWO 8-wave workgroups share a
  // CU and one tile's epilogue -- 13 % of the gate kernel's time with nothing beside it
  // (profiles/r3/abl_gate_epilogue.txt) -- runs beside the other's K loop.  What it gives up against the loop below: the
  // weights (L2-resident) are fetched ONE step ahead into a single register set, only the activations two;
  // the A fragments of one 32-row block at a time.
  constexpr bool LEAN = (NB == 1 && NP >= 2 && TAP2 && WM == 4 && X3_LEAN);
  VQ_STAMP(tp1);
  if constexpr (LEAN) {
    unsigned swA = 0, sxB = 0;                 // the two cursors: weights of the next A fetch, activations of the next B fetch
    int leftA = nsteps, leftB = nsteps;
    [[maybe_unused]] uint4 la0, la1, la2;
    float pb[CPT], qb[CPT];
    // ADMA: the weights -- already in the LDS image's order in their packed slab, a linear copy -- travel global -> LDS by
    // LDS-DMA (buffer_load_dwordx4 ... lds: one 1 KB run per wave and piece) instead of through 8 VGPRs and two ds_write_b128.
    // Issued from inline asm (hipcc would otherwise drain vmcnt(0) in front of every ds_read of the image); it is the FIRST
    // VMEM operation of its half step, so `vmcnt(CPT)` behind the half step's CPT activation loads retires it before the barrier
    // that publishes the image (the counter is in-order; hipcc's own counted waits, which do not know of it, only wait longer).
    constexpr bool ADMA = X3_ADMA != 0;
    [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    [[maybe_unused]] i32x4_t rw4;
    if constexpr (ADMA) rw4 = make_rsrc4(a.seg[0].w);
#define LN_FETCH_A(TAP1, BUF)                                                                 \
    {                                                                                         \
      const unsigned so_ = swA + ((TAP1) ? sw1 : 0u);                                         \
      if constexpr (ADMA) {                                                                   \
        _Pragma("unroll") for (int p = 0; p < NP; ++p)                                        \
          lds_dma16(lds_addr32(&As[BUF][p][0][0] + 64 * wave_u), va, rw4, so_ + (unsigned)p * wl2b); \
      } else {                                                                                \
      la0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, va, so_, 0)); \
      la1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, va, so_ + wl2b, 0)); \
      if constexpr (NP == 3) la2 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, va, so_ + 2u * wl2b, 0)); \
      }                                                                                       \
      if (TAP1) { leftA -= 2; swA += leftA > 0 ? wadvb : 0u; }                                \
    }
#define LN_FETCH_B(BV, TAP1)                                                                  \
    {                                                                                         \
      _Pragma("unroll") for (int e = 0; e < CPT; ++e)                                         \
        BV[e] = (TAP1) ? buf_ld(rx1, vb1, sxB + (unsigned)e * xcsb) : buf_ld(rx, vb, sxB + (unsigned)e * xcsb); \
      if (TAP1) { leftB -= 2; sxB += leftB > 0 ? xadvb : 0u; }                                \
    }
#define LN_STAGE(BV, KX, BUF, PRE)                                                            \
    {                                                                                         \
      if constexpr (!ADMA) {                                                                  \
      uint4* ad = &As[BUF][0][0][0];                                                          \
      ad[tid] = la0; ad[NT + tid] = la1;                                                      \
      if constexpr (NP == 3) ad[2 * NT + tid] = la2;                                          \
      }                                                                                       \
      stage_b(std::integral_constant<bool, (PRE)>{}, BV, KX, BUF);                            \
      if constexpr (ADMA) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CPT) : "memory");         \
    }
    auto lmma = [&](auto curc) {
      constexpr int cur = decltype(curc)::value;
      uint4 bf[2][NP];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < NP; ++p) bf[j][p] = Bs[cur][p][lk][wn * 64 + j * 32 + li];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint4 af[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) af[p] = As[cur][p][lk][wm * 64 + i * 32 + li];
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_chain<NP>(af, bf[j], acc[i][j]);     // same product order as the loop below
      }
    };
    // the condition step's operand images (see "the condition as a K step" behind the loop)
    [[maybe_unused]] auto stage_cond = [&](auto bufc) {
      constexpr int cbuf = decltype(bufc)::value;
        const int vb = a.lerp.v0[t0];
        // A: row a_m of the tile, k slots 0..7 = P[ch][vb .. vb + 7] (slot 7 never has a coefficient), slots 8..15 = 0
        uint4 wa[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) wa[p] = make_uint4(0u, 0u, 0u, 0u);
        if (a_hi == 0) {
          const int m = m0 + a_m, Chh = a.M >> 1;
          const int ch = ((m >> 5) & 1) * Chh + 32 * (m >> 6) + (m & 31);
          const float* pr = a.lerp.P + (long)b * a.lerp.p_bstride + (long)ch * a.lerp.Tl;
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pv[j] = pr[min(vb + j, a.lerp.Tl - 1)];
          unsigned pc[NP][4];
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            if constexpr (NP == 3) split3(pv[e], pv[e + 1], pc[0][e / 2], pc[1][e / 2], pc[2][e / 2]);
            else split2(pv[e], pv[e + 1], kp, pc[0][e / 2], pc[1][e / 2]);
          }
#pragma unroll
          for (int p = 0; p < NP; ++p) wa[p] = make_uint4(pc[p][0], pc[p][1], pc[p][2], pc[p][3]);
        }
        // B: column s_n, k slots 4 (tid / 128) .. + 3
        float cv[CPT];
        {
          const int t = min(t0 + s_n, a.Tout - 1);
          const int dv = a.lerp.v0[t] - vb;
          const float c0 = a.lerp.w0[t], c1 = a.lerp.w1[t];
#pragma unroll
          for (int e = 0; e < CPT; ++e) {
            const int j = s_c + e;
            cv[e] = j == dv ? c0 : (j == dv + 1 ? c1 : 0.f);
          }
        }
        uint4* ad = &As[cbuf][0][0][0];
#pragma unroll
        for (int p = 0; p < NP; ++p) ad[p * NT + tid] = wa[p];
        stage_b(std::false_type{}, cv, kc, cbuf);
    };
    if constexpr (SCALES_LATE) { if (nsteps <= 0) read_scales(); }      // (never: every launch of this loop has K steps)
    if (nsteps > 0) {
      LN_FETCH_B(pb, false);                   // step 0
      LN_FETCH_A(false, 0);                    // step 0
      LN_FETCH_B(qb, true);                    // step 1
      if constexpr (SCALES_LATE) { read_scales(); kcur = seg_kx(0); k1 = seg_kx(1); }
      if constexpr (EPI == EPI_GATE && NBUF == 3) {
        if (fold) stage_cond(std::integral_constant<int, 2>{});     // its loads travel with the first steps'; read after the loop: the loop's barriers order the writes
      }
      LN_STAGE(pb, kcur, 0, PRE0);
      __syncthreads();
      for (int i = 0; i < nsteps; i += 2) {    // nsteps is even: two taps per channel group
        LN_FETCH_A(true, 1);                   // weights of step i + 1
        LN_FETCH_B(pb, false);                 // activations of step i + 2
        lmma(std::integral_constant<int, 0>{});
        LN_STAGE(qb, k1, 1, PRE1);             // step i + 1
        __syncthreads();
        LN_FETCH_A(false, 0);                  // weights of step i + 2
        LN_FETCH_B(qb, true);                  // activations of step i + 3
        lmma(std::integral_constant<int, 1>{});
        LN_STAGE(pb, kcur, 0, PRE0);           // step i + 2
        __syncthreads();
      }
    }
#undef LN_FETCH_A
#undef LN_FETCH_B
#undef LN_STAGE
    // ---- the condition as a K step.  h += upsample(P)[t] = w0[t] P[v0[t]] + w1[t] P[v0[t] + 1] (net.py:54-55 after the
    // latent-rate projection, align-corners lerp) is itself a small matrix product: the 128 columns of a tile touch at
    // most 7 consecutive latent positions vb .. vb + 6 (the host guarantees Tout >= 26 Tl), so
    //     cond[m, t] = sum_{j < 8} P[ch(m), vb + j] * c_j[t],   c_j[t] = w0[t] (j = v0[t] - vb), w1[t] (j = v0[t] - vb + 1), 0 otherwise
    // is ONE more step of this very contraction (12 MFMAs per wave, +3 %).  The epilogue used to fetch 4 values of P per
    // output element through 128 dependent L2 loads per lane: 40 k of the gate workgroup's 130 k cycles (measured with
    // s_memtime stamps, round 5) -- now the epilogue starts with finished pre-activations.  P carries both biases
    // (vqvae_resblock_cproj::P_has_bd).  float32x2: P is scaled by 2^(14 - e_P), c by 2^(14 - emax + e_P) <= 2^14 (emax
    // includes e_P, see above): same product scale as every other step.
    VQ_STAMP(tpc);
    VQ_PHASE_ADD(EPI, 1, tpc - tp1);
    if constexpr (EPI == EPI_GATE) {
      if (fold) {
        if constexpr (NBUF == 3) lmma(std::integral_constant<int, 2>{});        // staged in the prologue (stage_cond), ordered by the loop's barriers
        else {
          stage_cond(std::integral_constant<int, 1>{});      // every wave has finished with image 1 (the loop's last barrier)
          __syncthreads();
          lmma(std::integral_constant<int, 1>{});
        }
      }
    }
    VQ_STAMP(tpd);
    VQ_PHASE_ADD(EPI, 2, tpd - tpc);
  } else
  if (nsteps > 0) {
    float pb[CPT], qb[CPT];
    X3_FETCH(pa0, pa1, pa2, pb, pkx);
    if (SCHED && !TAP2) advance();
    if constexpr (TAP2) { X3_FETCH1(qa0, qa1, qa2, qb, qkx); advance2(); }
    else X3_FETCH(qa0, qa1, qa2, qb, qkx);
    if (SCHED && !TAP2) advance();
    X3_STAGE(pa0, pa1, pa2, pb, pkx, 0, SEL0);
    __syncthreads();
    // top of a pair (i even): LDS buffer 0 holds step i, set Q holds (in flight) step i + 1.  (Whole pairs in the loop,
    // an odd last step behind it: with a `break` between the halves hipcc copied the accumulators between register sets
    // inside the loop and spilled 250 registers in the two-piece instantiations.)
    for (int i = 0; i + 1 < nsteps; i += 2) {
      X3_FETCH(pa0, pa1, pa2, pb, pkx);             // step i + 2 (past the end: re-reads the last step, never used)
      mma(I0{});
      X3_STAGE(qa0, qa1, qa2, qb, qkx, 1, SEL1);    // step i + 1 (TAP2: the second segment's set)
      if (SCHED && !TAP2) advance();
      __syncthreads();
      if constexpr (TAP2) { X3_FETCH1(qa0, qa1, qa2, qb, qkx); advance2(); }
      else X3_FETCH(qa0, qa1, qa2, qb, qkx);        // step i + 3
      mma(I1{});
      X3_STAGE(pa0, pa1, pa2, pb, pkx, 0, SEL0);    // step i + 2
      if (SCHED && !TAP2) advance();
      __syncthreads();
    }
    if (nsteps & 1) mma(I0{});                      // the last step of an odd count: staged by the prologue / the last pair
  }
#undef X3_FETCH
#undef X3_FETCH1
#undef X3_FETCH_
#undef X3_STAGE
  [[maybe_unused]] auto unscale = [&](f32x16 (&ac)[2][2]) {      // float32x2: back from the launch's product scale 2^(28 - emax)
    const int ku = emax - 28;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ac[i][j][r] = __builtin_ldexpf(ac[i][j][r], ku);
  };
  VQ_STAMP(tp2);
  if constexpr (NP == 2) unscale(acc);
  gemm_epilogue<EPI, WM, SPLITK, ((WM == 4 && !(NB == 1 && TAP2 && X3_LEAN)) || EPI == EPI_GATE_BWD), NP == 1, OUT>(a, acc, m0, t0, b, wm, wn, li, lk, ksp, tile_id, ntiles_all, kout, fold && NB == 1 && NP >= 2 && TAP2 && WM == 4 && X3_LEAN);   // two workgroups per CU: no room for the deep epilogue's 64 registers, and no need
#ifdef VQ_PHASE_TIMING
  if (NP == 2 && TAP2 && NB == 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VQ_STAMP(tp3);
    VQ_PHASE_ADD(EPI, 0, tp1 - tp0);
    if (!LEAN) VQ_PHASE_ADD(EPI, 1, tp2 - tp1);
    VQ_PHASE_ADD(EPI, 3, tp3 - tp2);
    VQ_PHASE_ADD(EPI, 4, 1);
  }
#endif
  if constexpr (NB == 2) {
    if constexpr (NP == 2) unscale(acc2);
    if (t0 + BN < a.Tout) gemm_epilogue<EPI, WM, false, true, NP == 1>(a, acc2, m0, t0 + BN, b, wm, wn, li, lk, 0, tile_id, ntiles_all);
  }
}

// ---------------------------------------------------------------------------
// lin128_stream_kernel -- the K = 128 -> M = 256 1x1 projection with residual add (ResidualBlock's
// `res` conv, modules.py:50-52; 19 launches per configs[1] step) as a STREAMING kernel.
//
// The shape is HBM-bound (315 MB per launch against 8 GFLOP), and the tiled GEMM kernel ran it at
// 2.4 TB/s: eight K steps are too short a loop to reach a steady state, and with one 98 KB workgroup
// per CU nothing overlapped a tile's 0.5 MB read-modify-write epilogue.  Here:
//   * one persistent 8-wave workgroup per CU walks 256 x NC column tiles;
//   * the WHOLE weight matrix lives in registers for the life of the workgroup: wave w owns rows
//     32w..32w+31, i.e. the A fragments of all 8 K steps (8 x NP 16-byte words per lane, read once from
//     the packed slab) -- no weight traffic, no A staging, no A LDS reads per tile;
//   * memory-level parallelism comes from registers, a whole tile ahead: the z tile (128 x NC fp32) of
//     tile i+2 is in flight while tile i+1 is multiplied; tile i+1 is split and written to the other LDS
//     buffer right behind tile i's MFMAs; the residual operands of tile i+1 are requested before tile
//     i's MFMAs; tile i's stores drain behind tile i+1's MFMAs.  One barrier per tile, every wait a
//     counted vmcnt.
// Products, K order and the epilogue's (acc + bias) + x are those of conv_gemm_x3_kernel, so the
// result is the same to the last bit whichever kernel the launch picks.
// ---------------------------------------------------------------------------
struct Lin128Args {
  const uint4* w; int ldw;                 // packed slab (pack_kernel, modes 1 / 2), one tap, K = 128
  const float* z; long z_bstride;          // (B, 128, T)
  const float* add; long add_bstride;      // (B, 256, T) residual, HAS_ADD only
  float* y; long y_bstride;                // (B, 256, T)
  const float* bias;                       // 256 or null
  int T, tiles_per_b, ntiles;
  // float32x2 (NP = 2): maxima of the weights (as packed) and of z (device pointer or host-known bound); amax_out
  // (nullable, any mode): atomicMax of |y| over the launch
  const unsigned* wamax; const unsigned* z_amax; float z_amax_static; unsigned* amax_out;
  int add16, y16;                          // template flags' runtime twins (host side only)
  // float32x2, PRE-SPLIT residual stream (presplit_pair): ADD16 -- `add` holds x_l as hi | lo dwords split under the
  // bound in its scale words add_scale; Y16 -- y = x_{l+1} is stored that way under the bound max|x_l| + *l1
  // (add_amax: the ACTUAL max |x_l|; l1: max_r (sum_c |Wr[r][c]| + |br[r]|), wl1_kernel), published to scale_out
  const unsigned* add_scale; const unsigned* add_amax; const float* l1; unsigned* scale_out;
  const unsigned* floor_w; const unsigned* floor_p;      // see GemmArgs
};

// Z16 (matmul mode 1): z is stored as bf16 (same element strides): fetched as 2 x CPC bytes per row and staged as is.
// ADD16 / Y16 (matmul mode 1): the residual stream is kept as bf16 -- x_l read, x_{l+1} = bf16((acc + bias) + x_l) stored
// with 2-byte accesses at the same element strides (the first block of a stack reads an fp32 x: ADD16 off, Y16 on).
template <int NP, int NC, bool HAS_ADD, bool Z16 = false, bool ADD16 = false, bool Y16 = false>
__global__ __launch_bounds__(512, 1) void lin128_stream_kernel(const Lin128Args a) {
  static_assert(!Z16 || NP == 1, "bf16-stored z: mode 1 only");
  static_assert((!ADD16 && !Y16) || ((NP == 1 || NP == 2) && HAS_ADD), "bf16 (mode 1) / pre-split (mode 3) residual stream: with the residual add");
  constexpr bool ADDPRE = ADD16 && NP == 2, YPRE = Y16 && NP == 2;     // pre-split x_l in / x_{l+1} out: 4-byte elements at the fp32 addresses
  constexpr bool ADDB16 = ADD16 && NP == 1, YB16 = Y16 && NP == 1;     // the bf16 stream of mode 1: 2-byte elements
  constexpr int KS = 8, NCB = NC / 32;
  constexpr int CPC = NC / 16;                     // columns per staging thread: 16 column groups x 32 channel quads = 512 threads
  constexpr int STEPW = NP * 2 * NC;               // 16-byte words per K step of the B image
  __shared__ uint4 Bs[2][KS * STEPW];              // [buf][s][piece][k-half][col]
  __shared__ float4 bias_s[64];                    // 256 biases
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lk = lane >> 5;
  const int T = a.T;

  // ---- the weights: this wave's 32 rows x 128 k, as MFMA A fragments, for the whole launch ----
  uint4 af[KS][NP];
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int p = 0; p < NP; ++p) af[s][p] = a.w[((long)(s * NP + p) * 2 + lk) * a.ldw + 32 * wave + li];
  if (tid < 256) reinterpret_cast<float*>(bias_s)[tid] = a.bias ? a.bias[tid] : 0.f;

  // ---- staging role: CPC consecutive columns x 4 consecutive channels per thread ----
  const int cg = tid & 15, kq = tid >> 4;
  const int k0 = 4 * kq;
  // word (s, piece, k-half, col) holds k = 16 s + 8 half .. + 7; this thread fills 8-byte half `sub` of it
  const int st_word = ((k0 >> 4) * NP * 2 + ((k0 >> 3) & 1)) * NC + CPC * cg;
  const int st_sub = (k0 >> 2) & 1;
  float zr[4][CPC];
  const int last = a.ntiles - 1;
#define L128_FETCH(TILE)                                                                       \
  {                                                                                            \
    const int tl_ = min((TILE), last);          /* past the end: re-read the last tile, unused */ \
    const int b_ = tl_ / a.tiles_per_b, t_ = (tl_ - b_ * a.tiles_per_b) * NC;                  \
    const float* p_ = a.z + (long)b_ * a.z_bstride + (long)k0 * T + t_ + CPC * cg;             \
    const unsigned short* h_ = reinterpret_cast<const unsigned short*>(a.z) + (long)b_ * a.z_bstride + (long)k0 * T + t_ + CPC * cg; \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                            \
      if constexpr (Z16) {                        /* raw bf16 bits in the low half of a register */ \
        if constexpr (CPC == 4) {                                                              \
          const uint2 v_ = L128_Z_NT ? __builtin_bit_cast(uint2, __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(h_ + (long)j * T))) : *reinterpret_cast<const uint2*>(h_ + (long)j * T); \
          zr[j][0] = __builtin_bit_cast(float, v_.x & 0xffffu); zr[j][1] = __builtin_bit_cast(float, v_.x >> 16); \
          zr[j][2] = __builtin_bit_cast(float, v_.y & 0xffffu); zr[j][3] = __builtin_bit_cast(float, v_.y >> 16); \
        } else {                                                                               \
          const unsigned v_ = L128_Z_NT ? __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(h_ + (long)j * T)) : *reinterpret_cast<const unsigned*>(h_ + (long)j * T); \
          zr[j][0] = __builtin_bit_cast(float, v_ & 0xffffu); zr[j][1] = __builtin_bit_cast(float, v_ >> 16); \
        }                                                                                      \
      } else                                                                                   \
      if constexpr (CPC == 4) {                                                                \
        const float4 v_ = L128_Z_NT ? __builtin_bit_cast(float4, __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p_ + (long)j * T))) : *reinterpret_cast<const float4*>(p_ + (long)j * T); \
        zr[j][0] = v_.x; zr[j][1] = v_.y; zr[j][2] = v_.z; zr[j][3] = v_.w;                    \
      } else {                                                                                 \
        const float2 v_ = L128_Z_NT ? __builtin_bit_cast(float2, __builtin_nontemporal_load(reinterpret_cast<const f32x2_t*>(p_ + (long)j * T))) : *reinterpret_cast<const float2*>(p_ + (long)j * T); \
        zr[j][0] = v_.x; zr[j][1] = v_.y;                                                      \
      }                                                                                        \
    }                                                                                          \
  }
#define L128_STAGE(BUF)                                                                        \
  {                                                                                            \
    uint4* base_ = &Bs[BUF][st_word];                                                          \
    _Pragma("unroll") for (int c = 0; c < CPC; ++c) {                                          \
      uint2* d_ = reinterpret_cast<uint2*>(base_ + c) + st_sub;                                \
      if constexpr (NP == 2) {                                                                 \
        unsigned h0, l0, h1, l1;                                                               \
        split2(zr[0][c], zr[1][c], kz, h0, l0);                                                \
        split2(zr[2][c], zr[3][c], kz, h1, l1);                                                \
        d_[0] = make_uint2(h0, h1);                                                            \
        d_[2 * (2 * NC)] = make_uint2(l0, l1);                                                 \
      } else if constexpr (NP == 3) {                                                          \
        unsigned h0, m0, l0, h1, m1, l1;                                                       \
        split3(zr[0][c], zr[1][c], h0, m0, l0);                                                \
        split3(zr[2][c], zr[3][c], h1, m1, l1);                                                \
        d_[0] = make_uint2(h0, h1);                                                            \
        d_[2 * (2 * NC)] = make_uint2(m0, m1);                                                 \
        d_[2 * (4 * NC)] = make_uint2(l0, l1);                                                 \
      } else if constexpr (Z16) {                                                              \
        d_[0] = make_uint2(__builtin_bit_cast(unsigned, zr[0][c]) | (__builtin_bit_cast(unsigned, zr[1][c]) << 16), \
                           __builtin_bit_cast(unsigned, zr[2][c]) | (__builtin_bit_cast(unsigned, zr[3][c]) << 16)); \
      } else {                                                                                 \
        d_[0] = make_uint2(pack_bf16x2(zr[0][c], zr[1][c]), pack_bf16x2(zr[2][c], zr[3][c]));  \
      }                                                                                        \
    }                                                                                          \
  }
  // residual operands of a tile, in the accumulator layout (requested a whole tile ahead)
  const unsigned voff = 4u * (unsigned)(4 * lk * T + li);
  // bf16 residual stream (ADD16 / Y16): 2-byte accesses would move 128 bytes per wave instruction (measured: +23 us per
  // launch).  A lane PAIR (columns t, t + 1) shares the dwords of a ROW pair (rows R, R + 1) instead: the even lane
  // owns (R, t .. t + 1), the odd lane (R + 1, t .. t + 1); what the other lane needs / produces travels by one DPP
  // swap.  voff16: this lane's dword of the row pair that starts at the descriptor offset of row R.
  [[maybe_unused]] const unsigned voff16 = 2u * (unsigned)((4 * lk + (li & 1)) * T + (li & ~1));
#define L128_XLOAD(XV, TILE)                                                                   \
  if constexpr (HAS_ADD) {                                                                     \
    const int tl_ = min((TILE), last);                                                         \
    const int b_ = tl_ / a.tiles_per_b, t_ = (tl_ - b_ * a.tiles_per_b) * NC;                  \
    const rsrc_t rx_ = make_rsrc(reinterpret_cast<const char*>(a.add) + (long)b_ * a.add_bstride * (ADDB16 ? 2 : 4)); \
    const unsigned sb_ = 4u * (unsigned)(32 * wave * T + t_);                                  \
    _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb)                                         \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                         \
        const unsigned so_ = sb_ + 4u * (unsigned)(cb * 32 + ((r & 3) + 8 * (r >> 2)) * T);    \
        if constexpr (ADDB16) {                   /* one dword per ROW PAIR (see voff16): entries r = 4q, 4q + 2 only */ \
          if ((r & 1) == 0) XV[cb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx_, voff16, so_ >> 1, L128_X_AUX)); \
        } else XV[cb][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx_, voff, so_, L128_X_AUX)); \
      }                                                                                        \
  }
  // one tile: MFMAs on LDS buffer CUR, next tile's z -> the other buffer, epilogue with XCUR while
  // XNXT (the next tile's residual) and the z tile after the next travel
#define L128_TILE(CUR, XCUR, XNXT)                                                             \
  {                                                                                            \
    const int b = tile / a.tiles_per_b, t0 = (tile - b * a.tiles_per_b) * NC;                  \
    L128_XLOAD(XNXT, tile + stride);                                                           \
    f32x16 acc[NCB];                                                                           \
    _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb)                                         \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;                         \
    const uint4* bb = &Bs[CUR][lk * NC + li];                                                  \
    _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                           \
      _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb) {                                     \
        uint4 bq[NP];                                                                          \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) bq[p] = bb[s * STEPW + p * 2 * NC + cb * 32]; \
        acc[cb] = mfma_chain<NP>(af[s], bq, acc[cb]);             /* small products first (as conv_gemm_x3_kernel) */ \
      }                                                                                        \
    }                                                                                          \
    L128_STAGE(CUR ^ 1);                                                                       \
    {                                                                                          \
      const rsrc_t ry = make_rsrc(reinterpret_cast<char*>(a.y) + (long)b * a.y_bstride * (YB16 ? 2 : 4)); \
      const unsigned sbase = 4u * (unsigned)(32 * wave * T + t0);                              \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                          \
        const float4 bq4 = bias_s[8 * wave + 2 * q + lk];         /* rows 32w + 8q + 4lk .. + 3 */ \
        const float bv[4] = {bq4.x, bq4.y, bq4.z, bq4.w};                                      \
        _Pragma("unroll") for (int cb = 0; cb < NCB; ++cb) {                                   \
          if constexpr (ADDPRE || YPRE) {             /* float32x2, pre-split stream: the fp32 path's addresses, dwords re-coded */ \
            _Pragma("unroll") for (int j = 0; j < 4; j += 2) {                                 \
              const int r = 4 * q + j;                                                         \
              float va = __builtin_ldexpf(acc[cb][r], ku) + bv[j], vb = __builtin_ldexpf(acc[cb][r + 1], ku) + bv[j + 1]; \
              if constexpr (ADDPRE) { va += presplit_value(XCUR[cb][r], kadd); vb += presplit_value(XCUR[cb][r + 1], kadd); } \
              else { va += XCUR[cb][r]; vb += XCUR[cb][r + 1]; }                               \
              am = fmaxf(am, fmaxf(fabsf(va), fabsf(vb)));                                     \
              const unsigned so_ = sbase + 4u * (unsigned)(cb * 32 + (j + 8 * q) * T);         \
              if constexpr (YPRE) {                                                            \
                unsigned da_, db_;                                                             \
                presplit_pair(va, vb, kout, da_, db_);                                         \
                __builtin_amdgcn_raw_buffer_store_b32((int)da_, ry, voff, so_, L128_ST_AUX);   \
                __builtin_amdgcn_raw_buffer_store_b32((int)db_, ry, voff, so_ + 4u * (unsigned)T, L128_ST_AUX); \
              } else {                                                                         \
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, va), ry, voff, so_, L128_ST_AUX); \
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vb), ry, voff, so_ + 4u * (unsigned)T, L128_ST_AUX); \
              }                                                                                \
            }                                                                                  \
          } else                                                                               \
          if constexpr (ADD16 || Y16) {                                                        \
            _Pragma("unroll") for (int j = 0; j < 4; j += 2) {         /* rows R = 32w + 8q + 4lk + j and R + 1 */ \
              const int r = 4 * q + j;                                                         \
              float va = acc[cb][r] + bv[j], vb = acc[cb][r + 1] + bv[j + 1];                  \
              if constexpr (ADD16) {                                                           \
                const unsigned own = __builtin_bit_cast(unsigned, XCUR[cb][r]);                \
                const unsigned got = (unsigned)__shfl_xor((int)own, 1);                        \
                /* even lane: own = row R cols (t, t + 1), got = row R + 1 cols (t, t + 1); odd lane (col t + 1): the other way round */ \
                const unsigned ra_ = (li & 1) ? got : own, rb_ = (li & 1) ? own : got;         \
                va += __builtin_bit_cast(float, (li & 1) ? (ra_ & 0xffff0000u) : (ra_ << 16)); \
                vb += __builtin_bit_cast(float, (li & 1) ? (rb_ & 0xffff0000u) : (rb_ << 16)); \
              } else { va += XCUR[cb][r]; vb += XCUR[cb][r + 1]; }                             \
              const unsigned so_ = sbase + 4u * (unsigned)(cb * 32 + (j + 8 * q) * T);         \
              if constexpr (Y16) {                                                             \
                const unsigned h = pack_bf16x2(va, vb);                    /* lo: row R, hi: row R + 1 (this lane's column) */ \
                const unsigned send = (li & 1) ? (h & 0xffffu) : (h >> 16); /* what the OTHER lane stores: its row, this column */ \
                const unsigned got = (unsigned)__shfl_xor((int)send, 1);                       \
                const unsigned pr = (li & 1) ? (got | (h & 0xffff0000u)) : ((h & 0xffffu) | (got << 16)); \
                __builtin_amdgcn_raw_buffer_store_b32((int)pr, ry, voff16, so_ >> 1, L128_ST_AUX); \
              } else {                                                                         \
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, va), ry, voff, so_, L128_ST_AUX); \
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vb), ry, voff, so_ + 4u * (unsigned)T, L128_ST_AUX); \
              }                                                                                \
            }                                                                                  \
          } else {                                                                             \
          _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                      \
            const int r = 4 * q + j;                                                           \
            float v = (NP == 2 ? __builtin_ldexpf(acc[cb][r], ku) : acc[cb][r]) + bv[j];       \
            if constexpr (HAS_ADD) v += XCUR[cb][r];                                           \
            am = fmaxf(am, fabsf(v));                                                          \
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (float)(v)), ry, voff, sbase + 4u * (unsigned)(cb * 32 + (j + 8 * q) * T), L128_ST_AUX);           \
          }                                                                                    \
          }                                                                                    \
        }                                                                                      \
      }                                                                                        \
    }                                                                                          \
    /* the tile after the next goes in flight behind this tile's stores (VMEM retires in order: the  \
       next STAGE's wait also covers these stores, which have had a whole MFMA phase to drain) */   \
    L128_FETCH(tile + 2 * stride);                                                             \
    __syncthreads();                                                                           \
  }

  int tile = blockIdx.x;
  const int stride = gridDim.x;
  if (tile >= a.ntiles) return;
  float xa[NCB][16], xb[NCB][16];
  L128_FETCH(tile);
  L128_XLOAD(xa, tile);
  // the scales, read while the first tile travels (the maxima are L2 hits; this workgroup is alone on its CU, so every
  // round trip of its prologue is exposed: they used to come one after the other)
  [[maybe_unused]] int kz = 0, ku = 0;             // float32x2: z is scaled by 2^kz, the accumulators come back by 2^ku
  if constexpr (NP == 2) {
    const int ew = amax_expo(amax_load(a.wamax)), ez = amax_expo(a.z_amax ? amax_load(a.z_amax) : __builtin_bit_cast(unsigned, a.z_amax_static));
    kz = 14 - ez; ku = ew + ez - 28;
  }
  [[maybe_unused]] int kadd = 0, kout = 0;         // pre-split stream: x_l comes back by 2^kadd, x_{l+1} is stored under 2^kout
  if constexpr (ADDPRE) kadd = amax_expo(amax_load(a.add_scale)) - 14;
  if constexpr (YPRE) {
    float bound = bound_margin(__builtin_bit_cast(float, amax_load(a.add_amax)) + a.l1[0]);
    if (a.floor_p != nullptr) {
      const int ef = amax_expo(amax_load(a.floor_p)) - amax_expo(amax_load(a.floor_w)) - 1;
      bound = fmaxf(bound, __builtin_ldexpf(1.f, min(max(ef, -100), 100)));
    }
    kout = 14 - amax_expo(__builtin_bit_cast(unsigned, bound));
    scale_publish(a.scale_out, bound);
  }
  float am = 0.f;
  L128_STAGE(0);
  __syncthreads();
  L128_FETCH(tile + stride);
  while (true) {
    L128_TILE(0, xa, xb);
    tile += stride;
    if (tile >= a.ntiles) break;
    L128_TILE(1, xb, xa);
    tile += stride;
    if (tile >= a.ntiles) break;
  }
  if (a.amax_out != nullptr) amax_commit(am, a.amax_out);
#undef L128_FETCH
#undef L128_STAGE
#undef L128_XLOAD
#undef L128_TILE
}

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

// max |w| of every job's source tensor (format 3 scales the weights by a power of two taken from it): grid
// (AMAX_SLOTS, njob), block x fills slot x
__global__ __launch_bounds__(256) void wamax_kernel(const PackArgs pa) {
  __shared__ float red[4];
  const PackJob& j = pa.job[blockIdx.y];
  const long total = (long)j.K * j.R * j.Cm;
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int mm = (int)(i % j.Cm);
    const long rest = i / j.Cm;
    const int k = (int)(rest % j.R), tap = (int)(rest / j.R);
    m = fmaxf(m, fabsf(j.src[(long)k * j.s_k + (long)mm * j.s_m + (long)tap * j.s_tap]));
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) j.amax[blockIdx.x] = __builtin_bit_cast(unsigned, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// max |x| over a tensor -> out[AMAX_SLOTS] (float bits), zeroed beforehand: the operand scale of a float32x2 launch
// whose operand was not produced by one of this library's amax-publishing epilogues
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long n, unsigned* out) {
  float m = 0.f;
  const long n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {              // four 16-byte loads in flight per thread
    const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v0.x), fabsf(v0.y))), fmaxf(fabsf(v0.z), fabsf(v0.w)));
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v1.x), fabsf(v1.y))), fmaxf(fabsf(v1.z), fabsf(v1.w)));
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v2.x), fabsf(v2.y))), fmaxf(fabsf(v2.z), fabsf(v2.w)));
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v3.x), fabsf(v3.y))), fmaxf(fabsf(v3.z), fabsf(v3.w)));
  }
  for (; i < n4; i += stride) {
    const float4 v = x4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
  amax_commit(m, out);
}

__global__ void pack_kernel(const PackArgs pa) {
  const PackJob& j = pa.job[blockIdx.y];
  if (pa.bf16 != 0) {
    // modes 1 and 2: per tap a slab of Rpad/16 K steps x [piece NP][k-half 2][ldw] 16-byte words, each
    // word the same piece of 8 consecutive k of one column (conv_gemm_x3_kernel's LDS image); NP = 3
    // (mode 2: exact split) or 1 (mode 1: the weight rounded to bf16; the slab keeps its fp32 stride)
    // (format 3: two fp16 pieces of w * 2^(14 - e); the slab keeps format 2's stride between taps, so one workspace
    // layout serves both)
    const int np = pa.bf16 == 2 ? 3 : (pa.bf16 == 3 ? 2 : 1);
    const int groups = j.Rpad / 8;
    const long total = (long)j.K * groups * j.mspan;
    const long tap_words = pa.bf16 >= 2 ? (long)(j.Rpad / 16) * 6 * j.ldw : (long)(j.Rpad / 4) * j.ldw;
    const int kw = pa.bf16 == 3 ? 14 - amax_expo(amax_load(j.amax)) : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int mp = (int)(i % j.mspan);
      const long rest = i / j.mspan;
      const int kg = (int)(rest % groups);
      const int tap = (int)(rest / groups);
      int m = mp;
      if (j.gate_half) {
        const int g = mp >> 6, r = mp & 63;
        m = (r < 32) ? (32 * g + r) : (j.gate_half + 32 * g + (r - 32));
        if (32 * g + (r & 31) >= j.gate_half) m = j.Cm;
      }
      unsigned h[4], md[4], l[4];
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const int k0 = 8 * kg + e, k1 = k0 + 1;
        const float v0 = (k0 < j.R && m < j.Cm) ? j.src[(long)k0 * j.s_k + (long)m * j.s_m + (long)tap * j.s_tap] : 0.f;
        const float v1 = (k1 < j.R && m < j.Cm) ? j.src[(long)k1 * j.s_k + (long)m * j.s_m + (long)tap * j.s_tap] : 0.f;
        if (pa.bf16 == 3) { split2(v0, v1, kw, h[e / 2], md[e / 2]); l[e / 2] = 0u; }
        else split3(v0, v1, h[e / 2], md[e / 2], l[e / 2]);
      }
      uint4* d = reinterpret_cast<uint4*>(j.dst) + tap * tap_words + ((long)(kg >> 1) * 2 * np + (kg & 1)) * j.ldw + j.m_off + mp;
      d[0L * j.ldw] = make_uint4(h[0], h[1], h[2], h[3]);
      if (np >= 2) d[2L * j.ldw] = make_uint4(md[0], md[1], md[2], md[3]);
      if (np == 3) d[4L * j.ldw] = make_uint4(l[0], l[1], l[2], l[3]);
    }
    return;
  }
  // bf16 mode: one 32-bit word holds the pair (k even, k odd); pair-row k/2 sits at row k/2 of the same
  // slab (the slab keeps its fp32 size and offsets, only its first half is used)
  const int rows = pa.bf16 ? j.Rpad / 2 : j.Rpad;
  const long total = (long)j.K * rows * j.mspan;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int mp = (int)(i % j.mspan);
    const long rest = i / j.mspan;
    const int kr = (int)(rest % rows);
    const int tap = (int)(rest / rows);
    int m = mp;
    if (j.gate_half) {
      const int g = mp >> 6, r = mp & 63;
      m = (r < 32) ? (32 * g + r) : (j.gate_half + 32 * g + (r - 32));
      if (32 * g + (r & 31) >= j.gate_half) m = j.Cm;   // beyond the real channels
    }
    float* d = j.dst + ((long)tap * j.Rpad + kr) * j.ldw + j.m_off + mp;
    if (pa.bf16) {
      const int k0 = 2 * kr, k1 = 2 * kr + 1;
      const float v0 = (k0 < j.R && m < j.Cm) ? j.src[(long)k0 * j.s_k + (long)m * j.s_m + (long)tap * j.s_tap] : 0.f;
      const float v1 = (k1 < j.R && m < j.Cm) ? j.src[(long)k1 * j.s_k + (long)m * j.s_m + (long)tap * j.s_tap] : 0.f;
      *reinterpret_cast<unsigned*>(d) = pack_bf16x2(v0, v1);
    } else {
      float v = 0.f;
      if (kr < j.R && m < j.Cm) v = j.src[(long)kr * j.s_k + (long)m * j.s_m + (long)tap * j.s_tap];
      *d = v;
    }
  }
}

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
      for (int i = 0; i + 1 < nsteps; i += 2) {
        adv();
        W3L_FETCH();                                        // step i + 1
        mma(I0{});
        W3L_STAGE(1, true);
        __syncthreads();
        if (i + 2 < nsteps) adv();
        W3L_FETCH();                                        // step i + 2 (past the end: re-reads the last step, unused)
        mma(I1{});
        W3L_STAGE(0, i + 2 < nsteps);
        __syncthreads();
      }
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

// ---------------------------------------------------------------------------
// wl1_kernel -- the weight norms behind the a-priori bounds of the pre-split tensors (see presplit_pair), three
// workgroups per ResidualBlock, once per step beside pack_kernel:
//   out[0] = max_r (sum_c |Wr[r][c]| + |br[r]|)   bounds |Wr z + br| for |z| <= 1                (x_{l+1}'s bound)
//   out[1] = max_c sum_r |Wr[r][c]|,  out[2] = max_c sum_s |Ws[s][c]|   bound |Wr^T g|, |Ws^T g| per unit max |g|  (gh_l's bound)
// Wr (Cr, Ch), Ws (Cs, Ch) row-major (Chainer (Cout, Cin, 1, 1)).  fp32 sums of <= 256 magnitudes: relative error
// 2^-16, inside bound_margin's 2^-10.
// ---------------------------------------------------------------------------
struct L1Job { const float* Wr; const float* br; const float* Ws; float* out; };
struct L1Args { L1Job job[MAXSEG]; int Cr, Cs, Ch; };
// grid (blocks of the stack, 3): y = 0 the row norms of Wr (a wave per row: lanes across the contiguous c axis), y = 1 / 2 the
// column norms of Wr / Ws (a thread per column, coalesced across c, the rows split over the workgroup's thread groups)
__global__ __launch_bounds__(256) void wl1_kernel(const L1Args a) {
  __shared__ float red[256];
  const L1Job& j = a.job[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, Ch = a.Ch;
  float m = 0.f;
  if (blockIdx.y == 0) {
    if (j.Wr)
      for (int r0 = 16 * wave; r0 < a.Cr; r0 += 64) {      // sixteen rows per pass: their loads travel together
        float sum[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          sum[i] = 0.f;
          const int r = min(r0 + i, a.Cr - 1);
#pragma unroll
          for (int q = 0; q < 4; ++q)            // (host: Ch <= 256; a fixed trip count, so that all loads of a pass are requested before the first is used)
            sum[i] += (lane + 64 * q < Ch) ? fabsf(j.Wr[(long)r * Ch + lane + 64 * q]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) sum[i] += __shfl_xor(sum[i], o);
          const int r = min(r0 + i, a.Cr - 1);
          m = fmaxf(m, sum[i] + (j.br ? fabsf(j.br[r]) : 0.f));
        }
      }
  } else {
    const float* W = blockIdx.y == 1 ? j.Wr : j.Ws;
    const int R = blockIdx.y == 1 ? a.Cr : a.Cs;
    const int ngrp = 256 / Ch;                       // thread groups that share the columns (host: Ch <= 256)
    const int c = tid % Ch, grp = tid / Ch;
    float sum = 0.f;
    if (W && grp < ngrp)
#pragma unroll 32
      for (int r = grp; r < R; r += ngrp) sum += fabsf(W[(long)r * Ch + c]);
    red[tid] = sum;
    __syncthreads();
    if (tid < Ch) { float t = 0.f; for (int g = 0; g < ngrp; ++g) t += red[g * Ch + tid]; m = t; }
  }
  m = wave_max(m);
  __syncthreads();
  if (lane == 0) red[wave] = m;
  __syncthreads();
  if (tid == 0) j.out[blockIdx.y] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

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

// sums the split-K partial tiles in split order and applies the linear epilogue
// (bias, residual add, accumulate, relu) of conv_gemm_kernel<EPI_LINEAR>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const GemmArgs a) {
  if (a.skip_flag != nullptr && *a.skip_flag != 0) return;
  const int ntiles_all = a.ntile_m * a.ntile_n * a.B;
  const long total = (long)ntiles_all * (128 * 128);
  float am = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i % (128 * 128));
    const int tile_id = (int)(i / (128 * 128));
    const int mt = tile_id % a.ntile_m;
    const int rest = tile_id / a.ntile_m;
    const int nt = rest % a.ntile_n, b = rest / a.ntile_n;
    const int m = mt * 128 + e / 128, t = nt * 128 + e % 128;
    if (m >= a.M || t >= a.Tout) continue;
    float v = 0.f;
    for (int s = 0; s < a.ksplit; ++s) v += a.partial[((long)s * ntiles_all + tile_id) * (128 * 128) + e];
    const int o = (a.out[1].y != nullptr && m >= a.out[0].rows) ? 1 : 0;
    const OutR& od = a.out[o];
    const int mr = o ? m - a.out[0].rows : m;
    const long off = (long)mr * a.Tout + t;
    if (od.bias) v += od.bias[mr];
    if (od.add) v = lin_combine(v, od.add[(long)b * od.add_bstride + off], od.add_is_mask);
    float* yp = od.y + (long)b * od.y_bstride + off;
    if (od.accumulate) v += *yp;
    if (od.relu) v = fmaxf(v, 0.f);
    am = fmaxf(am, fabsf(v));
    *yp = v;
  }
  if (a.out[0].amax_out != nullptr) amax_commit(am, a.out[0].amax_out);
}

// split-K plan for a small-grid, long-K linear GEMM (see GemmArgs::ksplit); 1 = no split
static int plan_ksplit(int M, int Tout, int B, int nk) {
  if (M % 256 == 0) return 1;
  const long tiles = (long)cdiv(M, 128) * cdiv(Tout, BN) * B;
  if (tiles > 128 || nk < 32) return 1;
  long s = 512 / tiles;                     // fill ~half the chip's 1024 slots
  if (s > nk / 8) s = nk / 8;               // at least 8 K steps per split
  if (s > 32) s = 32;
  return s < 2 ? 1 : (int)s;
}
static size_t ksplit_partial_floats(int M, int Tout, int B, int nk) {
  const int s = plan_ksplit(M, Tout, B, nk);
  return s > 1 ? (size_t)s * cdiv(M, 128) * cdiv(Tout, BN) * B * 128 * 128 : 0;
}

template <int EPI>
static int launch_gemm(GemmArgs& g, int tag, hipStream_t st) {
  // the arithmetic of THIS launch: mode 3 is float32x2 (NP = 2) where the caller provided maxima and a format-3 slab
  // for every segment, mode 2's six-product kernels everywhere else
  const int mode = g_matmul_dtype == 3 ? (g.f16x2 ? 3 : 2) : g_matmul_dtype;
  VQ_REQUIRE(!g.f16x2 || g_matmul_dtype == 3, "conv_gemm: float32x2 launch outside matmul mode 3");
  if (mode == 3)
    for (int i = 0; i < g.nseg; ++i)
      VQ_REQUIRE(g.seg[i].wamax && (g.seg[i].amax || g.seg[i].amax_static > 0.f), "conv_gemm: float32x2 segment %d without its maxima", i);
  const bool big = (g.M % 256 == 0) && (EPI != EPI_GATE_BWD);    // 256-row tiles (8 waves)
  const int bm = big ? 256 : 128;
  g.ntile_m = cdiv(g.M, bm);
  g.ntile_n = cdiv(g.Tout, BN);
  for (int i = 0; i < g.nseg; ++i) g.seg[i].vec = seg_vec_ok(g.seg[i]) ? 1 : 0;
  if (EPI == EPI_LINEAR && g.out[1].y != nullptr)
    VQ_REQUIRE(g.out[0].rows % 32 == 0, "conv_gemm: first output range must be a multiple of 32 rows");
  if (EPI == EPI_LINEAR && g.out[1].y == nullptr) g.out[0].rows = g.M;
  const long nblk = (long)g.ntile_m * g.ntile_n * g.B;
  if (nblk <= 0) return 0;
  VQ_REQUIRE(nblk < (1L << 31), "conv_gemm: grid too large");
  if (mode != 0)          // modes 1 - 3 address a batch item's activations with 32-bit buffer offsets
    for (int i = 0; i < g.nseg; ++i)
      VQ_REQUIRE((long)g.seg[i].cin * g.seg[i].x_cstride * 4 < (1L << 31), "conv_gemm: one batch item of segment %d exceeds 2 GB (Cin * T * 4 bytes)", i);
  // split-K when the caller provided a partial-tile buffer and the shape calls for it
  int nk = 0;
  for (int i = 0; i < g.nseg; ++i) nk += cdiv(g.seg[i].cin, BK);
  g.ksplit = 1;
  if (EPI == EPI_LINEAR && !big && g.partial != nullptr) {      // (a requested max |y| is then published by the reduce kernel)
    const int sp = plan_ksplit(g.M, g.Tout, g.B, nk);
    if (sp > 1) { g.ksplit = sp; g.ksteps_per_split = cdiv(nk, sp); }
  }
  const long grid = nblk * g.ksplit;
  // timing (vqvae_prof_*): the GEMM kernel of this call is timed by its own dispatch's events (a split-K reduce behind it is not)
  hipEvent_t pe0 = nullptr, pe1 = nullptr;
  const bool attach = prof_enabled(tag);     // (the pair is registered in front of the launch itself: every check below may still return)
  ProfScope ps(attach ? 0 : tag, st);
#define LG_LAUNCH(KERNEL, GRID, BLOCK, ARG)                                                                    \
  do {                                                                                                          \
    if (attach && prof_attach(tag, &pe0, &pe1)) hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, st, pe0, pe1, 0, ARG); \
    else hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, st, ARG);                                                   \
  } while (0)
  // the K = 128 -> 256-row projection with residual add (the ResidualBlock `res` conv): streaming kernel
  if constexpr (EPI == EPI_LINEAR) {
    static const int lin128 = getenv("VQVAE_LIN128") ? atoi(getenv("VQVAE_LIN128")) : 32;   // 0: off; 32 / 64: column tile
    const Seg& s0 = g.seg[0];
    if (lin128 && mode != 0 && g.nseg == 1 && g.M == 256 && s0.cin == 128 && g.out[1].y == nullptr &&
        !g.out[0].relu && !g.out[0].accumulate && !g.out[0].add_is_mask && s0.tmul == 1 && s0.tdiv == 1 && s0.toff == 0 && s0.Tin == g.Tout &&
        s0.x_cstride == g.Tout && g.Tout % 64 == 0 && s0.vec && s0.ldw >= 256 && g.lerp.P == nullptr &&
        g.skip_flag == nullptr && g.ksplit == 1) {
      Lin128Args la;
      la.w = reinterpret_cast<const uint4*>(s0.w); la.ldw = s0.ldw;
      la.z = s0.x; la.z_bstride = s0.x_bstride;
      la.add = g.out[0].add; la.add_bstride = g.out[0].add_bstride;
      la.y = g.out[0].y; la.y_bstride = g.out[0].y_bstride;
      la.bias = g.out[0].bias;
      la.T = g.Tout;
      la.wamax = s0.wamax; la.z_amax = s0.amax; la.z_amax_static = s0.amax_static; la.amax_out = g.out[0].amax_out;
      la.add16 = g.add16; la.y16 = g.y16;
      la.add_scale = g.add_scale; la.add_amax = g.add_amax; la.l1 = g.bound_l1; la.scale_out = g.scale_out;
      la.floor_w = g.floor_w; la.floor_p = g.floor_p;
      const int nc = (lin128 == 32 || g.z16) ? 32 : 64;
      la.tiles_per_b = g.Tout / nc; la.ntiles = la.tiles_per_b * g.B;
      static int n_cu = 0;
      if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
      }
      const unsigned nwg = (unsigned)(la.ntiles < n_cu ? la.ntiles : n_cu);
#define L128_LAUNCH(NPv, NCv)                                                                                        \
      do {                                                                                                           \
        if (la.add) LG_LAUNCH((lin128_stream_kernel<NPv, NCv, true>), dim3(nwg), dim3(512), la);     \
        else LG_LAUNCH((lin128_stream_kernel<NPv, NCv, false>), dim3(nwg), dim3(512), la);           \
      } while (0)
      if (mode == 2) { if (nc == 32) L128_LAUNCH(3, 32); else L128_LAUNCH(3, 64); }
      else if (mode == 3 && (g.add16 || g.y16)) {
        VQ_REQUIRE(la.add && g.y16 && g.add_amax && g.bound_l1 && g.scale_out && (!g.add16 || g.add_scale),
                   "conv_gemm: a pre-split residual stream needs the residual add, a pre-split output and its bound's inputs");
        if (g.add16) LG_LAUNCH((lin128_stream_kernel<2, 32, true, false, true, true>), dim3(nwg), dim3(512), la);
        else LG_LAUNCH((lin128_stream_kernel<2, 32, true, false, false, true>), dim3(nwg), dim3(512), la);
      }
      else if (mode == 3) L128_LAUNCH(2, 32);
      else if (g.add16 || g.y16) {
        VQ_REQUIRE(g.z16 && la.add && g.y16, "conv_gemm: a bf16 residual stream needs matmul mode 1's bf16 z, the residual add and a bf16 output");
        if (g.add16) LG_LAUNCH((lin128_stream_kernel<1, 32, true, true, true, true>), dim3(nwg), dim3(512), la);
        else LG_LAUNCH((lin128_stream_kernel<1, 32, true, true, false, true>), dim3(nwg), dim3(512), la);
      }
      else if (g.z16) {
        if (la.add) LG_LAUNCH((lin128_stream_kernel<1, 32, true, true>), dim3(nwg), dim3(512), la);
        else LG_LAUNCH((lin128_stream_kernel<1, 32, false, true>), dim3(nwg), dim3(512), la);
      }
      else { if (nc == 32) L128_LAUNCH(1, 32); else L128_LAUNCH(1, 64); }
#undef L128_LAUNCH
      VQ_LAUNCH_CHECK();
      return 0;
    }
  }
  VQ_REQUIRE(mode == 3 || g_matmul_dtype != 3 || (!g.x16 && !g.h16 && !g.add16 && !g.y16), "conv_gemm: pre-split tensors need a float32x2 launch (every segment with its maxima)");
  if (g.add16 || g.y16)
    VQ_REQUIRE(EPI == EPI_LINEAR && mode == 1 && big && g.Tout % BN == 0 && g.M % 256 == 0 && g.out[1].y == nullptr && !g.out[0].bias &&
               !g.out[0].relu && !g.out[0].accumulate && g.ksplit == 1,
               "conv_gemm: a bf16 residual / gradient stream needs matmul mode 1, whole 256-row tiles and a plain (acc + add) epilogue");
  // 256-column tiles when they still give every CU a workgroup (measured at configs[1]: dilated conv
  // forward and backward-data -6.5 %, the short 1x1 contractions unchanged)
  const long nblk2 = (long)g.ntile_m * cdiv(g.Tout, 2 * BN) * g.B;
  // two taps of one tensor: interleave them channel group by channel group (TAP2).  The choice depends
  // on the contraction only, never on the tile shape, so that a result does not change with the batch size.
  // A 1x1 conv over >= 64 channels into 256-row tiles (proj1 / proj2 and their backward-data, the latent-rate condition
  // projection): ONE segment, so it used to miss the two-tap loop and run 256 x 256 tiles, one workgroup per CU, 16 K steps
  // between a prologue and a 256 KB epilogue -- 16 GFLOP in 105 us, 0.18 of the three-product ceiling.  Its contraction is
  // presented as TWO segments, the lower and the upper half of the channels (same tensor, same shift; the second slab is the
  // first one's upper K steps), which the TAP2 / LEAN loop interleaves like two taps: two workgroups per CU, one tile's
  // epilogue beside the other's loop.  Depends on the contraction only (never on B or T); the K order changes, the products do not.
  // (proj1 / proj2 forward and backward-data 105 -> ~70 us each, the step 14.87 -> 14.73 ms: same box, two interleaved rounds)
  if constexpr (EPI == EPI_LINEAR) {
    Seg& s0 = g.seg[0];
    if (big && (mode == 2 || mode == 3) && g.nseg == 1 && g.ksplit == 1 && s0.cin >= 64 && s0.cin % 32 == 0 &&
        !(g.M == 256 && s0.cin == 128) &&      // (the residual 1x1's shape keeps the K order of lin128_stream_kernel, its bitwise twin)
        s0.tmul == 1 && s0.tdiv == 1 && !g.x16 && !g.z16 && !g.add16 && !g.y16) {
      Seg& s1 = g.seg[1];
      s1 = s0;
      const int half = s0.cin / 2;
      s0.cin = s1.cin = half;
      s1.x = s0.x + (long)half * s0.x_cstride;
      s1.w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(s0.w) + (size_t)(half / BK) * 32u * (mode == 3 ? 2 : 3) * (size_t)s0.ldw);
      g.nseg = 2;
    }
  }
  const bool tap2 = mode != 0 && g.nseg == 2 && g.ksplit == 1 &&
                    g.seg[0].cin == g.seg[1].cin && g.seg[0].cin % BK == 0 &&
                    g.seg[0].x_cstride == g.seg[1].x_cstride && g.seg[0].x_bstride == g.seg[1].x_bstride &&
                    g.seg[0].Tin == g.seg[1].Tin && g.seg[0].tmul == g.seg[1].tmul && g.seg[0].tdiv == g.seg[1].tdiv &&
                    g.seg[0].ldw == g.seg[1].ldw && g.seg[1].w >= g.seg[0].w &&          // second slab addressed off the first's descriptor
                    (reinterpret_cast<const char*>(g.seg[1].w) - reinterpret_cast<const char*>(g.seg[0].w)) < (1L << 30);
  // Two taps, two or more pieces per operand: 256 x 128 tiles whose loop fits 128 VGPRs (LEAN in conv_gemm_x3_kernel), TWO
  // workgroups per CU -- one tile's epilogue beside the other's K loop: gate kernel 211 -> 197 us, backward-data
  // 217 -> 198 us at configs[1] against the 256 x 256 tiles, which stay for every other contraction.  Same K
  // order and products as the 256 x 256-tile two-tap kernel (VQVAE_X3_LEAN=0, the A/B alternate): the choice never changes a result.
  static const int x3_lean = getenv("VQVAE_X3_LEAN") ? atoi(getenv("VQVAE_X3_LEAN")) : X3_LEAN;
  const bool lean = x3_lean && X3_LEAN && tap2 && big && mode != 0 && EPI != EPI_GATE_BWD;   // mode 1: the 256 x 128 kernel needs 122 VGPRs as it is
  const bool wide = mode != 0 && big && !lean && nblk2 >= 256;
  if constexpr (EPI == EPI_GATE) {       // the latent-rate condition as one more K step: the 256 x 128-tile two-tap loop, modes 2 / 3 (see the kernel)
    g.lerp.fold = (g.lerp.fold && g.lerp.P && lean && (mode == 2 || (mode == 3 && g.lerp.amax)) && g.Tout % BN == 0 &&
                   (long)g.Tout >= 26L * g.lerp.Tl) ? 1 : 0;
  }
  if (wide) g.ntile_n = cdiv(g.Tout, 2 * BN);
#define X3_LAUNCH(WMv, NBv, NPv, blocks, threads)                                                                    \
  do {                                                                                                                \
    if (tap2) LG_LAUNCH((conv_gemm_x3_kernel<EPI, WMv, NBv, NPv, true>), dim3((unsigned)(blocks)), dim3(threads), g);  \
    else LG_LAUNCH((conv_gemm_x3_kernel<EPI, WMv, NBv, NPv, false>), dim3((unsigned)(blocks)), dim3(threads), g);      \
  } while (0)
#define X3_LAUNCH_MODE(WMv, NBv, blocks, threads)                                                                    \
  do {                                                                                                                \
    if (mode == 2) X3_LAUNCH(WMv, NBv, 3, blocks, threads);                                                           \
    else if (mode == 3) X3_LAUNCH(WMv, NBv, 2, blocks, threads);                                                      \
    else X3_LAUNCH(WMv, NBv, 1, blocks, threads);                                                                     \
  } while (0)
  // the gate-derivative epilogue always runs 128-row tiles (`big` is false): its 256-row variants are
  // not instantiated
  // activations stored as bf16 (matmul mode 1): a linear GEMM over z tensors (z16: every segment), or the caller's mask
  const int xm = (EPI == EPI_LINEAR && g.z16) ? 3 : g.x16;
  if (mode == 3 && (xm != 0 || g.h16 || (EPI == EPI_GATE_BWD && g.pb_part))) {       // float32x2 with pre-split tensors (see presplit_pair) / the fused pull-back
    VQ_REQUIRE(g.ksplit == 1, "conv_gemm: pre-split tensors: no split-K");
    if constexpr (EPI == EPI_GATE) {
      VQ_REQUIRE(tap2 && lean && xm == 3 && g.seg[0].amax == g.seg[1].amax && g.seg[0].wamax == g.seg[1].wamax,
                 "conv_gemm: gate GEMM over a pre-split x: both taps of one tensor, 256 x 128 tiles");
      LG_LAUNCH((conv_gemm_x3_kernel<EPI_GATE, 4, 1, 2, true, 3>), dim3((unsigned)nblk), dim3(512), g);
    } else if constexpr (EPI == EPI_LINEAR) {
      VQ_REQUIRE(tap2 && lean && xm == 3 && g.seg[0].amax == g.seg[1].amax && g.seg[0].wamax == g.seg[1].wamax && !g.add16 && !g.y16,
                 "conv_gemm: backward-data GEMM over a pre-split gh: both taps of one tensor, 256 x 128 tiles");
      LG_LAUNCH((conv_gemm_x3_kernel<EPI_LINEAR, 4, 1, 2, true, 3>), dim3((unsigned)nblk), dim3(512), g);
    } else {
      VQ_REQUIRE(xm == 0 && (!g.h16 || (g.bound_l1 && g.scale_out)), "conv_gemm: gate-derivative GEMM storing a pre-split gh needs its bound's inputs (fp32 operands)");
      if (g.pb_part) VQ_REQUIRE(g.M == 128 && g.Tout % BN == 0 && g.lerp.v0 && g.lerp.w0 && g.lerp.w1 && (long)g.Tout >= 64L * g.lerp.Tl,
                                "conv_gemm: the fused latent pull-back serves 128 gate channels, T %% 128 == 0, T >= 64 Tl");
      const int out = (g.h16 ? 1 : 0) | (g.pb_part ? 2 : 0);
#define GB_LAUNCH(OUTv)                                                                                                  \
      do {                                                                                                                \
        if (tap2) LG_LAUNCH((conv_gemm_x3_kernel<EPI_GATE_BWD, 2, 1, 2, true, 0, OUTv>), dim3((unsigned)grid), dim3(256), g); \
        else LG_LAUNCH((conv_gemm_x3_kernel<EPI_GATE_BWD, 2, 1, 2, false, 0, OUTv>), dim3((unsigned)grid), dim3(256), g);  \
      } while (0)
      if (out == 1) GB_LAUNCH(1); else if (out == 2) GB_LAUNCH(2); else GB_LAUNCH(3);
#undef GB_LAUNCH
    }
    VQ_LAUNCH_CHECK();
    return 0;
  }
  if (xm != 0) {
    VQ_REQUIRE(mode == 1 && g.ksplit == 1, "conv_gemm: bf16-stored activations need matmul mode 1");
    for (int i = 0; i < g.nseg; ++i) VQ_REQUIRE(g.seg[i].tmul == 1 && g.seg[i].tdiv == 1, "conv_gemm: bf16-stored activations: stride-1 segments only");
    if constexpr (EPI == EPI_LINEAR) {
      VQ_REQUIRE(big && xm == 3, "conv_gemm: bf16-stored activations of a linear GEMM: every segment, 256-row tiles");
      if (tap2 && lean) LG_LAUNCH((conv_gemm_x3_kernel<EPI_LINEAR, 4, 1, 1, true, 3>), dim3((unsigned)nblk), dim3(512), g);
      else if (wide) LG_LAUNCH((conv_gemm_x3_kernel<EPI_LINEAR, 4, 2, 1, false, 1>), dim3((unsigned)nblk2), dim3(512), g);
      else LG_LAUNCH((conv_gemm_x3_kernel<EPI_LINEAR, 4, 1, 1, false, 1>), dim3((unsigned)nblk), dim3(512), g);
    } else if constexpr (EPI == EPI_GATE_BWD) {
      VQ_REQUIRE(tap2 && xm == 1, "conv_gemm: gate-derivative GEMM with a bf16-stored g_res: [g_res | g_skip] of one shape");
      LG_LAUNCH((conv_gemm_x3_kernel<EPI_GATE_BWD, 2, 1, 1, true, 1>), dim3((unsigned)grid), dim3(256), g);
    } else if constexpr (EPI == EPI_GATE) {
      VQ_REQUIRE(tap2 && lean && xm == 3, "conv_gemm: gate GEMM over a bf16-stored x: both taps, 256 x 128 tiles");
      LG_LAUNCH((conv_gemm_x3_kernel<EPI_GATE, 4, 1, 1, true, 3>), dim3((unsigned)nblk), dim3(512), g);
    } else {
      VQ_REQUIRE(false, "conv_gemm: bf16-stored activations are not built for this epilogue");
    }
    VQ_LAUNCH_CHECK();
    return 0;
  }
  if constexpr (EPI != EPI_GATE_BWD) {
    if (big && mode != 0) {
      if (wide) X3_LAUNCH_MODE(4, 2, nblk2, 512);
      else X3_LAUNCH_MODE(4, 1, nblk, 512);
    } else if (big) {
      LG_LAUNCH((conv_gemm_kernel<EPI, 4, false>), dim3((unsigned)nblk), dim3(512), g);
    }
    if (!big) {
      if (mode != 0) X3_LAUNCH_MODE(2, 1, grid, 256);
      else LG_LAUNCH((conv_gemm_kernel<EPI, 2, false>), dim3((unsigned)grid), dim3(256), g);
    }
  } else {
    if (mode != 0) X3_LAUNCH_MODE(2, 1, grid, 256);
    else LG_LAUNCH((conv_gemm_kernel<EPI, 2, false>), dim3((unsigned)grid), dim3(256), g);
  }
#undef X3_LAUNCH_MODE
#undef X3_LAUNCH
#undef LG_LAUNCH
  VQ_LAUNCH_CHECK();
  if (g.ksplit > 1) {
    const long total = nblk * 128 * 128;
    int nb = (int)((total + 255) / 256);
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(nb), dim3(256), 0, st, g);
    VQ_LAUNCH_CHECK();
  }
  return 0;
}

// fmt: the slab format (PackArgs::bf16), -1 = the matmul mode's default (mode 3: format 2 -- its float32x2 launches ask
// for format 3 explicitly and give every job its AMAX_SLOTS `amax` words)
static int launch_pack(PackArgs& pa, hipStream_t st, int fmt = -1) {
  if (pa.njob == 0) return 0;
  pa.bf16 = fmt >= 0 ? fmt : (g_matmul_dtype == 3 ? 2 : g_matmul_dtype);
  long mx = 0;
  for (int i = 0; i < pa.njob; ++i) {
    long t = (long)pa.job[i].K * pa.job[i].Rpad * pa.job[i].mspan;
    if (pa.bf16 != 0) t /= 8;
    if (t > mx) mx = t;
  }
  int nb = (int)((mx + 255) / 256);
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  if (pa.bf16 == 3) {
    for (int i = 0; i < pa.njob; ++i) VQ_REQUIRE(pa.job[i].amax != nullptr, "pack: format 3 job without an amax slot");
    hipLaunchKernelGGL(wamax_kernel, dim3(AMAX_SLOTS, pa.njob), dim3(256), 0, st, pa);
    VQ_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(pack_kernel, dim3(nb, pa.njob), dim3(256), 0, st, pa);
  VQ_LAUNCH_CHECK();
  return 0;
}

// A^T slab for the forward GEMM of a Chainer (Cout,Cin,K) weight: k=ci, m=co
static PackJob pack_fwd_job(float* dst, const float* W, int Cout, int Cin, int K, int gate_half,
                            int ldw, int m_off, int mspan) {
  PackJob j;
  j.dst = dst; j.src = W; j.R = Cin; j.Cm = Cout; j.K = K;
  j.s_k = K; j.s_m = (long)Cin * K; j.s_tap = 1;
  j.gate_half = gate_half; j.Rpad = pad16(Cin); j.ldw = ldw; j.m_off = m_off; j.mspan = mspan;
  j.amax = nullptr;
  return j;
}
// A^T slab for the bwd-data GEMM: k=co, m=ci
static PackJob pack_bwd_job(float* dst, const float* W, int Cout, int Cin, int K, int ldw) {
  PackJob j;
  j.dst = dst; j.src = W; j.R = Cout; j.Cm = Cin; j.K = K;
  j.s_k = (long)Cin * K; j.s_m = K; j.s_tap = 1;
  j.gate_half = 0; j.Rpad = pad16(Cout); j.ldw = ldw; j.m_off = 0; j.mspan = ldw;
  j.amax = nullptr;
  return j;
}

struct WgradPlan { int ntile_m, ntile_n, steps_per_b, steps_per_split, nsplit, nseg; size_t slab_floats, bslab_floats; };

// wgrad_kernel runs 2 workgroups per CU (205 VGPR): 512 resident slots on 256 CUs.  The K axis is
// the flattened (batch, time) axis cut into WBK-wide steps; choose the number of K splits so that
// tiles x splits is just under a whole number of residency rounds, with as few splits as that
// allows (every split costs one 64 KB partial slab per tile, written and re-read by the reduce).
#ifndef W3_SLOTS_256
#define W3_SLOTS_256 512
#endif
static WgradPlan plan_wgrad(int M, int B, int Tout, const int* cins, int nseg) {
  WgradPlan p;
  p.nseg = nseg;
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
  const long slots = (M % 256 == 0 && g_matmul_dtype != 0) ? W3_SLOTS_256 : 512;
  long want = 1;
  double best = -1.0;
  for (long w = 1; w <= maxs; ++w) {
    const long sps = (total_steps + w - 1) / w;
    const long ns = (total_steps + sps - 1) / sps;        // splits actually produced
    const long blocks = tiles * ns;
    const long rounds = (blocks + slots - 1) / slots;
    const double eff = (double)blocks / (double)(rounds * slots);
    if (eff > best + 1e-9) { best = eff; want = w; }
    if (eff >= 0.92 && blocks >= slots) { want = w; break; }
  }
  p.steps_per_split = (int)((total_steps + want - 1) / want);
  p.nsplit = (int)((total_steps + p.steps_per_split - 1) / p.steps_per_split);
  p.slab_floats = (size_t)p.nsplit * p.ntile_m * p.ntile_n * BM * BN;
  p.bslab_floats = (size_t)p.nsplit * nseg * p.ntile_m * BM;
  return p;
}

static int launch_wgrad(WgradArgs& w, const WgradPlan& p, float* ws, int tag, hipStream_t st) {
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
    if (w.x16 && w.g16) hipLaunchKernelGGL((wgrad3_kernel<4, 1, 2, true, true>), grid, dim3(512), 0, st, w);
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
    if (w.x16 && w.g16) hipLaunchKernelGGL((wgrad3_kernel<4, 1, 1, true, true>), grid, dim3(512), 0, st, w);
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

using namespace vq;

extern "C" int vqvae_set_matmul_dtype(int dtype) {
  VQ_REQUIRE(dtype >= 0 && dtype <= 3, "set_matmul_dtype: 0 (fp32 MFMA), 1 (bf16 operands), 2 (fp32 as six bf16 MFMA products) or 3 (fp32 as three fp16 MFMA products)");
  vq::g_matmul_dtype = dtype;
  return 0;
}
extern "C" int vqvae_get_matmul_dtype(void) { return vq::g_matmul_dtype; }
#ifdef VQ_PHASE_TIMING
extern "C" int vqvae_debug_phases(unsigned long long* out, int reset) {       // dev aid, see g_phase
  VQ_CHECK_HIP(hipDeviceSynchronize());
  VQ_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(vq::g_phase), sizeof(unsigned long long) * 24));
  if (reset) { unsigned long long z[24] = {0}; VQ_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(vq::g_phase), z, sizeof(z))); }
  return 0;
}
#endif
extern "C" int vqvae_set_wgrad_impl(int impl) {
  VQ_REQUIRE(impl == 0 || impl == 1, "set_wgrad_impl: 0 (auto) or 1 (generic kernel)");
  vq::g_wgrad_impl = impl;
  return 0;
}

// ---------------------------------------------------------------------------
// C ABI: generic conv1d
// ---------------------------------------------------------------------------
static int check_conv_desc(const vqvae_conv1d_desc* d) {
  VQ_REQUIRE(d, "conv1d: null desc");
  VQ_REQUIRE(d->B > 0 && d->Cin > 0 && d->Cout > 0 && d->Tin > 0 && d->Tout > 0, "conv1d: bad dims");
  VQ_REQUIRE(d->K >= 1 && d->K <= MAXTAPS, "conv1d: K=%d unsupported (1..%d)", d->K, MAXTAPS);
  VQ_REQUIRE(d->stride >= 1 && d->dil >= 1 && d->pad >= 0, "conv1d: bad stride/dil/pad");
  const int nat = (d->Tin + 2 * d->pad - d->dil * (d->K - 1) - 1) / d->stride + 1;
  VQ_REQUIRE(d->Tout <= nat, "conv1d: Tout=%d exceeds natural output length %d", d->Tout, nat);
  return 0;
}

static size_t conv_pack_floats(const vqvae_conv1d_desc* d) {
  size_t f = (size_t)d->K * slab_rows(d->Cin) * pad128(d->Cout);
  size_t b = (size_t)d->K * slab_rows(d->Cout) * pad128(d->Cin);
  return f > b ? f : b;
}

// Stride-2 convs (the encoder, net.py:14-28): their weight gradient contracts gy[t] with x[2 t + e].  The fast
// weight-gradient kernels read 16-byte runs of consecutive t, so x is first split into its even and odd phases
// (one pass over x, into the workspace); tap e then is a STRIDE-1 segment of one phase: x[2 t + e] = xe[t + e / 2]
// for even e, xo[t + (e - 1) / 2] for odd e.  (Until round 3 these layers ran the generic fp32-MFMA kernel.)
static int phase_pitch(const vqvae_conv1d_desc* d) { return ((d->Tin + 1) / 2 + 3) & ~3; }
static bool phase_split_ok(const vqvae_conv1d_desc* d) {
  return d->stride == 2 && d->dil == 1 && d->K <= MAXSEG && d->Tout % 4 == 0 && d->Tin >= 8;
}
static size_t phase_split_floats(const vqvae_conv1d_desc* d) {
  return phase_split_ok(d) ? (size_t)2 * d->B * d->Cin * phase_pitch(d) + 64 : 0;
}
__global__ void phase_split_kernel(const float* __restrict__ x, long rows, int Tin, int Tp, float* __restrict__ xe,
                                   float* __restrict__ xo, const int32_t* __restrict__ skip_flag) {
  if (skip_flag != nullptr && *skip_flag != 0) return;
  const long total = rows * Tp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / Tp;
    const int tp = (int)(i - r * Tp);
    const float* xr = x + r * Tin;
    xe[i] = 2 * tp < Tin ? xr[2 * tp] : 0.f;
    xo[i] = 2 * tp + 1 < Tin ? xr[2 * tp + 1] : 0.f;
  }
}

// matmul mode 3, generic conv entry points: the float32x2 kernels need the operands' absolute maxima before they run.
// Nobody hands them to these entry points, so a launch large enough to pay for it (>= 8 GFLOP: proj1 / proj2 at the
// configs; vqvae_set_f32x2_min_gflop overrides, 0 = every launch, which is how the parity and
// accuracy tests reach these kernels at their small shapes) runs
// absmax_kernel over its activation operand first (one read of the tensor: ~25 us per 126 MB against ~60 us saved);
// everything smaller keeps mode 2's kernels.  The maxima live in the last 64 bytes of the workspace.
static double g_f32x2_min_gflop = -1.0;        // < 0: not set (vqvae_set_f32x2_min_gflop), 8 then
static bool conv_f16x2(const vqvae_conv1d_desc* d) {
  if (g_matmul_dtype != 3) return false;
  if (g_f32x2_min_gflop < 0.0) g_f32x2_min_gflop = 8.0;
  return 2.0 * d->B * d->Tout * (double)d->Cout * d->Cin * d->K >= g_f32x2_min_gflop * 1e9;
}
extern "C" int vqvae_set_f32x2_min_gflop(double gflop) {
  VQ_REQUIRE(gflop >= 0.0, "set_f32x2_min_gflop: negative threshold");
  g_f32x2_min_gflop = gflop;
  return 0;
}
static int launch_absmax(const float* x, long n, unsigned* out, hipStream_t st) {      // out[AMAX_SLOTS] zeroed beforehand
  long nb = (n / 16 + 255) / 256;
  if (nb > 1024) nb = 1024;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, n, out);
  VQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int vqvae_absmax(const float* x, size_t n, uint32_t* amax, vqvae_stream_t s) {
  VQ_REQUIRE(x && amax, "absmax: null pointer");
  hipStream_t st = (hipStream_t)s;
  VQ_CHECK_HIP(hipMemsetAsync(amax, 0, AMAX_SLOTS * sizeof(uint32_t), st));
  return launch_absmax(x, (long)n, amax, st);
}

extern "C" size_t vqvae_conv1d_workspace_bytes(const vqvae_conv1d_desc* d) {
  if (!d) return 0;
  int cins[MAXTAPS];
  for (int i = 0; i < d->K && i < MAXTAPS; ++i) cins[i] = d->Cin;
  WgradPlan p = plan_wgrad(d->Cout, d->B, d->Tout, cins, d->K < MAXTAPS ? d->K : MAXTAPS);
  size_t wg = (p.slab_floats + p.bslab_floats) * sizeof(float) + phase_split_floats(d) * sizeof(float);
  size_t pk = conv_pack_floats(d) * sizeof(float);
  // forward / backward-data may split K: packed weights first, then the partial tiles
  size_t pf = ksplit_partial_floats(d->Cout, d->Tout, d->B, d->K * cdiv(d->Cin, BK));
  size_t pb = ksplit_partial_floats(d->Cin, d->Tin, d->B, d->K * cdiv(d->Cout, BK));
  size_t gm = align_up(pk, 256) + (pf > pb ? pf : pb) * sizeof(float);
  return align_up(wg > gm ? wg : gm, 256) + 512;      // (the last 128 bytes: maxima of a float32x2 launch)
}
static unsigned* conv_amax_slots(const vqvae_conv1d_desc* d, void* ws) {
  return reinterpret_cast<unsigned*>((char*)ws + vqvae_conv1d_workspace_bytes(d) - 2 * AMAX_SLOTS * sizeof(unsigned));
}

// ---- weights packed AHEAD of the launch that reads them (vqvae_conv1d_amax::packed).  The forward / backward-data entry
// points re-lay W into their workspace in front of every GEMM: one or two small launches (wamax_kernel, pack_kernel) that a
// latency-bound chain of small convs -- the encoder, the condition embed, their backward -- pays once per conv on its
// critical path.  The weights only change in the optimizer, so a caller may pack all of a step's slabs at once, on
// another stream, as soon as the optimizer is done (vqvae_amd/backend.py: PackPrefetch), and hand each launch its slab.
// A packed buffer = the slab, then (256-byte aligned) the AMAX_SLOTS words of max |W| that a float32x2 launch reads.
static size_t conv_slab_bytes(const vqvae_conv1d_desc* d, int backward) {
  const size_t f = backward ? (size_t)d->K * slab_rows(d->Cout) * pad128(d->Cin) : (size_t)d->K * slab_rows(d->Cin) * pad128(d->Cout);
  return align_up(f * sizeof(float), 256);
}
extern "C" size_t vqvae_conv1d_packed_bytes(const vqvae_conv1d_desc* d, int backward) {
  if (!d || d->K < 1 || d->K > MAXTAPS) return 0;
  return conv_slab_bytes(d, backward) + 256;
}
extern "C" int vqvae_conv1d_pack(int n, const vqvae_conv1d_desc* descs, const float* const* W, const int* backward,
                                 void* const* packed, vqvae_stream_t s) {
  VQ_REQUIRE(n >= 0 && (n == 0 || (descs && W && backward && packed)), "conv1d_pack: null pointer");
  hipStream_t st = (hipStream_t)s;
  for (int fmt3 = 0; fmt3 < 2; ++fmt3) {           // one run of launches per slab format (float32x2 launches: format 3 + the maxima)
    PackArgs pa; pa.njob = 0;
    for (int i = 0; i < n; ++i) {
      const vqvae_conv1d_desc* d = descs + i;
      if (int e = check_conv_desc(d)) return e;
      VQ_REQUIRE(W[i] && packed[i], "conv1d_pack: null pointer in job %d", i);
      if ((conv_f16x2(d) ? 1 : 0) != fmt3) continue;
      PackJob j = backward[i] ? pack_bwd_job((float*)packed[i], W[i], d->Cout, d->Cin, d->K, pad128(d->Cin))
                              : pack_fwd_job((float*)packed[i], W[i], d->Cout, d->Cin, d->K, 0, pad128(d->Cout), 0, pad128(d->Cout));
      j.amax = fmt3 ? reinterpret_cast<unsigned*>((char*)packed[i] + conv_slab_bytes(d, backward[i])) : nullptr;
      pa.job[pa.njob++] = j;
      if (pa.njob == MAXSEG) { if (int e = launch_pack(pa, st, fmt3 ? 3 : -1)) return e; pa.njob = 0; }
    }
    if (int e = launch_pack(pa, st, fmt3 ? 3 : -1)) return e;
  }
  return 0;
}

static int conv1d_fwd_impl(const vqvae_conv1d_desc* d, const float* x, const float* W, const float* b, float* y,
                           void* ws, size_t ws_bytes, const int32_t* skip_flag, const vqvae_conv1d_amax* cam,
                           vqvae_stream_t s);
extern "C" int vqvae_conv1d_fwd(const vqvae_conv1d_desc* d, const float* x, const float* W,
                                const float* b, float* y, void* ws, size_t ws_bytes,
                                vqvae_stream_t s) {
  return conv1d_fwd_impl(d, x, W, b, y, ws, ws_bytes, nullptr, nullptr, s);
}
extern "C" int vqvae_conv1d_fwd_cond(const vqvae_conv1d_desc* d, const float* x, const float* W,
                                     const float* b, float* y, void* ws, size_t ws_bytes,
                                     const int32_t* skip_flag, vqvae_stream_t s) {
  return conv1d_fwd_impl(d, x, W, b, y, ws, ws_bytes, skip_flag, nullptr, s);
}
extern "C" int vqvae_conv1d_fwd_amax(const vqvae_conv1d_desc* d, const float* x, const float* W,
                                     const float* b, float* y, void* ws, size_t ws_bytes,
                                     const vqvae_conv1d_amax* amax, vqvae_stream_t s) {
  return conv1d_fwd_impl(d, x, W, b, y, ws, ws_bytes, nullptr, amax, s);
}
extern "C" int vqvae_conv1d_uses_f32x2(const vqvae_conv1d_desc* d) { return (d && conv_f16x2(d)) ? 1 : 0; }

static int conv1d_fwd_impl(const vqvae_conv1d_desc* d, const float* x, const float* W, const float* b, float* y,
                           void* ws, size_t ws_bytes, const int32_t* skip_flag, const vqvae_conv1d_amax* cam,
                           vqvae_stream_t s) {
  if (int e = check_conv_desc(d)) return e;
  VQ_REQUIRE(x && W && y && ws, "conv1d_fwd: null pointer");
  hipStream_t st = (hipStream_t)s;
  const int ldw = pad128(d->Cout), rp = slab_rows(d->Cin);
  if ((size_t)d->K * rp * ldw * sizeof(float) > ws_bytes) { set_error("conv1d_fwd: workspace too small"); return VQVAE_E_WORKSPACE; }
  const void* pre = (cam && skip_flag == nullptr) ? cam->packed : nullptr;      // packed ahead (vqvae_conv1d_pack)
  float* pk = pre ? (float*)pre : (float*)ws;
  const bool f16 = conv_f16x2(d) && skip_flag == nullptr && ws_bytes >= vqvae_conv1d_workspace_bytes(d);
  VQ_REQUIRE(!pre || f16 == conv_f16x2(d), "conv1d_fwd: workspace too small for the launch the packed slab was written for");
  unsigned* am = f16 ? conv_amax_slots(d, ws) : nullptr;
  unsigned* wam = !f16 ? nullptr : (pre ? reinterpret_cast<unsigned*>((char*)pre + conv_slab_bytes(d, 0)) : am + AMAX_SLOTS);
  const unsigned* xam = am;          // the operand's maximum: the caller's (it travelled with the tensor) or a scan
  if (f16 && cam && cam->x) xam = cam->x;
  else if (f16) {
    VQ_CHECK_HIP(hipMemsetAsync(am, 0, AMAX_SLOTS * sizeof(unsigned), st));
    if (int e = launch_absmax(x, (long)d->B * d->Cin * d->Tin, am, st)) return e;
  }
  if (!pre) {
    PackArgs pa; pa.njob = 1;
    pa.job[0] = pack_fwd_job(pk, W, d->Cout, d->Cin, d->K, 0, ldw, 0, ldw);
    pa.job[0].amax = wam;
    if (int e = launch_pack(pa, st, f16 ? 3 : -1)) return e;
  }
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.nseg = d->K;
  g.f16x2 = f16 ? 1 : 0;
  for (int j = 0; j < d->K; ++j) {
    Seg& sg = g.seg[j];
    sg.x = x; sg.x_bstride = (long)d->Cin * d->Tin; sg.x_cstride = d->Tin; sg.cin = d->Cin; sg.Tin = d->Tin;
    sg.tmul = d->stride; sg.toff = j * d->dil - d->pad; sg.tdiv = 1;
    sg.w = pk + (size_t)j * rp * ldw; sg.ldw = ldw;
    if (f16) { sg.amax = xam; sg.wamax = wam; }
  }
  g.M = d->Cout; g.Tout = d->Tout; g.B = d->B;
  g.out[0].y = y; g.out[0].y_bstride = (long)d->Cout * d->Tout; g.out[0].rows = d->Cout;
  g.out[0].bias = b; g.out[0].relu = d->relu;
  g.out[0].amax_out = cam ? cam->out : nullptr;
  g.skip_flag = skip_flag;
  {
    const size_t pkb = align_up((size_t)d->K * rp * ldw * sizeof(float), 256);
    const size_t need = ksplit_partial_floats(d->Cout, d->Tout, d->B, d->K * cdiv(d->Cin, BK)) * sizeof(float);
    if (need > 0 && pkb + need <= ws_bytes) g.partial = (float*)((char*)ws + pkb);
  }
  return launch_gemm<EPI_LINEAR>(g, VQVAE_PROF_CONV_FWD, st);
}

static int conv1d_bwd_data_impl(const vqvae_conv1d_desc* d, const float* W, const float* gy, float* gx, int accumulate,
                                void* ws, size_t ws_bytes, const vqvae_conv1d_amax* cam, vqvae_stream_t s, const float* x_relu = nullptr);
extern "C" int vqvae_conv1d_bwd_data_relu(const vqvae_conv1d_desc* d, const float* W, const float* gy, const float* x_relu,
                                          float* gx, void* ws, size_t ws_bytes, const vqvae_conv1d_amax* amax, vqvae_stream_t s) {
  VQ_REQUIRE(x_relu, "conv1d_bwd_data_relu: null x_relu");
  return conv1d_bwd_data_impl(d, W, gy, gx, 0, ws, ws_bytes, amax, s, x_relu);
}
extern "C" int vqvae_conv1d_bwd_data(const vqvae_conv1d_desc* d, const float* W, const float* gy,
                                     float* gx, int accumulate, void* ws, size_t ws_bytes,
                                     vqvae_stream_t s) {
  return conv1d_bwd_data_impl(d, W, gy, gx, accumulate, ws, ws_bytes, nullptr, s);
}
extern "C" int vqvae_conv1d_bwd_data_amax(const vqvae_conv1d_desc* d, const float* W, const float* gy,
                                          float* gx, int accumulate, void* ws, size_t ws_bytes,
                                          const vqvae_conv1d_amax* amax, vqvae_stream_t s) {
  return conv1d_bwd_data_impl(d, W, gy, gx, accumulate, ws, ws_bytes, amax, s);
}
static int conv1d_bwd_data_impl(const vqvae_conv1d_desc* d, const float* W, const float* gy, float* gx, int accumulate,
                                void* ws, size_t ws_bytes, const vqvae_conv1d_amax* cam, vqvae_stream_t s, const float* x_relu) {
  if (int e = check_conv_desc(d)) return e;
  VQ_REQUIRE(!x_relu || !accumulate, "conv1d_bwd_data_relu: the ReLU mask applies to a freshly written gx (accumulate == 0)");
  VQ_REQUIRE(W && gy && gx && ws, "conv1d_bwd_data: null pointer");
  hipStream_t st = (hipStream_t)s;
  const int ldw = pad128(d->Cin), rp = slab_rows(d->Cout);
  if ((size_t)d->K * rp * ldw * sizeof(float) > ws_bytes) { set_error("conv1d_bwd_data: workspace too small"); return VQVAE_E_WORKSPACE; }
  const void* pre = cam ? cam->packed : nullptr;      // packed ahead (vqvae_conv1d_pack, backward form)
  float* pk = pre ? (float*)pre : (float*)ws;
  const bool f16 = conv_f16x2(d) && ws_bytes >= vqvae_conv1d_workspace_bytes(d);
  VQ_REQUIRE(!pre || f16 == conv_f16x2(d), "conv1d_bwd_data: workspace too small for the launch the packed slab was written for");
  unsigned* am = f16 ? conv_amax_slots(d, ws) : nullptr;
  unsigned* wam = !f16 ? nullptr : (pre ? reinterpret_cast<unsigned*>((char*)pre + conv_slab_bytes(d, 1)) : am + AMAX_SLOTS);
  const unsigned* gam = am;
  if (f16 && cam && cam->gy) gam = cam->gy;
  else if (f16) {
    VQ_CHECK_HIP(hipMemsetAsync(am, 0, AMAX_SLOTS * sizeof(unsigned), st));
    if (int e = launch_absmax(gy, (long)d->B * d->Cout * d->Tout, am, st)) return e;
  }
  if (!pre) {
    PackArgs pa; pa.njob = 1;
    pa.job[0] = pack_bwd_job(pk, W, d->Cout, d->Cin, d->K, ldw);
    pa.job[0].amax = wam;
    if (int e = launch_pack(pa, st, f16 ? 3 : -1)) return e;
  }
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.nseg = d->K;
  g.f16x2 = f16 ? 1 : 0;
  for (int j = 0; j < d->K; ++j) {
    Seg& sg = g.seg[j];
    sg.x = gy; sg.x_bstride = (long)d->Cout * d->Tout; sg.x_cstride = d->Tout; sg.cin = d->Cout; sg.Tin = d->Tout;
    // t_out(gy) = (u + pad - j*dil) / stride
    sg.tmul = 1; sg.toff = d->pad - j * d->dil; sg.tdiv = d->stride;
    sg.w = pk + (size_t)j * rp * ldw; sg.ldw = ldw;
    if (f16) { sg.amax = gam; sg.wamax = wam; }
  }
  g.M = d->Cin; g.Tout = d->Tin; g.B = d->B;
  g.out[0].y = gx; g.out[0].y_bstride = (long)d->Cin * d->Tin; g.out[0].rows = d->Cin;
  g.out[0].accumulate = accumulate;
  g.out[0].amax_out = cam ? cam->out : nullptr;
  if (x_relu) { g.out[0].add = x_relu; g.out[0].add_bstride = (long)d->Cin * d->Tin; g.out[0].add_is_mask = 1; }
  {
    const size_t pkb = align_up((size_t)d->K * rp * ldw * sizeof(float), 256);
    const size_t need = ksplit_partial_floats(d->Cin, d->Tin, d->B, d->K * cdiv(d->Cout, BK)) * sizeof(float);
    if (need > 0 && pkb + need <= ws_bytes) g.partial = (float*)((char*)ws + pkb);
  }
  return launch_gemm<EPI_LINEAR>(g, VQVAE_PROF_CONV_BWD_DATA, st);
}

static int conv1d_bwd_weight_impl(const vqvae_conv1d_desc* d, const float* x, const float* gy, float* gW, float* gb,
                                  int accumulate, void* ws, size_t ws_bytes, const int32_t* skip_flag,
                                  const vqvae_conv1d_amax* cam, vqvae_stream_t s);
extern "C" int vqvae_conv1d_bwd_weight(const vqvae_conv1d_desc* d, const float* x, const float* gy,
                                       float* gW, float* gb, int accumulate, void* ws,
                                       size_t ws_bytes, vqvae_stream_t s) {
  return conv1d_bwd_weight_impl(d, x, gy, gW, gb, accumulate, ws, ws_bytes, nullptr, nullptr, s);
}
extern "C" int vqvae_conv1d_bwd_weight_cond(const vqvae_conv1d_desc* d, const float* x, const float* gy,
                                            float* gW, float* gb, int accumulate, void* ws,
                                            size_t ws_bytes, const int32_t* skip_flag,
                                            vqvae_stream_t s) {
  return conv1d_bwd_weight_impl(d, x, gy, gW, gb, accumulate, ws, ws_bytes, skip_flag, nullptr, s);
}
extern "C" int vqvae_conv1d_bwd_weight_amax(const vqvae_conv1d_desc* d, const float* x, const float* gy,
                                            float* gW, float* gb, int accumulate, void* ws,
                                            size_t ws_bytes, const vqvae_conv1d_amax* amax, vqvae_stream_t s) {
  return conv1d_bwd_weight_impl(d, x, gy, gW, gb, accumulate, ws, ws_bytes, nullptr, amax, s);
}
static int conv1d_bwd_weight_impl(const vqvae_conv1d_desc* d, const float* x, const float* gy, float* gW, float* gb,
                                  int accumulate, void* ws, size_t ws_bytes, const int32_t* skip_flag,
                                  const vqvae_conv1d_amax* cam, vqvae_stream_t s) {
  if (int e = check_conv_desc(d)) return e;
  VQ_REQUIRE(x && gy && gW && ws, "conv1d_bwd_weight: null pointer");
  hipStream_t st = (hipStream_t)s;
  int cins[MAXSEG];
  for (int i = 0; i < d->K; ++i) cins[i] = d->Cin;
  WgradPlan p = plan_wgrad(d->Cout, d->B, d->Tout, cins, d->K);
  if ((p.slab_floats + p.bslab_floats) * sizeof(float) > ws_bytes) { set_error("conv1d_bwd_weight: workspace too small"); return VQVAE_E_WORKSPACE; }
  WgradArgs w; memset(&w, 0, sizeof(w));
  w.gy = gy; w.gy_bstride = (long)d->Cout * d->Tout; w.M = d->Cout; w.Tout = d->Tout; w.B = d->B;
  w.nseg = d->K;
  const bool phases = phase_split_ok(d) && (p.slab_floats + p.bslab_floats + phase_split_floats(d)) * sizeof(float) <= ws_bytes;
  float* xe = nullptr;
  float* xo = nullptr;
  const int Tp = phase_pitch(d);
  if (phases) {
    xe = (float*)ws + ((p.slab_floats + p.bslab_floats + 63) / 64) * 64;
    xo = xe + (size_t)d->B * d->Cin * Tp;
    const long rows = (long)d->B * d->Cin;
    long nb = (rows * Tp + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(phase_split_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, rows, d->Tin, Tp, xe, xo, skip_flag);
    VQ_LAUNCH_CHECK();
  }
  for (int j = 0; j < d->K; ++j) {
    WSeg& sg = w.seg[j];
    const int e = j * d->dil - d->pad;
    if (phases) {
      const bool odd = (e & 1) != 0;
      sg.x = odd ? xo : xe; sg.x_bstride = (long)d->Cin * Tp; sg.x_cstride = Tp; sg.cin = d->Cin;
      sg.Tin = odd ? d->Tin / 2 : (d->Tin + 1) / 2;
      sg.tmul = 1; sg.toff = odd ? (e - 1) >> 1 : e >> 1; sg.tdiv = 1;       // (arithmetic shifts: floor for negative e)
    } else {
      sg.x = x; sg.x_bstride = (long)d->Cin * d->Tin; sg.x_cstride = d->Tin; sg.cin = d->Cin; sg.Tin = d->Tin;
      sg.tmul = d->stride; sg.toff = e; sg.tdiv = 1;
    }
    sg.gw = gW + j; sg.gw_co_stride = (long)d->Cin * d->K; sg.gw_ci_stride = d->K;
  }
  w.seg[0].gb = gb; w.accumulate = accumulate;
  w.skip_flag = skip_flag;
  if (conv_f16x2(d) && !phases && d->stride == 1 && skip_flag == nullptr && ws_bytes >= vqvae_conv1d_workspace_bytes(d)) {
    unsigned* am = conv_amax_slots(d, ws);
    const unsigned* xam = am;
    const unsigned* gam = am + AMAX_SLOTS;
    if (cam && cam->x) xam = cam->x;
    else {
      VQ_CHECK_HIP(hipMemsetAsync(am, 0, AMAX_SLOTS * sizeof(unsigned), st));
      if (int e = launch_absmax(x, (long)d->B * d->Cin * d->Tin, am, st)) return e;
    }
    if (cam && cam->gy) gam = cam->gy;
    else {
      VQ_CHECK_HIP(hipMemsetAsync(am + AMAX_SLOTS, 0, AMAX_SLOTS * sizeof(unsigned), st));
      if (int e = launch_absmax(gy, (long)d->B * d->Cout * d->Tout, am + AMAX_SLOTS, st)) return e;
    }
    w.f16x2 = 1; w.amax_gy = gam;
    for (int j = 0; j < d->K; ++j) w.seg[j].amax_x = xam;
  }
  return launch_wgrad(w, p, (float*)ws, VQVAE_PROF_CONV_WGRAD, st);
}

// ---------------------------------------------------------------------------
// C ABI: WaveNet ResidualBlock
// ---------------------------------------------------------------------------
namespace {
struct RbLayout {
  size_t gh, pk_d, pk_c, pk_o, pk_gz_r, pk_gz_s, pk_bd, pk_bc, hdr, slabs, total;   // float offsets
  WgradPlan p_h, p_r, p_s;
};

enum { HDR_D = 0, HDR_O = AMAX_SLOTS, HDR_GZ_R = 2 * AMAX_SLOTS, HDR_GZ_S = 3 * AMAX_SLOTS, HDR_BD = 4 * AMAX_SLOTS, HDR_N = 5,   // word offsets into RbLayout::hdr
       HDR_L1 = HDR_N * AMAX_SLOTS, HDR_L1_WORDS = 16 };       // wl1_kernel's three floats (pre-split bounds), behind the maxima

static RbLayout rb_layout(const vqvae_resblock_desc* d) {
  RbLayout L;
  const int Ch = d->Cd / 2;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 63) / 64 * 64; return r; };
  L.gh = take((size_t)d->B * d->Cd * d->T);
  L.pk_d = take((size_t)d->K * slab_rows(d->Cr) * pad128(d->Cd));      // fwd dilated conv
  L.pk_c = take((size_t)slab_rows(d->Cc) * pad128(d->Cd));             // fwd cond proj
  L.pk_o = take((size_t)slab_rows(Ch) * pad128(d->Cr + d->Cs));        // fwd res|skip
  L.pk_gz_r = take((size_t)slab_rows(d->Cr) * pad128(Ch));             // bwd gz from g_res
  L.pk_gz_s = take((size_t)slab_rows(d->Cs) * pad128(Ch));             // bwd gz from g_skip
  L.pk_bd = take((size_t)d->K * slab_rows(d->Cd) * pad128(d->Cr));     // bwd-data dilated conv
  L.pk_bc = take((size_t)slab_rows(d->Cd) * pad128(d->Cc));            // bwd-data cond proj
  L.hdr = take(HDR_N * AMAX_SLOTS + HDR_L1_WORDS);                     // float32x2: max |W| of each format-3 slab (HDR_* x AMAX_SLOTS words), then wl1_kernel's norms
  int cins[MAXSEG];
  for (int j = 0; j < d->K; ++j) cins[j] = d->Cr;
  cins[d->K] = d->Cc;
  L.p_h = plan_wgrad(d->Cd, d->B, d->T, cins, d->K + 1);
  int cz[1] = {Ch};
  L.p_r = plan_wgrad(d->Cr, d->B, d->T, cz, 1);
  L.p_s = plan_wgrad(d->Cs, d->B, d->T, cz, 1);
  size_t sl = L.p_h.slab_floats + L.p_h.bslab_floats;
  {   // without a condition tensor the same launch has K segments: its plan may need MORE splits
    WgradPlan pk = plan_wgrad(d->Cd, d->B, d->T, cins, d->K);
    if (pk.slab_floats + pk.bslab_floats > sl) sl = pk.slab_floats + pk.bslab_floats;
  }
  size_t s2 = L.p_r.slab_floats + L.p_r.bslab_floats;
  size_t s3 = L.p_s.slab_floats + L.p_s.bslab_floats;
  if (s2 > sl) sl = s2;
  if (s3 > sl) sl = s3;
  L.slabs = take(sl);
  L.total = o;
  return L;
}

// (The same predicate also stores the gate tensors [tanh | sigmoid] as bf16 in that mode -- GemmArgs::g16: there the
// backward pass differentiates the ROUNDED gates, which the oracle's bf16 mode mirrors: configs[4]'s "bf16" taken one
// step further than operand rounding, 63 MB less written and 63 MB less read per block.)
// z (B, Cd/2, T) is read only through GEMM staging (res 1x1, skip sum, res / skip weight gradients).  In matmul
// mode 1 that staging rounds it to bf16, so for the configs-sized blocks every producer and consumer agrees -- through
// this one predicate -- to keep it in HBM as bf16 (same element strides, the caller's buffer is simply half used):
// identical results, 31.5 MB less per launch that touches it at configs[4].
static bool z_bf16(const vqvae_resblock_desc* d) {
  return g_matmul_dtype == 1 && d->Cd / 2 == 128 && d->Cr == 256 && d->Cs % 256 == 0 && d->T % 64 == 0 &&
         (long)d->B * (d->Cr > d->Cs ? d->Cr : d->Cs) * d->T * 4 < (1L << 31);     // the bf16-reading kernels address with 32-bit offsets
}

static bool gates_bf16(const vqvae_resblock_desc* d) {
  return z_bf16(d);
}

// gh = [ga; gb] (B, Cd, T) stored as bf16 (vqvae_resblock_desc::storage & VQVAE_STORE_GH_BF16): the same blocks, on
// the caller's request.
static int bf16_storage_supported(const vqvae_resblock_desc* d) {
  int m = 0;
  if (gates_bf16(d) && d->K == 2 && d->Cd == 256) m |= VQVAE_STORE_GH_BF16;
  static const int lin128 = getenv("VQVAE_LIN128") ? atoi(getenv("VQVAE_LIN128")) : 32;
  // (the gate GEMM reads a bf16 x in its 256 x 128-tile two-tap form only: not offered when an A/B switch turns that form off)
  static const int lean = getenv("VQVAE_X3_LEAN") ? atoi(getenv("VQVAE_X3_LEAN")) : X3_LEAN;
  if (lin128 && lean && gates_bf16(d) && d->K == 2 && d->Cd == 256 && d->T % 128 == 0) m |= VQVAE_STORE_X_BF16 | VQVAE_STORE_RES_BF16;
  if (gates_bf16(d) && d->K == 2 && d->Cd == 256 && d->Cs == d->Cr && d->T % 128 == 0) m |= VQVAE_STORE_GX_BF16 | VQVAE_STORE_GRES_BF16;
  return m;
}

// matmul mode 3 (float32x2): the tensors of the packed chain that can be kept PRE-SPLIT (see presplit_pair) -- the
// configs-sized blocks whose two-tap GEMMs run the 256 x 128-tile loop and whose residual 1x1 runs the streaming kernel.
// vqvae_set_presplit(0) makes the library report none.
static int g_presplit = 7;           // (vqvae_set_presplit) bit 0: gh, bit 1: the residual stream, bit 2: sigmoid + z instead of tanh + sigmoid + z
static int f16x2_storage_supported(const vqvae_resblock_desc* d) {
  const int on = g_presplit;
  static const int lin128 = getenv("VQVAE_LIN128") ? atoi(getenv("VQVAE_LIN128")) : 32;
  static const int lean = getenv("VQVAE_X3_LEAN") ? atoi(getenv("VQVAE_X3_LEAN")) : X3_LEAN;
  if (!on || g_matmul_dtype != 3 || g_wgrad_impl == 1 || !lean) return 0;
  if (!(d->K == 2 && d->Cd == 256 && d->Cr == 256 && d->T % 128 == 0 && d->dil < d->T &&
        (long)d->B * d->Cd * d->T * 4 < (1L << 31))) return 0;
  int m = 0;
  if (on & 1) m |= VQVAE_STORE_GH_F16X2;
  if ((on & 2) && lin128) m |= VQVAE_STORE_X_F16X2 | VQVAE_STORE_RES_F16X2;
  if (on & 4) m |= VQVAE_STORE_GATES_SIG;
  return m;
}

static int check_rb(const vqvae_resblock_desc* d) {
  VQ_REQUIRE(d, "resblock: null desc");
  VQ_REQUIRE(d->storage == 0 || (d->storage & ~(bf16_storage_supported(d) | f16x2_storage_supported(d))) == 0,
             "resblock: desc.storage = %d asks for bf16 / pre-split tensors this shape / matmul mode does not keep (supported: %d)", d->storage, bf16_storage_supported(d) | f16x2_storage_supported(d));
  VQ_REQUIRE(d->B > 0 && d->T > 0 && d->Cr > 0 && d->Cd > 0 && d->Cs > 0 && d->Cc > 0, "resblock: bad dims");
  VQ_REQUIRE(d->Cd % 64 == 0, "resblock: dilated_channels/2 must be a multiple of 32 (got Cd=%d)", d->Cd);
  VQ_REQUIRE(d->K >= 1 && d->K <= MAXTAPS, "resblock: filter_size %d unsupported", d->K);
  VQ_REQUIRE(d->dil >= 1, "resblock: bad dilation");
  return 0;
}
}  // namespace

// What vqvae_resstack_pack wrote where: the slabs' layout depends on the matmul mode at pack time, and the kernels that
// read them are chosen by the mode at call time -- a vqvae_set_matmul_dtype between the two would make them read one
// format as another, silently.  Host-side tag per packed buffer, checked by every _packed entry point.
static std::mutex g_packed_mu;
static std::map<const char*, std::pair<size_t, int>> g_packed_fmt;       // base -> (bytes, matmul mode at pack time)
static void packed_register(const void* base, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_packed_mu);
  const char* b = (const char*)base;
  for (auto it = g_packed_fmt.begin(); it != g_packed_fmt.end();)          // drop stale overlapping entries (the pool reuses addresses)
    it = (it->first < b + bytes && b < it->first + it->second.first) ? g_packed_fmt.erase(it) : ++it;
  g_packed_fmt[b] = std::make_pair(bytes, g_matmul_dtype);
}
static int packed_check(const void* p) {
  std::lock_guard<std::mutex> lk(g_packed_mu);
  auto it = g_packed_fmt.upper_bound((const char*)p);
  VQ_REQUIRE(it != g_packed_fmt.begin(), "packed slabs %p were not written by vqvae_resstack_pack", p);
  --it;
  VQ_REQUIRE((const char*)p < it->first + it->second.first, "packed slabs %p were not written by vqvae_resstack_pack", p);
  VQ_REQUIRE(it->second.second == g_matmul_dtype, "packed slabs were written in matmul mode %d, used in mode %d", it->second.second, g_matmul_dtype);
  return 0;
}

extern "C" int vqvae_resblock_bf16_storage(const vqvae_resblock_desc* d) {
  if (!d || d->Cd <= 0) return 0;
  return bf16_storage_supported(d);
}
extern "C" int vqvae_set_presplit(int mask) {
  VQ_REQUIRE(mask >= 0 && mask <= 7, "set_presplit: bit 0 = gh, bit 1 = the residual stream, bit 2 = sigmoid + z instead of tanh + sigmoid + z");
  g_presplit = mask;
  return 0;
}
extern "C" int vqvae_resblock_f16x2_storage(const vqvae_resblock_desc* d) {
  if (!d || d->Cd <= 0) return 0;
  return f16x2_storage_supported(d);
}

extern "C" size_t vqvae_resblock_workspace_bytes(const vqvae_resblock_desc* d) {
  if (!d || d->Cd <= 0) return 0;
  return rb_layout(d).total * sizeof(float) + 256;
}

// `packed`: this block's weight slabs as vqvae_resstack_pack laid them out (the [pk_d, slabs) region of
// RbLayout), or NULL: pack into the workspace now.
static int resblock_fwd_impl(const vqvae_resblock_desc* d, const vqvae_resblock_params* p,
                             const float* x, const float* cond,
                             const vqvae_resblock_cproj* cproj, float* res, float* skip,
                             int skip_accumulate, float* gates, float* z, void* ws,
                             size_t ws_bytes, const float* packed, const vqvae_resblock_amax* am, vqvae_stream_t s) {
  if (int e = check_rb(d)) return e;
  // matmul mode 3: the packed chain runs the float32x2 kernels (vqvae_resstack_pack wrote format-3 slabs) and needs the
  // caller's maxima; the pack-per-call form keeps mode 2's kernels
  const bool f16 = g_matmul_dtype == 3 && packed != nullptr;
  if (f16) VQ_REQUIRE(am && am->x && (res == nullptr || am->res), "resblock_fwd_packed: matmul mode 3 needs amax->x (and amax->res with a residual output)");
  VQ_REQUIRE(p && x && (cond || cproj) && gates && z && ws, "resblock_fwd: null pointer");
  VQ_REQUIRE(p->Wd && (cproj || p->Wc) && (skip == nullptr || p->Ws) && (res == nullptr || p->Wr), "resblock_fwd: null weight");
  if (cproj) VQ_REQUIRE(cproj->P && cproj->v0 && cproj->w0 && cproj->w1 && cproj->Tl >= 2, "resblock_fwd: bad cproj");
  hipStream_t st = (hipStream_t)s;
  RbLayout L = rb_layout(d);
  if (L.total * sizeof(float) > ws_bytes) { set_error("resblock_fwd: workspace too small"); return VQVAE_E_WORKSPACE; }
  float* w = packed ? const_cast<float*>(packed) - L.pk_d : (float*)ws;       // only the pk_* offsets are used through w
  const int Ch = d->Cd / 2, T = d->T;
  const int ldd = pad128(d->Cd);
  const int Mo = (res ? d->Cr : 0) + (skip ? d->Cs : 0);
  const int ldo = pad128(Mo > 0 ? Mo : 1);
  if (packed) VQ_REQUIRE(cproj && !skip, "resblock_fwd_packed: the packed form serves ResidualNet's chain (latent-rate condition, no per-block skip)");
  const bool xpre = (d->storage & VQVAE_STORE_X_F16X2) != 0, rpre = res && (d->storage & VQVAE_STORE_RES_F16X2);
  if (d->storage & (VQVAE_STORE_X_F16X2 | VQVAE_STORE_RES_F16X2)) {
    VQ_REQUIRE(f16, "resblock_fwd: a pre-split residual stream (desc.storage) is kept by the packed float32x2 chain only");
    VQ_REQUIRE(!xpre || res == nullptr || rpre, "resblock_fwd: a pre-split x with an fp32 residual output is not built");
    VQ_REQUIRE(!rpre || (am->x_max && am->res_scale), "resblock_fwd: a pre-split residual output needs amax->x_max and amax->res_scale");
  }
  if (d->storage & (VQVAE_STORE_X_BF16 | VQVAE_STORE_RES_BF16)) {
    VQ_REQUIRE(packed, "resblock_fwd: a bf16 residual stream (desc.storage) is kept by the packed chain form only");
    VQ_REQUIRE(!(d->storage & VQVAE_STORE_X_BF16) || res == nullptr || (d->storage & VQVAE_STORE_RES_BF16), "resblock_fwd: a bf16 x with an fp32 residual output is not built");
  }

  PackArgs pa; pa.njob = 0;
  if (!packed) {
  pa.job[pa.njob++] = pack_fwd_job(w + L.pk_d, p->Wd, d->Cd, d->Cr, d->K, Ch, ldd, 0, ldd);
  if (!cproj) pa.job[pa.njob++] = pack_fwd_job(w + L.pk_c, p->Wc, d->Cd, d->Cc, 1, Ch, ldd, 0, ldd);
  if (res && skip) {
    VQ_REQUIRE(d->Cr % 32 == 0, "resblock: residual_channels must be a multiple of 32 when res and skip share a launch");
    pa.job[pa.njob++] = pack_fwd_job(w + L.pk_o, p->Wr, d->Cr, Ch, 1, 0, ldo, 0, d->Cr);
    pa.job[pa.njob++] = pack_fwd_job(w + L.pk_o, p->Ws, d->Cs, Ch, 1, 0, ldo, d->Cr, ldo - d->Cr);
  } else if (res) {
    pa.job[pa.njob++] = pack_fwd_job(w + L.pk_o, p->Wr, d->Cr, Ch, 1, 0, ldo, 0, ldo);
  } else if (skip) {
    pa.job[pa.njob++] = pack_fwd_job(w + L.pk_o, p->Ws, d->Cs, Ch, 1, 0, ldo, 0, ldo);
  }
  if (int e = launch_pack(pa, st)) return e;
  }

  // K1: h = dilconv(x) + cond_proj(c) + biases -> gate
  {
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.nseg = d->K + (cproj ? 0 : 1);
    const int rp = slab_rows(d->Cr);
    for (int j = 0; j < d->K; ++j) {
      Seg& sg = g.seg[j];
      sg.x = x; sg.x_bstride = (long)d->Cr * T; sg.x_cstride = T; sg.cin = d->Cr; sg.Tin = T;
      sg.tmul = 1; sg.toff = -(d->K - 1 - j) * d->dil; sg.tdiv = 1;
      sg.w = w + L.pk_d + (size_t)j * rp * ldd; sg.ldw = ldd;
      if (f16) { sg.amax = am->x; sg.wamax = reinterpret_cast<const unsigned*>(w + L.hdr) + HDR_D; }
    }
    g.f16x2 = f16 ? 1 : 0;
    if (cproj) {      // condition projection (incl. its bias) arrives pre-computed at latent rate
      g.lerp.P = cproj->P; g.lerp.p_bstride = cproj->P_bstride; g.lerp.Tl = cproj->Tl;
      g.lerp.v0 = cproj->v0; g.lerp.w0 = cproj->w0; g.lerp.w1 = cproj->w1;
      g.lerp.fold = 1; g.lerp.amax = cproj->P_amax;       // a request: launch_gemm decides (kernel form, shape)
    } else {
      Seg& sc = g.seg[d->K];
      sc.x = cond; sc.x_bstride = (long)d->Cc * T; sc.x_cstride = T; sc.cin = d->Cc; sc.Tin = T;
      sc.tmul = 1; sc.toff = 0; sc.tdiv = 1; sc.w = w + L.pk_c; sc.ldw = ldd;
    }
    g.M = d->Cd; g.Tout = T; g.B = d->B;
    g.out[0].y = gates; g.out[0].y_bstride = (long)d->Cd * T; g.out[0].bias = (cproj && cproj->P_has_bd) ? nullptr : p->bd;
    g.out[0].bias2 = cproj ? nullptr : p->bc;
    g.out[0].rows = d->Cd;
    g.out[1].y = z; g.out[1].y_bstride = (long)Ch * T;
    g.z16 = z_bf16(d) ? 1 : 0;
    g.g16 = gates_bf16(d) ? 1 : 0;
    g.gsig = (d->storage & VQVAE_STORE_GATES_SIG) ? 1 : 0;
    g.x16 = (d->storage & VQVAE_STORE_X_BF16) ? 3 : 0;
    if (xpre) g.x16 = 3;                         // both taps read the pre-split x_l; amax->x holds its scale words
    if (int e = launch_gemm<EPI_GATE>(g, VQVAE_PROF_RESBLOCK_GATE, st)) return e;
  }
  // K2: [res; skip] = [Wr; Ws] z (+ x) (+= skip)
  if (Mo > 0) {
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.nseg = 1;
    Seg& sg = g.seg[0];
    sg.x = z; sg.x_bstride = (long)Ch * T; sg.x_cstride = T; sg.cin = Ch; sg.Tin = T;
    sg.tmul = 1; sg.toff = 0; sg.tdiv = 1; sg.w = w + L.pk_o; sg.ldw = ldo;
    if (f16) { sg.amax_static = 1.f; sg.wamax = reinterpret_cast<const unsigned*>(w + L.hdr) + HDR_O; g.f16x2 = 1; }   // |z| = |tanh * sigmoid| <= 1
    g.M = Mo; g.Tout = T; g.B = d->B;
    int o = 0;
    if (res) {
      g.out[0].y = res; g.out[0].y_bstride = (long)d->Cr * T; g.out[0].rows = d->Cr;
      g.out[0].add = x; g.out[0].add_bstride = (long)d->Cr * T; g.out[0].bias = p->br;
      g.out[0].amax_out = am ? am->res : nullptr;
      o = 1;
    }
    if (skip) {
      g.out[o].y = skip; g.out[o].y_bstride = (long)d->Cs * T; g.out[o].rows = d->Cs;
      g.out[o].bias = p->bs; g.out[o].accumulate = skip_accumulate;
    }
    g.z16 = z_bf16(d) ? 1 : 0;
    g.add16 = (res && (d->storage & VQVAE_STORE_X_BF16)) ? 1 : 0;
    g.y16 = (res && (d->storage & VQVAE_STORE_RES_BF16)) ? 1 : 0;
    if (rpre) {
      g.add16 = xpre ? 1 : 0; g.y16 = 1;
      g.add_scale = am->x; g.add_amax = am->x_max; g.scale_out = am->res_scale;
      g.bound_l1 = w + L.hdr + HDR_L1;
      if (cproj && cproj->P_amax) {          // the next block's gate GEMM adds its condition as a K step: leave it room
        g.floor_p = cproj->P_amax;
        g.floor_w = reinterpret_cast<const unsigned*>(w + (L.slabs - L.pk_d) + L.hdr) + HDR_D;      // the next block's slice of the packed slabs
      }
    }
    if (g.add16 || g.y16) VQ_REQUIRE(!skip, "resblock_fwd: a bf16 residual stream (desc.storage) is served by the chain form only (no per-block skip output)");
    if (int e = launch_gemm<EPI_LINEAR>(g, VQVAE_PROF_RESBLOCK_OUT, st)) return e;
  }
  return 0;
}

extern "C" int vqvae_resblock_fwd(const vqvae_resblock_desc* d, const vqvae_resblock_params* p,
                                  const float* x, const float* cond,
                                  const vqvae_resblock_cproj* cproj, float* res, float* skip,
                                  int skip_accumulate, float* gates, float* z, void* ws,
                                  size_t ws_bytes, vqvae_stream_t s) {
  return resblock_fwd_impl(d, p, x, cond, cproj, res, skip, skip_accumulate, gates, z, ws, ws_bytes, nullptr, nullptr, s);
}

extern "C" int vqvae_resblock_fwd_packed(const vqvae_resblock_desc* d, const vqvae_resblock_params* p,
                                         const float* x, const vqvae_resblock_cproj* cproj, float* res,
                                         float* gates, float* z, void* ws, size_t ws_bytes,
                                         const void* packed, const vqvae_resblock_amax* amax, vqvae_stream_t s) {
  VQ_REQUIRE(packed, "resblock_fwd_packed: null packed slabs");
  if (int e = packed_check(packed)) return e;
  return resblock_fwd_impl(d, p, x, nullptr, cproj, res, nullptr, 0, gates, z, ws, ws_bytes, (const float*)packed, amax, s);
}

extern "C" size_t vqvae_resstack_packed_bytes(const vqvae_resblock_desc* d) {
  if (!d || d->Cd <= 0) return 0;
  const RbLayout L = rb_layout(d);
  return (L.slabs - L.pk_d) * sizeof(float);
}

// Every weight slab the chain of ResidualNet needs for one training step -- forward (gated dilated conv,
// res 1x1) and backward (gz from g_res / g_skip, dilated-conv backward-data) of every block -- re-laid in
// ceil(5 nblocks / 24) launches, once per step: the weights only change in the optimizer.  (Round 2
// re-packed inside every resblock_fwd / resblock_bwd call: 74 launches per configs[1] step.)
extern "C" int vqvae_resstack_pack(const vqvae_resblock_desc* d, int nblocks,
                                   const vqvae_resblock_params* params, const int* has_res,
                                   void* packed, size_t packed_bytes, vqvae_stream_t s) {
  if (int e = check_rb(d)) return e;
  VQ_REQUIRE(nblocks >= 1 && params && has_res && packed, "resstack_pack: null pointer / no blocks");
  const RbLayout L = rb_layout(d);
  const size_t per = L.slabs - L.pk_d;
  if (per * sizeof(float) * nblocks > packed_bytes) { set_error("resstack_pack: packed buffer too small"); return VQVAE_E_WORKSPACE; }
  hipStream_t st = (hipStream_t)s;
  const int Ch = d->Cd / 2;
  const int ldd = pad128(d->Cd), ldz = pad128(Ch), ldr = pad128(d->Cr);
  const bool f16 = g_matmul_dtype == 3;      // float32x2 chain: two scaled fp16 pieces, max |W| of every slab in the block's header
  packed_register(packed, per * sizeof(float) * nblocks);
  PackArgs pa; pa.njob = 0;
  auto flush = [&]() -> int { if (pa.njob == 0) return 0; const int e = launch_pack(pa, st, f16 ? 3 : -1); pa.njob = 0; return e; };
  auto add = [&](PackJob j, float* w, int slot) { j.amax = f16 ? reinterpret_cast<unsigned*>(w + L.hdr) + slot : nullptr; pa.job[pa.njob++] = j; };
  for (int l = 0; l < nblocks; ++l) {
    float* w = (float*)packed + (size_t)l * per - L.pk_d;
    const vqvae_resblock_params& p = params[l];
    VQ_REQUIRE(p.Wd && p.Ws && (!has_res[l] || p.Wr), "resstack_pack: null weight in block %d", l);
    if (pa.njob + 5 > MAXSEG) { if (int e = flush()) return e; }
    add(pack_fwd_job(w + L.pk_d, p.Wd, d->Cd, d->Cr, d->K, Ch, ldd, 0, ldd), w, HDR_D);
    if (has_res[l]) {
      const int ldo = pad128(d->Cr);
      add(pack_fwd_job(w + L.pk_o, p.Wr, d->Cr, Ch, 1, 0, ldo, 0, ldo), w, HDR_O);
      add(pack_bwd_job(w + L.pk_gz_r, p.Wr, d->Cr, Ch, 1, ldz), w, HDR_GZ_R);
    }
    add(pack_bwd_job(w + L.pk_gz_s, p.Ws, d->Cs, Ch, 1, ldz), w, HDR_GZ_S);
    add(pack_bwd_job(w + L.pk_bd, p.Wd, d->Cd, d->Cr, d->K, ldr), w, HDR_BD);
  }
  if (int e = flush()) return e;
  if (f16 && f16x2_storage_supported(d) && Ch <= 256) {       // the weight norms behind the pre-split tensors' bounds
    for (int l0 = 0; l0 < nblocks; l0 += MAXSEG) {
      L1Args la; memset(&la, 0, sizeof(la));
      la.Cr = d->Cr; la.Cs = d->Cs; la.Ch = Ch;
      const int n = nblocks - l0 < MAXSEG ? nblocks - l0 : MAXSEG;
      for (int i = 0; i < n; ++i) {
        const vqvae_resblock_params& p = params[l0 + i];
        float* w = (float*)packed + (size_t)(l0 + i) * per - L.pk_d;
        la.job[i].Wr = has_res[l0 + i] ? p.Wr : nullptr; la.job[i].br = has_res[l0 + i] ? p.br : nullptr;
        la.job[i].Ws = p.Ws; la.job[i].out = w + L.hdr + HDR_L1;
      }
      hipLaunchKernelGGL(wl1_kernel, dim3(n, 3), dim3(256), 0, st, la);
      VQ_LAUNCH_CHECK();
    }
  }
  return 0;
}

static int resblock_bwd_impl(const vqvae_resblock_desc* d, const vqvae_resblock_params* p,
                                  const float* x, const float* cond, const float* gates,
                                  const float* z, const float* g_res, const float* g_skip,
                                  float* gx, float* gcond, int gcond_accumulate, float* gh_out,
                                  const vqvae_resblock_grads* gr, int grads_accumulate, void* ws,
                                  size_t ws_bytes, const float* packed, const vqvae_resblock_amax* am, vqvae_stream_t s) {
  if (int e = check_rb(d)) return e;
  const bool f16 = g_matmul_dtype == 3 && packed != nullptr;      // see resblock_fwd_impl
  if (f16) VQ_REQUIRE(am && am->g_skip && am->gh && (g_res == nullptr || am->g_res), "resblock_bwd_packed: matmul mode 3 needs amax->g_skip, amax->gh (and amax->g_res with a residual gradient)");
  VQ_REQUIRE(p && x && gates && z && g_skip && ws && gr, "resblock_bwd: null pointer");
  VQ_REQUIRE(cond || (!gcond && !gr->gWc && !gr->gbc), "resblock_bwd: condition gradients requested without a condition tensor");
  hipStream_t st = (hipStream_t)s;
  RbLayout L = rb_layout(d);
  if (L.total * sizeof(float) > ws_bytes) { set_error("resblock_bwd: workspace too small"); return VQVAE_E_WORKSPACE; }
  float* w = (float*)ws;
  const int Ch = d->Cd / 2, T = d->T;
  float* gh = gh_out ? gh_out : w + L.gh;
  const int ldz = pad128(Ch), ldr = pad128(d->Cr), ldc = pad128(d->Cc);
  // packed slabs (vqvae_resstack_pack): only the pk_* offsets are read through wpk
  float* wpk = packed ? const_cast<float*>(packed) - L.pk_d : w;
  if (packed) VQ_REQUIRE(!gcond && gh_out, "resblock_bwd_packed: the packed form serves ResidualNet's chain (no per-block condition gradient, gh kept)");
  const bool h16 = (d->storage & VQVAE_STORE_GH_BF16) != 0;
  const bool gres16 = g_res && (d->storage & VQVAE_STORE_GRES_BF16), gx16 = gx && (d->storage & VQVAE_STORE_GX_BF16);
  if (h16 || (d->storage & (VQVAE_STORE_GRES_BF16 | VQVAE_STORE_GX_BF16)))
    VQ_REQUIRE(packed, "resblock_bwd: bf16-stored gh / gradient stream (desc.storage) are kept by the packed chain form only");
  const bool hpre = (d->storage & VQVAE_STORE_GH_F16X2) != 0;
  if (hpre) VQ_REQUIRE(f16 && am->gh_scale, "resblock_bwd: a pre-split gh (desc.storage) is kept by the packed float32x2 chain only and needs amax->gh_scale");

  if (!packed) {
  PackArgs pa; pa.njob = 0;
  if (g_res) pa.job[pa.njob++] = pack_bwd_job(w + L.pk_gz_r, p->Wr, d->Cr, Ch, 1, ldz);
  pa.job[pa.njob++] = pack_bwd_job(w + L.pk_gz_s, p->Ws, d->Cs, Ch, 1, ldz);
  if (gx) pa.job[pa.njob++] = pack_bwd_job(w + L.pk_bd, p->Wd, d->Cd, d->Cr, d->K, ldr);
  if (gcond) pa.job[pa.njob++] = pack_bwd_job(w + L.pk_bc, p->Wc, d->Cd, d->Cc, 1, ldc);
  if (int e = launch_pack(pa, st)) return e;
  }

  // K3: gz = Wr^T g_res + Ws^T g_skip ; gh = gate'(gz)
  {
    GemmArgs g; memset(&g, 0, sizeof(g));
    int n = 0;
    if (g_res) {
      Seg& sg = g.seg[n++];
      sg.x = g_res; sg.x_bstride = (long)d->Cr * T; sg.x_cstride = T; sg.cin = d->Cr; sg.Tin = T;
      sg.tmul = 1; sg.toff = 0; sg.tdiv = 1; sg.w = wpk + L.pk_gz_r; sg.ldw = ldz;
      if (f16) { sg.amax = am->g_res; sg.wamax = reinterpret_cast<const unsigned*>(wpk + L.hdr) + HDR_GZ_R; }
    }
    Seg& ss = g.seg[n++];
    ss.x = g_skip; ss.x_bstride = (long)d->Cs * T; ss.x_cstride = T; ss.cin = d->Cs; ss.Tin = T;
    ss.tmul = 1; ss.toff = 0; ss.tdiv = 1; ss.w = wpk + L.pk_gz_s; ss.ldw = ldz;
    if (f16) { ss.amax = am->g_skip; ss.wamax = reinterpret_cast<const unsigned*>(wpk + L.hdr) + HDR_GZ_S; }
    g.f16x2 = f16 ? 1 : 0;
    g.nseg = n;
    g.M = Ch; g.Tout = T; g.B = d->B;
    g.out[0].y = gh; g.out[0].y_bstride = (long)d->Cd * T; g.out[0].rows = Ch;
    g.out[0].add = gates; g.out[0].add_bstride = (long)d->Cd * T;
    g.out[0].amax_out = am ? am->gh : nullptr;
    g.g16 = gates_bf16(d) ? 1 : 0;
    if (d->storage & VQVAE_STORE_GATES_SIG) { g.gsig = 1; g.zsrc = z; }
    g.h16 = (h16 || hpre) ? 1 : 0;
    g.x16 = gres16 ? 1 : 0;                       // segment 0 = g_res
    if (f16 && am->pb_part) {                     // the latent pull-back of gh in this launch's epilogue
      g.pb_part = am->pb_part; g.lerp.v0 = am->pb_v0; g.lerp.w0 = am->pb_w0; g.lerp.w1 = am->pb_w1; g.lerp.Tl = am->pb_Tl;
    }
    if (hpre) {                                   // bound: out[1] max|g_res| + out[2] max|g_skip| (wl1_kernel), segment order [g_res,] g_skip
      g.bound_l1 = wpk + L.hdr + HDR_L1 + (g_res ? 1 : 2);
      g.scale_out = am->gh_scale;
    }
    if (int e = launch_gemm<EPI_GATE_BWD>(g, VQVAE_PROF_RESBLOCK_BWD_GZ, st)) return e;
  }
  // K4: gx = g_res + sum_j Wd_j^T gh[t + (K-1-j) dil]
  if (gx) {
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.nseg = d->K;
    const int rp = slab_rows(d->Cd);
    for (int j = 0; j < d->K; ++j) {
      Seg& sg = g.seg[j];
      sg.x = gh; sg.x_bstride = (long)d->Cd * T; sg.x_cstride = T; sg.cin = d->Cd; sg.Tin = T;
      sg.tmul = 1; sg.toff = (d->K - 1 - j) * d->dil; sg.tdiv = 1;
      sg.w = wpk + L.pk_bd + (size_t)j * rp * ldr; sg.ldw = ldr;
      if (f16) { sg.amax = hpre ? am->gh_scale : am->gh; sg.wamax = reinterpret_cast<const unsigned*>(wpk + L.hdr) + HDR_BD; }
    }
    g.f16x2 = f16 ? 1 : 0;
    g.M = d->Cr; g.Tout = T; g.B = d->B;
    g.out[0].y = gx; g.out[0].y_bstride = (long)d->Cr * T; g.out[0].rows = d->Cr;
    g.out[0].add = g_res; g.out[0].add_bstride = (long)d->Cr * T;
    g.out[0].amax_out = am ? am->gx : nullptr;
    g.x16 = (h16 || hpre) ? 3 : 0;
    g.add16 = gres16 ? 1 : 0; g.y16 = gx16 ? 1 : 0;
    if (int e = launch_gemm<EPI_LINEAR>(g, VQVAE_PROF_RESBLOCK_BWD_GX, st)) return e;
  }
  // K5: gcond (+)= Wc^T gh
  if (gcond) {
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.nseg = 1;
    Seg& sg = g.seg[0];
    sg.x = gh; sg.x_bstride = (long)d->Cd * T; sg.x_cstride = T; sg.cin = d->Cd; sg.Tin = T;
    sg.tmul = 1; sg.toff = 0; sg.tdiv = 1; sg.w = w + L.pk_bc; sg.ldw = ldc;
    g.M = d->Cc; g.Tout = T; g.B = d->B;
    g.out[0].y = gcond; g.out[0].y_bstride = (long)d->Cc * T; g.out[0].rows = d->Cc;
    g.out[0].accumulate = gcond_accumulate;
    if (int e = launch_gemm<EPI_LINEAR>(g, VQVAE_PROF_RESBLOCK_BWD_GC, st)) return e;
  }
  // K6a: gWd, gWc, gbd, gbc from gh
  if (gr->gWd || gr->gWc || gr->gbd || gr->gbc) {
    WgradArgs wa; memset(&wa, 0, sizeof(wa));
    wa.gy = gh; wa.gy_bstride = (long)d->Cd * T; wa.M = d->Cd; wa.Tout = T; wa.B = d->B;
    wa.nseg = d->K + (cond ? 1 : 0);
    for (int j = 0; j < d->K; ++j) {
      WSeg& sg = wa.seg[j];
      sg.x = x; sg.x_bstride = (long)d->Cr * T; sg.x_cstride = T; sg.cin = d->Cr; sg.Tin = T;
      sg.tmul = 1; sg.toff = -(d->K - 1 - j) * d->dil; sg.tdiv = 1;
      sg.gw = gr->gWd ? gr->gWd + j : nullptr; sg.gw_co_stride = (long)d->Cr * d->K; sg.gw_ci_stride = d->K;
    }
    if (cond) {
      WSeg& sc = wa.seg[d->K];
      sc.x = cond; sc.x_bstride = (long)d->Cc * T; sc.x_cstride = T; sc.cin = d->Cc; sc.Tin = T;
      sc.tmul = 1; sc.toff = 0; sc.tdiv = 1;
      sc.gw = gr->gWc; sc.gw_co_stride = d->Cc; sc.gw_ci_stride = 1;
    }
    wa.seg[0].gb = gr->gbd; wa.seg[0].gb2 = gr->gbc; wa.accumulate = grads_accumulate;
    int cins2[MAXTAPS + 1];
    for (int j = 0; j < d->K; ++j) cins2[j] = d->Cr;
    cins2[d->K] = d->Cc;
    WgradPlan ph = cond ? L.p_h : plan_wgrad(d->Cd, d->B, d->T, cins2, d->K);
    if (int e = launch_wgrad(wa, ph, w + L.slabs, VQVAE_PROF_RESBLOCK_WGRAD, st)) return e;
  }
  // K6b / K6c: gWr, gbr from g_res ; gWs, gbs from g_skip
  for (int which = 0; which < 2; ++which) {
    const float* gy = which ? g_skip : g_res;
    float* gW = which ? gr->gWs : gr->gWr;
    float* gb = which ? gr->gbs : gr->gbr;
    const int M = which ? d->Cs : d->Cr;
    if (!gy || (!gW && !gb)) continue;
    WgradArgs wa; memset(&wa, 0, sizeof(wa));
    wa.gy = gy; wa.gy_bstride = (long)M * T; wa.M = M; wa.Tout = T; wa.B = d->B;
    wa.nseg = 1;
    WSeg& sg = wa.seg[0];
    sg.x = z; sg.x_bstride = (long)Ch * T; sg.x_cstride = T; sg.cin = Ch; sg.Tin = T;
    sg.tmul = 1; sg.toff = 0; sg.tdiv = 1;
    sg.gw = gW; sg.gw_co_stride = Ch; sg.gw_ci_stride = 1;
    wa.seg[0].gb = gb; wa.accumulate = grads_accumulate;
    wa.x16 = z_bf16(d) ? 1 : 0;
    if (int e = launch_wgrad(wa, which ? L.p_s : L.p_r, w + L.slabs, VQVAE_PROF_RESBLOCK_WGRAD, st)) return e;
  }
  return 0;
}

extern "C" int vqvae_resblock_bwd(const vqvae_resblock_desc* d, const vqvae_resblock_params* p,
                                  const float* x, const float* cond, const float* gates,
                                  const float* z, const float* g_res, const float* g_skip,
                                  float* gx, float* gcond, int gcond_accumulate, float* gh_out,
                                  const vqvae_resblock_grads* gr, int grads_accumulate, void* ws,
                                  size_t ws_bytes, vqvae_stream_t s) {
  return resblock_bwd_impl(d, p, x, cond, gates, z, g_res, g_skip, gx, gcond, gcond_accumulate, gh_out, gr,
                           grads_accumulate, ws, ws_bytes, nullptr, nullptr, s);
}

// the chain part of a block's backward (gz, gate derivative -> gh_out, backward-data -> gx) on packed slabs
extern "C" int vqvae_resblock_bwd_packed(const vqvae_resblock_desc* d, const vqvae_resblock_params* p,
                                         const float* x, const float* gates, const float* z,
                                         const float* g_res, const float* g_skip, float* gx, float* gh_out,
                                         void* ws, size_t ws_bytes, const void* packed,
                                         const vqvae_resblock_amax* amax, vqvae_stream_t s) {
  VQ_REQUIRE(packed, "resblock_bwd_packed: null packed slabs");
  if (int e = packed_check(packed)) return e;
  vqvae_resblock_grads none;
  memset(&none, 0, sizeof(none));
  return resblock_bwd_impl(d, p, x, nullptr, gates, z, g_res, g_skip, gx, nullptr, 0, gh_out, &none, 0, ws, ws_bytes,
                           (const float*)packed, amax, s);
}

// ---------------------------------------------------------------------------
// C ABI: ResidualNet-level contractions (WaveNet/modules.py:89-96).
//
// skip_connections = sum_l skip_l(z_l) is ONE GEMM over K = nblocks * Cd/2
// (the z_l of all blocks are kept in HBM for backward anyway), instead of
// nblocks read-modify-write passes over the (B,Cs,T) accumulator; likewise the
// condition gradient sum_l Wc_l^T gh_l is one GEMM over K = nblocks * Cd, and
// the skip-weight gradients share one launch (g_skip is their common operand).
// ---------------------------------------------------------------------------
struct PtrList { const float* p[MAXSEG]; };
__global__ void bias_sum_list_kernel(const PtrList bl, int nb, int n, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
  for (int l = 0; l < nb; ++l) v += bl.p[l][i];
  out[i] = v;
}

extern "C" size_t vqvae_resstack_workspace_bytes(const vqvae_resblock_desc* d, int nblocks) {
  if (!d || nblocks < 1 || nblocks > MAXSEG) return 0;
  const int Ch = d->Cd / 2;
  size_t skip_pk = (size_t)nblocks * slab_rows(Ch) * pad128(d->Cs) + pad128(d->Cs);
  size_t gc_pk = (size_t)nblocks * slab_rows(d->Cd) * pad128(d->Cc);
  int cz[MAXSEG];
  for (int i = 0; i < nblocks; ++i) cz[i] = Ch;
  // the res-conv gradients skip blocks without a residual output (the last one), and the split
  // count -- hence the slab volume -- is not monotonic in the segment count: cover every count
  size_t wg = 0;
  for (int n = 1; n <= nblocks; ++n) {
    WgradPlan p = plan_wgrad(d->Cs, d->B, d->T, cz, n);
    WgradPlan p2 = plan_wgrad(d->Cr, d->B, d->T, cz, n);
    if (p.slab_floats + p.bslab_floats > wg) wg = p.slab_floats + p.bslab_floats;
    if (p2.slab_floats + p2.bslab_floats > wg) wg = p2.slab_floats + p2.bslab_floats;
  }
  size_t m = skip_pk > gc_pk ? skip_pk : gc_pk;
  if (wg > m) m = wg;
  return m * sizeof(float) + 1024 + MAXSEG * AMAX_SLOTS * sizeof(unsigned);     // (+ the weights' maxima of a float32x2 skip sum)
}

// stage: 1 = pack the slabs / sum the biases into ws only (vqvae_resstack_skip_prepare), 2 = the GEMM only, over a ws that was
// prepared (vqvae_resstack_skip_fwd_prepared), 3 = both (vqvae_resstack_skip_fwd)
static int resstack_skip_impl(const vqvae_resblock_desc* d, int nblocks, const float* const* Ws, const float* const* bs,
                              const float* const* z, float* skip, int accumulate, int relu, void* ws, size_t ws_bytes,
                              uint32_t* skip_amax_out, vqvae_stream_t s, int stage);
extern "C" int vqvae_resstack_skip_fwd(const vqvae_resblock_desc* d, int nblocks,
                                       const float* const* Ws, const float* const* bs,
                                       const float* const* z, float* skip, int accumulate, int relu,
                                       void* ws, size_t ws_bytes, uint32_t* skip_amax_out, vqvae_stream_t s) {
  VQ_REQUIRE(Ws && bs && z && skip, "resstack_skip_fwd: null pointer");
  return resstack_skip_impl(d, nblocks, Ws, bs, z, skip, accumulate, relu, ws, ws_bytes, skip_amax_out, s, 3);
}
extern "C" int vqvae_resstack_skip_prepare(const vqvae_resblock_desc* d, int nblocks, const float* const* Ws,
                                           const float* const* bs, void* ws, size_t ws_bytes, vqvae_stream_t s) {
  VQ_REQUIRE(Ws && bs, "resstack_skip_prepare: null pointer");
  return resstack_skip_impl(d, nblocks, Ws, bs, nullptr, nullptr, 0, 0, ws, ws_bytes, nullptr, s, 1);
}
extern "C" int vqvae_resstack_skip_fwd_prepared(const vqvae_resblock_desc* d, int nblocks, const float* const* z, float* skip,
                                                int accumulate, int relu, const void* ws, size_t ws_bytes,
                                                uint32_t* skip_amax_out, vqvae_stream_t s) {
  VQ_REQUIRE(z && skip, "resstack_skip_fwd_prepared: null pointer");
  return resstack_skip_impl(d, nblocks, nullptr, nullptr, z, skip, accumulate, relu, const_cast<void*>(ws), ws_bytes, skip_amax_out, s, 2);
}
static int resstack_skip_impl(const vqvae_resblock_desc* d, int nblocks, const float* const* Ws, const float* const* bs,
                              const float* const* z, float* skip, int accumulate, int relu, void* ws, size_t ws_bytes,
                              uint32_t* skip_amax_out, vqvae_stream_t s, int stage) {
  if (int e = check_rb(d)) return e;
  VQ_REQUIRE(nblocks >= 1 && nblocks <= MAXSEG, "resstack_skip_fwd: 1..%d blocks", MAXSEG);
  VQ_REQUIRE(ws, "resstack_skip_fwd: null pointer");
  if (ws_bytes < vqvae_resstack_workspace_bytes(d, nblocks)) { set_error("resstack_skip_fwd: workspace too small"); return VQVAE_E_WORKSPACE; }
  hipStream_t st = (hipStream_t)s;
  const int Ch = d->Cd / 2, T = d->T;
  const int ld = pad128(d->Cs), rp = slab_rows(Ch);
  float* w = (float*)ws;
  float* bsum = w + (size_t)nblocks * rp * ld;
  // matmul mode 3: float32x2 -- the operand is the gate output, |z| <= 1 by construction, so only the weights' maxima
  // (one per block, behind the bias sums) have to be found
  const bool f16 = g_matmul_dtype == 3;
  unsigned* wam = reinterpret_cast<unsigned*>(bsum + pad128(d->Cs));
  if (stage & 1) {
    PackArgs pa; pa.njob = 0;
    PtrList bl;
    for (int l = 0; l < nblocks; ++l) {
      pa.job[pa.njob] = pack_fwd_job(w + (size_t)l * rp * ld, Ws[l], d->Cs, Ch, 1, 0, ld, 0, ld);
      pa.job[pa.njob++].amax = f16 ? wam + l * AMAX_SLOTS : nullptr;
      bl.p[l] = bs[l];
    }
    if (int e = launch_pack(pa, st, f16 ? 3 : -1)) return e;
    hipLaunchKernelGGL(bias_sum_list_kernel, dim3(cdiv(d->Cs, 256)), dim3(256), 0, st, bl, nblocks, d->Cs, bsum);
    VQ_LAUNCH_CHECK();
  }
  if (!(stage & 2)) return 0;
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.nseg = nblocks;
  for (int l = 0; l < nblocks; ++l) {
    Seg& sg = g.seg[l];
    sg.x = z[l]; sg.x_bstride = (long)Ch * T; sg.x_cstride = T; sg.cin = Ch; sg.Tin = T;
    sg.tmul = 1; sg.toff = 0; sg.tdiv = 1; sg.w = w + (size_t)l * rp * ld; sg.ldw = ld;
    if (f16) { sg.amax_static = 1.f; sg.wamax = wam + l * AMAX_SLOTS; }
  }
  g.f16x2 = f16 ? 1 : 0;
  g.M = d->Cs; g.Tout = T; g.B = d->B;
  g.out[0].y = skip; g.out[0].y_bstride = (long)d->Cs * T; g.out[0].rows = d->Cs;
  g.out[0].bias = bsum;
  g.out[0].accumulate = accumulate;
  g.out[0].relu = relu ? 1 : 0;                  // the F.relu behind ResidualNet (modules.py:158) in this epilogue: one pass over (B, Cs, T) less
  g.out[0].amax_out = skip_amax_out;
  g.z16 = z_bf16(d) ? 1 : 0;
  g.x_nt = X3_SKIP_X_NT;
  return launch_gemm<EPI_LINEAR>(g, VQVAE_PROF_RESSTACK_SKIP, st);
}

extern "C" int vqvae_resstack_gcond_bwd(const vqvae_resblock_desc* d, int nblocks,
                                        const float* const* Wc, const float* const* gh,
                                        float* gcond, int accumulate, void* ws, size_t ws_bytes,
                                        vqvae_stream_t s) {
  if (int e = check_rb(d)) return e;
  VQ_REQUIRE(nblocks >= 1 && nblocks <= MAXSEG, "resstack_gcond_bwd: 1..%d blocks", MAXSEG);
  VQ_REQUIRE(Wc && gh && gcond && ws, "resstack_gcond_bwd: null pointer");
  VQ_REQUIRE(!(d->storage & VQVAE_STORE_GH_BF16), "resstack_gcond_bwd: reads fp32 gh (the latent-rate chain pulls a bf16 gh back with vqvae_upsample_linear_bwd_bf16)");
  if (ws_bytes < vqvae_resstack_workspace_bytes(d, nblocks)) { set_error("resstack_gcond_bwd: workspace too small"); return VQVAE_E_WORKSPACE; }
  hipStream_t st = (hipStream_t)s;
  const int T = d->T;
  const int ld = pad128(d->Cc), rp = slab_rows(d->Cd);
  float* w = (float*)ws;
  PackArgs pa; pa.njob = 0;
  for (int l = 0; l < nblocks; ++l)
    pa.job[pa.njob++] = pack_bwd_job(w + (size_t)l * rp * ld, Wc[l], d->Cd, d->Cc, 1, ld);
  if (int e = launch_pack(pa, st)) return e;
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.nseg = nblocks;
  for (int l = 0; l < nblocks; ++l) {
    Seg& sg = g.seg[l];
    sg.x = gh[l]; sg.x_bstride = (long)d->Cd * T; sg.x_cstride = T; sg.cin = d->Cd; sg.Tin = T;
    sg.tmul = 1; sg.toff = 0; sg.tdiv = 1; sg.w = w + (size_t)l * rp * ld; sg.ldw = ld;
  }
  g.M = d->Cc; g.Tout = T; g.B = d->B;
  g.out[0].y = gcond; g.out[0].y_bstride = (long)d->Cc * T; g.out[0].rows = d->Cc;
  g.out[0].accumulate = accumulate;
  return launch_gemm<EPI_LINEAR>(g, VQVAE_PROF_RESBLOCK_BWD_GC, st);
}

extern "C" int vqvae_resstack_skip_wgrad(const vqvae_resblock_desc* d, int nblocks,
                                         const float* g_skip, const float* const* z,
                                         float* const* gWs, float* const* gbs, int accumulate,
                                         void* ws, size_t ws_bytes, const uint32_t* g_skip_amax, vqvae_stream_t s) {
  if (int e = check_rb(d)) return e;
  VQ_REQUIRE(nblocks >= 1 && nblocks <= MAXSEG, "resstack_skip_wgrad: 1..%d blocks", MAXSEG);
  VQ_REQUIRE(g_skip && z && gWs && ws, "resstack_skip_wgrad: null pointer");
  if (ws_bytes < vqvae_resstack_workspace_bytes(d, nblocks)) { set_error("resstack_skip_wgrad: workspace too small"); return VQVAE_E_WORKSPACE; }
  hipStream_t st = (hipStream_t)s;
  const int Ch = d->Cd / 2, T = d->T;
  int cz[MAXSEG];
  for (int i = 0; i < nblocks; ++i) cz[i] = Ch;
  WgradPlan p = plan_wgrad(d->Cs, d->B, T, cz, nblocks);
  WgradArgs wa; memset(&wa, 0, sizeof(wa));
  wa.gy = g_skip; wa.gy_bstride = (long)d->Cs * T; wa.M = d->Cs; wa.Tout = T; wa.B = d->B;
  wa.nseg = nblocks;
  for (int l = 0; l < nblocks; ++l) {
    WSeg& sg = wa.seg[l];
    sg.x = z[l]; sg.x_bstride = (long)Ch * T; sg.x_cstride = T; sg.cin = Ch; sg.Tin = T;
    sg.tmul = 1; sg.toff = 0; sg.tdiv = 1;
    sg.gw = gWs[l]; sg.gw_co_stride = Ch; sg.gw_ci_stride = 1;
    wa.gbl[l] = gbs ? gbs[l] : nullptr;
    sg.amax_x_static = 1.f;                                   // |z| <= 1
  }
  wa.ngbl = gbs ? nblocks : 0;
  wa.accumulate = accumulate;
  wa.x16 = z_bf16(d) ? 1 : 0;
  if (g_matmul_dtype == 3 && g_skip_amax) { wa.f16x2 = 1; wa.amax_gy = g_skip_amax; }
  return launch_wgrad(wa, p, (float*)ws, VQVAE_PROF_WGRAD_RES_SKIP, st);
}

// gWr_l (+)= g_res_l z_l^T, gbr_l (+)= rowsum(g_res_l) for every block whose g_res_l
// is non-NULL -- one launch, each segment carrying its own output-gradient tensor.
extern "C" int vqvae_resstack_res_wgrad(const vqvae_resblock_desc* d, int nblocks,
                                        const float* const* g_res, const float* const* z,
                                        float* const* gWr, float* const* gbr, int accumulate,
                                        void* ws, size_t ws_bytes, const uint32_t* const* g_res_amax,
                                        vqvae_stream_t s) {
  if (int e = check_rb(d)) return e;
  VQ_REQUIRE(nblocks >= 1 && nblocks <= MAXSEG, "resstack_res_wgrad: 1..%d blocks", MAXSEG);
  VQ_REQUIRE(g_res && z && gWr && ws, "resstack_res_wgrad: null pointer");
  hipStream_t st = (hipStream_t)s;
  const int Ch = d->Cd / 2, T = d->T;
  int cz[MAXSEG];
  int n = 0;
  WgradArgs wa; memset(&wa, 0, sizeof(wa));
  for (int l = 0; l < nblocks; ++l) {
    if (!g_res[l] || !gWr[l]) continue;
    WSeg& sg = wa.seg[n];
    sg.gy = g_res[l];
    sg.x = z[l]; sg.x_bstride = (long)Ch * T; sg.x_cstride = T; sg.cin = Ch; sg.Tin = T;
    sg.tmul = 1; sg.toff = 0; sg.tdiv = 1;
    sg.gw = gWr[l]; sg.gw_co_stride = Ch; sg.gw_ci_stride = 1;
    sg.gb = gbr ? gbr[l] : nullptr;
    sg.amax_x_static = 1.f;                                   // |z| <= 1
    sg.amax_gy = g_res_amax ? g_res_amax[l] : nullptr;
    if (!sg.amax_gy) g_res_amax = nullptr;                    // one segment without its maximum: the whole launch keeps mode 2's kernel
    cz[n++] = Ch;
  }
  if (n == 0) return 0;
  if (g_matmul_dtype == 3 && g_res_amax) wa.f16x2 = 1;
  WgradPlan p = plan_wgrad(d->Cr, d->B, T, cz, n);
  if ((p.slab_floats + p.bslab_floats) * sizeof(float) > ws_bytes) { set_error("resstack_res_wgrad: workspace too small"); return VQVAE_E_WORKSPACE; }
  wa.gy = wa.seg[0].gy; wa.gy_bstride = (long)d->Cr * T; wa.M = d->Cr; wa.Tout = T; wa.B = d->B;
  wa.nseg = n;
  wa.accumulate = accumulate;
  wa.x16 = z_bf16(d) ? 1 : 0;
  wa.g16 = (d->storage & VQVAE_STORE_GRES_BF16) ? 1 : 0;      // every g_res of this launch
  return launch_wgrad(wa, p, (float*)ws, VQVAE_PROF_WGRAD_RES_SKIP, st);
}

extern "C" int vqvae_resblock_wgrad(const vqvae_resblock_desc* d, const float* x, const float* gh,
                                    float* gWd, float* gbd, int accumulate, void* ws,
                                    size_t ws_bytes, vqvae_stream_t s) {
  if (int e = check_rb(d)) return e;
  VQ_REQUIRE(x && gh && ws && (gWd || gbd), "resblock_wgrad: null pointer");
  hipStream_t st = (hipStream_t)s;
  const int T = d->T;
  int cins[MAXTAPS];
  for (int j = 0; j < d->K; ++j) cins[j] = d->Cr;
  WgradPlan p = plan_wgrad(d->Cd, d->B, T, cins, d->K);
  if ((p.slab_floats + p.bslab_floats) * sizeof(float) > ws_bytes) { set_error("resblock_wgrad: workspace too small"); return VQVAE_E_WORKSPACE; }
  WgradArgs wa; memset(&wa, 0, sizeof(wa));
  wa.gy = gh; wa.gy_bstride = (long)d->Cd * T; wa.M = d->Cd; wa.Tout = T; wa.B = d->B;
  wa.nseg = d->K;
  for (int j = 0; j < d->K; ++j) {
    WSeg& sg = wa.seg[j];
    sg.x = x; sg.x_bstride = (long)d->Cr * T; sg.x_cstride = T; sg.cin = d->Cr; sg.Tin = T;
    sg.tmul = 1; sg.toff = -(d->K - 1 - j) * d->dil; sg.tdiv = 1;
    sg.gw = gWd ? gWd + j : nullptr; sg.gw_co_stride = (long)d->Cr * d->K; sg.gw_ci_stride = d->K;
  }
  wa.seg[0].gb = gbd;
  wa.accumulate = accumulate;
  wa.g16 = (d->storage & VQVAE_STORE_GH_BF16) ? 1 : 0;
  return launch_wgrad(wa, p, (float*)ws, VQVAE_PROF_RESBLOCK_WGRAD, st);
}

// gWd_l (+)= sum_{b,t} gh_l[b,:,t] x_l[b,:,t - (K-1-j) dil_l]^T, gbd_l (+)= rowsum(gh_l) for several
// blocks in ONE launch: every (block, tap) pair is a segment with its own output-gradient tensor,
// input tensor and time shift, so the K splits (and the partial slabs the reduce re-reads) are
// shared by nblocks * K * Cr/128 tiles instead of paid per block.
extern "C" int vqvae_resstack_dil_wgrad(const vqvae_resblock_desc* d, int nblocks, const int* dils,
                                        const float* const* x, const float* const* gh,
                                        float* const* gWd, float* const* gbd, int accumulate,
                                        void* ws, size_t ws_bytes, const uint32_t* const* x_amax,
                                        const uint32_t* const* gh_amax, vqvae_stream_t s) {
  if (int e = check_rb(d)) return e;
  VQ_REQUIRE(nblocks >= 1 && nblocks * d->K <= MAXSEG, "resstack_dil_wgrad: nblocks * filter_size must be 1..%d", MAXSEG);
  VQ_REQUIRE(dils && x && gh && gWd && ws, "resstack_dil_wgrad: null pointer");
  hipStream_t st = (hipStream_t)s;
  const int T = d->T;
  int cins[MAXSEG];
  WgradArgs wa; memset(&wa, 0, sizeof(wa));
  int n = 0;
  for (int l = 0; l < nblocks; ++l) {
    VQ_REQUIRE(x[l] && gh[l] && dils[l] >= 1, "resstack_dil_wgrad: bad block %d", l);
    for (int j = 0; j < d->K; ++j) {
      WSeg& sg = wa.seg[n];
      sg.gy = gh[l];
      sg.x = x[l]; sg.x_bstride = (long)d->Cr * T; sg.x_cstride = T; sg.cin = d->Cr; sg.Tin = T;
      sg.tmul = 1; sg.toff = -(d->K - 1 - j) * dils[l]; sg.tdiv = 1;
      sg.gw = gWd[l] ? gWd[l] + j : nullptr; sg.gw_co_stride = (long)d->Cr * d->K; sg.gw_ci_stride = d->K;
      sg.gb = (j == 0 && gbd) ? gbd[l] : nullptr;
      sg.amax_x = x_amax ? x_amax[l] : nullptr;
      sg.amax_gy = gh_amax ? gh_amax[l] : nullptr;
      if (!sg.amax_x || !sg.amax_gy) x_amax = gh_amax = nullptr;      // (see resstack_res_wgrad)
      cins[n++] = d->Cr;
    }
  }
  if (g_matmul_dtype == 3 && x_amax && gh_amax) wa.f16x2 = 1;
  WgradPlan p = plan_wgrad(d->Cd, d->B, T, cins, n);
  if ((p.slab_floats + p.bslab_floats) * sizeof(float) > ws_bytes) { set_error("resstack_dil_wgrad: workspace too small"); return VQVAE_E_WORKSPACE; }
  wa.gy = wa.seg[0].gy; wa.gy_bstride = (long)d->Cd * T; wa.M = d->Cd; wa.Tout = T; wa.B = d->B;
  wa.nseg = n;
  wa.accumulate = accumulate;
  wa.g16 = (d->storage & (VQVAE_STORE_GH_BF16 | VQVAE_STORE_GH_F16X2)) ? 1 : 0;
  wa.x16 = (d->storage & (VQVAE_STORE_X_BF16 | VQVAE_STORE_X_F16X2)) ? 1 : 0;        // every block of this launch (the caller groups them accordingly)
  if (d->storage & (VQVAE_STORE_GH_F16X2 | VQVAE_STORE_X_F16X2))
    VQ_REQUIRE(wa.f16x2, "resstack_dil_wgrad: pre-split operands (desc.storage) need the float32x2 launch: every block's scale words");
  return launch_wgrad(wa, p, (float*)ws, VQVAE_PROF_WGRAD_DIL, st);
}

extern "C" size_t vqvae_resstack_dil_wgrad_workspace_bytes(const vqvae_resblock_desc* d, int nblocks) {
  if (!d || nblocks < 1 || nblocks * d->K > MAXSEG) return 0;
  int cins[MAXSEG];
  for (int i = 0; i < nblocks * d->K; ++i) cins[i] = d->Cr;
  size_t need = 0;             // any group of 1..nblocks blocks may be flushed; fewer tiles can mean more splits
  for (int n = 1; n <= nblocks; ++n) {
    WgradPlan p = plan_wgrad(d->Cd, d->B, d->T, cins, n * d->K);
    if (p.slab_floats + p.bslab_floats > need) need = p.slab_floats + p.bslab_floats;
  }
  return need * sizeof(float) + 256;
}

// ---------------------------------------------------------------------------
// The second half of the fused latent pull-back (gemm_epilogue, EPI_GATE_BWD, OUT bit 1): every 128-column tile left its
// sums for the four latent positions under it; an output position collects the (at most three) tiles that cover it, in
// ascending tile order.
// ---------------------------------------------------------------------------
namespace vq {
// one workgroup per (block l, batch item b, 64 gate channels): a thread walks ITS channel's tiles in ascending order (float4
// per tile: coalesced across the channels) and adds them into its row of an LDS image of the output, which then leaves in
// whole rows -- deterministic (one owner per channel, fixed order), every byte of `part` read once
constexpr int PBR_C = 64;
__global__ __launch_bounds__(PBR_C) void pullback_reduce_kernel(const float* __restrict__ part, const int32_t* __restrict__ v0,
                                                               int nblocks, int B, int nt, int Cd, int Tl, float* __restrict__ gP, long gP_bstride) {
  extern __shared__ float img[];                 // [PBR_C][Tl + 1]
  const int chunks = Cd / PBR_C;
  const int cchunk = blockIdx.x % chunks;
  const int b = (blockIdx.x / chunks) % B;
  const int l = blockIdx.x / (chunks * B);
  const int c = cchunk * PBR_C + threadIdx.x;
  float* row = img + threadIdx.x * (Tl + 1);
  for (int v = 0; v < Tl; ++v) row[v] = 0.f;
  const float4* src = reinterpret_cast<const float4*>(part) + (((long)l * B + b) * nt) * Cd + c;
#pragma unroll 6                                 // (the tiles' loads are independent: six travel together)
  for (int n = 0; n < nt; ++n) {
    const int vb = v0[n * BN];                   // wave-uniform
    const float4 p = src[(long)n * Cd];
    row[vb] += p.x;
    if (vb + 1 < Tl) row[vb + 1] += p.y;
    if (vb + 2 < Tl) row[vb + 2] += p.z;
    if (vb + 3 < Tl) row[vb + 3] += p.w;
  }
  __syncthreads();
  float* dst = gP + (long)b * gP_bstride + ((long)l * Cd + cchunk * PBR_C) * Tl;      // PBR_C consecutive rows of Tl: one contiguous run
  for (int i = threadIdx.x; i < PBR_C * Tl; i += PBR_C) dst[i] = img[(i / Tl) * (Tl + 1) + i % Tl];
}
}  // namespace vq

extern "C" int vqvae_pullback_reduce_into(const float* part, const int32_t* v0, int nblocks, int B, int T, int Cd, int Tl,
                                          float* gP, size_t gP_bstride, vqvae_stream_t s) {
  VQ_REQUIRE(part && v0 && gP && nblocks > 0 && B > 0 && Cd > 0 && Tl > 0 && T % vq::BN == 0 && (long)T >= 64L * Tl,
             "pullback_reduce: bad arguments (T %% 128 == 0, T >= 64 Tl)");
  VQ_REQUIRE(Cd % vq::PBR_C == 0 && (size_t)vq::PBR_C * (Tl + 1) * 4 <= 64 * 1024, "pullback_reduce: Cd %% 64 == 0, Tl <= 255");
  VQ_REQUIRE(gP_bstride >= (size_t)nblocks * Cd * Tl, "pullback_reduce: batch stride of gP smaller than the rows written");
  const size_t lds = (size_t)vq::PBR_C * (Tl + 1) * sizeof(float);
  hipLaunchKernelGGL(vq::pullback_reduce_kernel, dim3((unsigned)(nblocks * B * (Cd / vq::PBR_C))), dim3(vq::PBR_C), lds, (hipStream_t)s, part, v0, nblocks, B, T / vq::BN, Cd, Tl, gP, (long)gP_bstride);
  VQ_LAUNCH_CHECK();
  return 0;
}
extern "C" int vqvae_pullback_reduce(const float* part, const int32_t* v0, int nblocks, int B, int T, int Cd, int Tl,
                                     float* gP, vqvae_stream_t s) {
  return vqvae_pullback_reduce_into(part, v0, nblocks, B, T, Cd, Tl, gP, (size_t)nblocks * Cd * Tl, s);
}
