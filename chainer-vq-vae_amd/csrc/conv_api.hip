// conv_api.hip -- the C ABI of the conv / ResidualBlock / ResidualNet entry points (include/vqvae_hip.h) and what only they
// launch: weight packing (pack_kernel), maxima and norms behind the float32x2 scales and bounds (wamax / absmax / wl1
// kernels), the stride-2 phase split, the latent pull-back reduce.  The contraction kernels live in conv_gemm_x3.hip,
// conv_gemm_fp32.hip and wgrad.hip (gemm_common.h).
#include "gemm_common.h"
#include <stdlib.h>
#include <map>
#include <mutex>

namespace vq {

int g_matmul_dtype = 2;
int g_wgrad_impl = 0;

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
// wl1_kernel -- the weight norms behind the a-priori bounds of the pre-split tensors (see presplit_pair), three
// workgroups per ResidualBlock, once per step beside pack_kernel:
//   out[0] = max_r (sum_c |Wr[r][c]| + |br[r]|)   bounds |Wr z + br| for |z| <= 1                (x_{l+1}'s bound)
//   out[1] = max_c sum_r |Wr[r][c]|,  out[2] = max_c sum_s |Ws[s][c]|   bound |Wr^T g|, |Ws^T g| per unit max |g|  (gh_l's bound)
// Wr (Cr, Ch), Ws (Cs, Ch) row-major (Chainer (Cout, Cin, 1, 1)).  fp32 sums of <= 256 magnitudes: relative error
// 2^-16, inside bound_margin's 2^-10.
// ---------------------------------------------------------------------------
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
  const bool both_presplit = (wa.f16x2 && (d->storage & VQVAE_STORE_GH_F16X2) && (d->storage & VQVAE_STORE_X_F16X2)) ||
                             (g_matmul_dtype == 1 && (d->storage & VQVAE_STORE_GH_BF16) && (d->storage & VQVAE_STORE_X_BF16) && T % 64 == 0);   // (mode 1: both stored as bf16)
  WgradPlan p = plan_wgrad(d->Cd, d->B, T, cins, n, both_presplit && wgrad_dma_shape(d->Cd, T, cins, n));
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
    for (int wide = 0; wide < 2; ++wide) {     // the launch takes the wide plan when both operands are stored pre-split (wgrad3_dma_kernel)
      if (wide && !wgrad_dma_shape(d->Cd, d->T, cins, n * d->K)) continue;
      WgradPlan p = plan_wgrad(d->Cd, d->B, d->T, cins, n * d->K, wide != 0);
      if (p.slab_floats + p.bslab_floats > need) need = p.slab_floats + p.bslab_floats;
    }
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
