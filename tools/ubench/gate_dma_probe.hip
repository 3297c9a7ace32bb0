// Probe (dev tool, round 5): the two-tap 256 x 128 float32x2 K loop with BOTH operands brought into LDS by
// LDS-DMA (buffer_load_dwordx4 ... lds) -- no VGPR staging, no split, no ds_write -- from
//   * the packed weight slab pack_kernel already writes ([k-step][piece][k-half][row] 16-byte words), and
//   * an activation tensor stored PRE-SPLIT in fragment order: [b][channel group of 8][piece][t] 16-byte words
//     (8 fp16 of 8 consecutive channels at one t): a dilated tap is a row offset of 16 bytes x dilation.
// Questions:
//   1. does an out-of-range lane of a buffer_load ... lds write 0 to its LDS slot?
//   2. what does ds_read_b64_tr_b16 deliver (lane / element map), i.e. can a K = time fragment be read out of
//      the channel-grouped image?
//   3. how long does the K loop take (gate launch equivalent: B = 16, T = 7680, 256 rows, K = 2 x 256)?
// build: hipcc --offload-arch=gfx950 -O3 -o gate_dma_probe gate_dma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));
// one LDS-DMA of 16 bytes per lane: LDS[m0 + 16 lane] = buffer[voff + soff]; issued from inline asm so that hipcc's wait-count
// pass does not know about it (it would drain vmcnt(0) in front of every ds_read otherwise): the caller counts vmcnt itself
__device__ __forceinline__ void dma16(unsigned lds_dst, unsigned voff, i32x4 rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_dst), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc4(const void* p, unsigned bytes) {
  const unsigned long long a = (unsigned long long)p;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(unsigned long long)(lds_ptr)p; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- 1. out-of-range lanes of an LDS-DMA ------------------------------------------------------------
__global__ void oob_probe(const unsigned* src, unsigned* out) {
  __shared__ uint4 lds[64];
  lds[threadIdx.x] = make_uint4(0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu);
  __syncthreads();
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, 32 * 16, 0x00020000);   // 32 words in range
  const unsigned vo = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16u;     // odd lanes, and lanes >= 32, are out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)&lds[0], 16, vo, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const uint4 v = lds[threadIdx.x];
  out[4 * threadIdx.x + 0] = v.x; out[4 * threadIdx.x + 1] = v.y; out[4 * threadIdx.x + 2] = v.z; out[4 * threadIdx.x + 3] = v.w;
}

// ---- 2. ds_read_b64_tr_b16 ----------------------------------------------------------------------------
typedef short s4_t __attribute__((ext_vector_type(4)));
__global__ void tr_probe(unsigned short* out, int mode) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned off;        // in 16-bit elements
  if (mode == 0) off = 4u * l;                                      // lane-linear 8-byte addresses
  else {
    // channel-grouped image [cg][t][8 ch] (one piece): element (cg, t, c) at ((cg * 16 + t) * 8 + c), 16 t per block.
    // wanted: lane (m = l & 31, kg = l >> 5) gets t = 8 kg + {0..3} (this read) of channel m.
    const int g = l >> 4, i = l & 15;                               // 16-lane group, lane in group
    const int chbase = 16 * (g & 1), tbase = 8 * (g >> 1);          // group: channels chbase..+15, times tbase..+3
    const int row = i >> 2, chunk = i & 3;                          // row = time within the 4, chunk = 4 channels
    const int ch = chbase + 4 * chunk, t = tbase + row;
    off = (unsigned)(((ch >> 3) * 16 + t) * 8 + (ch & 7));
  }
  const s4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4_t*)(lds + off));
  for (int j = 0; j < 4; ++j) out[4 * l + j] = (unsigned short)v[j];
}

// ---- 3. the K loop --------------------------------------------------------------------------------------
constexpr int NSTEP = 32;          // 2 taps x 16 channel groups of 16
constexpr int NBUF = 3;
struct LoopArgs {
  const uint4* w;                  // [NSTEP][2 pieces][2 k-halves][256] words
  const uint4* x;                  // [B][32 channel groups][2 pieces][T] words
  float* y;                        // (B, 384, T) fp32
  int T, dil, epi, ntile_n;
};

template <int MODE>                // 0: all-DMA; 1: A by DMA, B by DMA, two buffers + vmcnt(0) (simplest form)
__global__ __launch_bounds__(512, 4) void gate_loop(const LoopArgs a) {
  __shared__ uint4 As[NBUF][2][2][256];
  __shared__ uint4 Bs[NBUF][2][2][128];
  const int nblk = gridDim.x;
  int logical;
  {
    const int id = blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, xcd = id & 7;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
  }
  const int nt = logical % a.ntile_n, b = logical / a.ntile_n;
  const int t0 = nt * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const i32x4 rw = make_rsrc4(a.w, NSTEP * 1024 * 16);
  const i32x4 rx = make_rsrc4(a.x + (long)b * 64 * a.T, 64u * (unsigned)a.T * 16u);
  // B: this wave's (piece, k-half, column half)
  const int bp = wave >> 2, bkh = (wave >> 1) & 1, bth = wave & 1;
  const int tc = t0 + 64 * bth + lane;
  const unsigned vb1 = 16u * (unsigned)tc;                                       // tap 1: x[t]
  const unsigned vb0 = tc - a.dil >= 0 ? 16u * (unsigned)(tc - a.dil) : 0x80000000u;   // tap 0: x[t - dil]
  const unsigned rowb = 16u * (unsigned)a.T;                                     // bytes per (channel group, piece) row
  const unsigned va = 16u * (unsigned)lane;

  auto issue = [&](int s, int buf) {
    // weights: chunks 2 wave, 2 wave + 1 of the step's 16 (64 words each)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = 2 * wave + j;
      dma16(lds_addr(&As[buf][0][0][0] + 64 * c), va, rw, (unsigned)(s * 1024 + c * 64) * 16u);
    }
    const int cg = 2 * (s >> 1) + bkh;
    dma16(lds_addr(&Bs[buf][bp][bkh][64 * bth]), (s & 1) ? vb1 : vb0, rx, (unsigned)(cg * 2 + bp) * rowb);
  };
  auto compute = [&](int buf) {
    uint4 bf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) bf[j][p] = Bs[buf][p][lk][wn * 64 + j * 32 + li];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint4 af[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) af[p] = As[buf][p][lk][wm * 64 + i * 32 + li];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[1]), __builtin_bit_cast(f16x8, bf[j][0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[j][1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0]), __builtin_bit_cast(f16x8, bf[j][0]), c, 0, 0, 0);
        acc[i][j] = c;
      }
    }
  };
  if constexpr (MODE == 0) {
    issue(0, 0);
    issue(1, 1);
    int buf = 0;
    for (int s = 0; s < NSTEP; ++s) {
      if (s + 1 < NSTEP) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + 2 < NSTEP) { const int nb = buf == 0 ? 2 : buf - 1; issue(s + 2, nb); }
      compute(buf);
      buf = buf == 2 ? 0 : buf + 1;
    }
  } else {
    issue(0, 0);
    for (int s = 0; s < NSTEP; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + 1 < NSTEP) issue(s + 1, (s + 1) & 1);
      compute(s & 1);
    }
  }
  // ---- epilogue stand-ins
  if (a.epi == 0) {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) v += acc[i][j][r];
    if (v == 12345.678f) a.y[tid] = v;
    return;
  }
  // epi 1: the accumulators as fp32 (B, 256, T) (a linear epilogue's stores); epi 2: plus 128 more rows (the gate kernel's 1.5 x)
  const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y + (long)b * 384 * a.T, 0, 384 * a.T * 4, 0x00020000);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk, col = t0 + wn * 64 + j * 32 + li;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc[i][j][r]), ry, 4u * (unsigned)col, 4u * (unsigned)(row * a.T), 2);
        if (a.epi == 2 && i == 0)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc[i][j][r] * 0.5f), ry, 4u * (unsigned)col, 4u * (unsigned)((256 + (row >> 1)) * a.T), 2);
      }
}

int main(int argc, char** argv) {
  // 1
  {
    unsigned *src, *out;
    CK(hipMalloc(&src, 64 * 16)); CK(hipMalloc(&out, 64 * 16));
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000u + i;
    CK(hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice));
    oob_probe<<<1, 64>>>(src, out);
    CK(hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost));
    printf("[1] LDS-DMA, out-of-range lanes (odd lanes and lanes >= 32): word.x per lane\n   ");
    for (int l = 0; l < 64; ++l) printf("%x%s", h[4 * l], (l & 15) == 15 ? "\n   " : " ");
    printf("\n");
  }
  // 2
  for (int mode = 0; mode < 2; ++mode) {
    unsigned short* out;
    CK(hipMalloc(&out, 64 * 4 * 2));
    tr_probe<<<1, 64>>>(out, mode);
    std::vector<unsigned short> h(256);
    CK(hipMemcpy(h.data(), out, 512, hipMemcpyDeviceToHost));
    printf("[2] ds_read_b64_tr_b16 mode %d: lane: 4 element indices read%s\n", mode, mode ? " (decoded as cg,t,ch)" : "");
    for (int l = 0; l < 64; ++l) {
      printf("   l%2d:", l);
      for (int j = 0; j < 4; ++j) {
        if (mode == 0) printf(" %4d", h[4 * l + j]);
        else { const int e = h[4 * l + j]; printf(" (cg%d t%2d c%d)", e / 128, (e / 8) % 16, e % 8); }
      }
      printf((l & 1) ? "\n" : "   ");
    }
  }
  // 3
  const int B = 16, T = 7680;
  uint4 *w, *x; float* y;
  const size_t wbytes = (size_t)NSTEP * 1024 * 16, xbytes = (size_t)B * 64 * T * 16, ybytes = (size_t)B * 384 * T * 4;
  CK(hipMalloc(&w, wbytes)); CK(hipMalloc(&x, xbytes)); CK(hipMalloc(&y, ybytes));
  {
    std::vector<unsigned short> hw(wbytes / 2), hx(xbytes / 2);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    // random fp16 in roughly [-2, 2): sign, exponent 10..15, random mantissa
    auto f16 = [&]() { const unsigned r = rnd(); return (unsigned short)(((r >> 31) << 15) | ((10u + ((r >> 20) % 6u)) << 10) | ((r >> 8) & 0x3ffu)); };
    for (auto& v : hw) v = f16();
    for (auto& v : hx) v = f16();
    CK(hipMemcpy(w, hw.data(), wbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, hx.data(), xbytes, hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 2; ++mode)
    for (int epi = 0; epi < 3; ++epi)
      for (int dil : {1, 512}) {
        LoopArgs a{w, x, y, T, dil, epi, T / 128};
        const int grid = B * (T / 128);
        auto launch = [&]() {
          if (mode == 0) gate_loop<0><<<grid, 512>>>(a); else gate_loop<1><<<grid, 512>>>(a);
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("[3] mode %d (%s) epi %d dil %3d: %.1f us per launch  (%.0f TFLOP/s of fp32 work)\n", mode, mode ? "2 buffers, vmcnt(0)" : "3 buffers, counted waits", epi, dil,
               ms * 1e3 / 20, 32.21e9 / (ms * 1e-3 / 20) / 1e12);
      }
  return 0;
}
