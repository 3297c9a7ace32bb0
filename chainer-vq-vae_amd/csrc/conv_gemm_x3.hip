// conv_gemm_x3.hip -- matmul modes 1-3: forward and backward-data of every conv on the 16-bit matrix pipe
// (conv_gemm_x3_kernel: NP = 3 six bf16 products of an exact split, NP = 2 three fp16 products of a scaled split, NP = 1
// operands rounded to bf16), the streaming residual 1x1 (lin128_stream_kernel), the split-K reduce, and launch_gemm, which
// picks the kernel of a launch in every mode.
#include "gemm_common.h"

namespace vq {

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
int plan_ksplit(int M, int Tout, int B, int nk) {
  if (M % 256 == 0) return 1;
  const long tiles = (long)cdiv(M, 128) * cdiv(Tout, BN) * B;
  if (tiles > 128 || nk < 32) return 1;
  long s = 512 / tiles;                     // fill ~half the chip's 1024 slots
  if (s > nk / 8) s = nk / 8;               // at least 8 K steps per split
  if (s > 32) s = 32;
  return s < 2 ? 1 : (int)s;
}
size_t ksplit_partial_floats(int M, int Tout, int B, int nk) {
  const int s = plan_ksplit(M, Tout, B, nk);
  return s > 1 ? (size_t)s * cdiv(M, 128) * cdiv(Tout, BN) * B * 128 * 128 : 0;
}

template <int EPI>
int launch_gemm(GemmArgs& g, int tag, hipStream_t st) {
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
      if (int e = launch_gemm_fp32<EPI>(g, 4, (unsigned)nblk, attach ? tag : 0, st)) return e;
    }
    if (!big) {
      if (mode != 0) X3_LAUNCH_MODE(2, 1, grid, 256);
      else if (int e = launch_gemm_fp32<EPI>(g, 2, (unsigned)grid, attach ? tag : 0, st)) return e;
    }
  } else {
    if (mode != 0) X3_LAUNCH_MODE(2, 1, grid, 256);
    else if (int e = launch_gemm_fp32<EPI>(g, 2, (unsigned)grid, attach ? tag : 0, st)) return e;
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

template int launch_gemm<EPI_LINEAR>(GemmArgs&, int, hipStream_t);
template int launch_gemm<EPI_GATE>(GemmArgs&, int, hipStream_t);
template int launch_gemm<EPI_GATE_BWD>(GemmArgs&, int, hipStream_t);

}  // namespace vq
