// Incremental WaveNet generation: WaveNet.generate / ResidualNet.generate /
// ResidualBlock.push+pop (WaveNet/modules.py:58-74, 98-110, 232-255) and the sampling loop of
// generate.py:105-145, one audio sample per step.
//
// A step is a chain of ~2 * n_blocks + 4 dependent matrix-vector products on vectors of a few hundred
// floats: it is latency-bound, not MFMA- or HBM-bound.  The design therefore minimises the dependent
// edges and keeps the host out of the loop:
//   * the step index lives in device memory; every kernel reads it, the last one advances it, so
//     the SAME launch sequence is valid for every step and can be captured once into a hipGraph
//     (vqvae_graph_*) and replayed;
//   * each block's queue (modules.py:58-62) is a ring of `dilation` slots: slot t % dilation holds
//     x[t - dilation] until the block's second kernel overwrites it with x[t];
//   * the sampler runs on the device from caller-supplied uniform doubles (the numbers NumPy's
//     global RNG would hand generate.py:117/136) and writes the next input vector in place.
// Weights are read in their Chainer layout (no packing); 15.8 MB of fp32 weights stay L2/MALL resident.
#include "common.h"

namespace vq {

constexpr int GEN_NB = 4;        // sequences generated in lockstep (generate.py:42 runs 1)
constexpr int GEN_MAXOUT = 1024; // categorical sampler: classes held in LDS

__device__ __forceinline__ float gen_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- dilated 2-tap "conv" on the queue ends (+ condition projection, + gate) ------------------
struct GenConvArgs {
  const float* W;      // (rows, Cin, 2): tap 0 meets the OLD sample, tap 1 the new one
  const float* b;
  const float* Wc;     // (rows, Cc) or nullptr
  const float* bc;
  const float* x_old;  // ring (ring_len, n, Cin), or (n, Cin) when ring_len == 0
  const float* x_new;  // (n, Cin)
  const float* cond;   // element (b, k) of this step at cond[b*bstride + k*cstride + (cond_step ? t : 0)]
  long cond_bstride, cond_cstride;
  float* out;          // (n, rows) linear, (n, rows/2) gated
  const int* step;
  int max_steps, cond_step, n, rows, Cin, Cc, ring_len, gate;
};

__global__ __launch_bounds__(256) void gen_conv_kernel(GenConvArgs a) {
  const int t = *a.step;
  if (t >= a.max_steps) return;
  const int lane = threadIdx.x & 63;
  const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int half = a.rows >> 1;
  const int ntask = a.gate ? half : a.rows;
  if (task >= ntask) return;
  const float* xo = a.x_old + (a.ring_len > 0 ? (size_t)(t % a.ring_len) * a.n * a.Cin : 0);
  const float* cnd = a.cond + (a.cond_step ? t : 0);
  float acc[2][GEN_NB];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int b = 0; b < GEN_NB; ++b) acc[r][b] = 0.f;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (r == 1 && !a.gate) break;
    const int row = task + r * half;
    const float2* w = reinterpret_cast<const float2*>(a.W + (size_t)row * a.Cin * 2);
#pragma unroll 4
    for (int k = lane; k < a.Cin; k += 64) {
      const float2 wv = w[k];
#pragma unroll
      for (int b = 0; b < GEN_NB; ++b)
        if (b < a.n) acc[r][b] += wv.x * xo[b * a.Cin + k] + wv.y * a.x_new[b * a.Cin + k];
    }
    if (a.Wc) {
      const float* wc = a.Wc + (size_t)row * a.Cc;
#pragma unroll 3
      for (int k = lane; k < a.Cc; k += 64) {
        const float wv = wc[k];
#pragma unroll
        for (int b = 0; b < GEN_NB; ++b)
          if (b < a.n) acc[r][b] += wv * cnd[b * a.cond_bstride + k * a.cond_cstride];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int b = 0; b < GEN_NB; ++b) acc[r][b] = gen_wave_sum(acc[r][b]);
  if (lane != 0) return;
#pragma unroll
  for (int b = 0; b < GEN_NB; ++b) {
    if (b >= a.n) break;
    float h0 = acc[0][b] + a.b[task];
    if (a.Wc) h0 += a.bc[task];
    if (a.gate) {
      float h1 = acc[1][b] + a.b[task + half];
      if (a.Wc) h1 += a.bc[task + half];
      a.out[b * half + task] = tanhf(h0) * (1.f / (1.f + expf(-h1)));   // modules.py:47-48
    } else {
      a.out[b * a.rows + task] = h0;
    }
  }
}

// ---- 1x1 projections (res + skip of a block, proj1, proj2) --------------------------------------
enum { GEN_RELU_IN = 1, GEN_RELU_OUT = 2, GEN_ACCUM = 4 };
struct GenDenseJob {
  const float* W;    // (rows, K)
  const float* b;
  const float* in;   // (n, K)
  const float* add;  // (n, rows) or nullptr
  float* out;        // (n, rows)
  int rows, K, flags;
};
struct GenDenseArgs {
  GenDenseJob job[2];
  int njobs, n;
  const int* step;
  int max_steps;
  // side duties of the last workgroup: queue push (modules.py:71-74) and embed-queue shift (247)
  const float* push_src; float* ring; int ring_len, push_elems;
  const float* shift_src; float* shift_dst; int shift_elems;
};

__global__ __launch_bounds__(256) void gen_dense_kernel(GenDenseArgs a) {
  const int t = *a.step;
  if (t >= a.max_steps) return;
  if (blockIdx.x == gridDim.x - 1) {
    if (a.ring) {
      float* dst = a.ring + (size_t)(t % a.ring_len) * a.push_elems;
      for (int i = threadIdx.x; i < a.push_elems; i += 256) dst[i] = a.push_src[i];
    }
    if (a.shift_dst)
      for (int i = threadIdx.x; i < a.shift_elems; i += 256) a.shift_dst[i] = a.shift_src[i];
  }
  const int lane = threadIdx.x & 63;
  int task = blockIdx.x * 4 + (threadIdx.x >> 6);
  int j = 0;
  if (task >= a.job[0].rows) { task -= a.job[0].rows; j = 1; }
  if (j >= a.njobs || task >= a.job[j].rows) return;
  const GenDenseJob& jb = a.job[j];
  float acc[GEN_NB];
#pragma unroll
  for (int b = 0; b < GEN_NB; ++b) acc[b] = 0.f;
  const float* w = jb.W + (size_t)task * jb.K;
#pragma unroll 4
  for (int k = lane; k < jb.K; k += 64) {
    const float wv = w[k];
#pragma unroll
    for (int b = 0; b < GEN_NB; ++b)
      if (b < a.n) {
        float v = jb.in[b * jb.K + k];
        if (jb.flags & GEN_RELU_IN) v = fmaxf(v, 0.f);
        acc[b] += wv * v;
      }
  }
#pragma unroll
  for (int b = 0; b < GEN_NB; ++b) acc[b] = gen_wave_sum(acc[b]);
  if (lane != 0) return;
#pragma unroll
  for (int b = 0; b < GEN_NB; ++b) {
    if (b >= a.n) break;
    float v = acc[b] + jb.b[task];
    if (jb.flags & GEN_RELU_OUT) v = fmaxf(v, 0.f);
    if (jb.add) v += jb.add[b * jb.rows + task];
    if (jb.flags & GEN_ACCUM) v = jb.out[b * jb.rows + task] + v;     // modules.py:105-109 order
    jb.out[b * jb.rows + task] = v;
  }
}

// ---- sampling + feedback + step advance (generate.py:109-145) -----------------------------------
struct GenFinishArgs {
  const float* logits;      // (n, out_dim)
  const double* uniforms;   // (T, n, n_uniform)
  const void* forced;       // (T, n) int32 / float or nullptr
  void* out;                // int32 index / float value at [b * out_bstride + t]
  long out_bstride;
  float* logits_out;        // (T, n, out_dim) or nullptr
  float* x_cur;             // (n, input_dim): next input, written in place
  int* step;
  int max_steps, n, out_dim, input_dim, mode, n_uniform;
  float log_scale_min;
};

__global__ __launch_bounds__(256) void gen_finish_kernel(GenFinishArgs a) {
  __shared__ float p_s[GEN_MAXOUT];
  __shared__ double cdf_s[GEN_MAXOUT];
  __shared__ float red_s[4];
  __shared__ int cnt_s[4];
  const int t = *a.step;
  if (t >= a.max_steps) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.logits_out)
    for (int i = tid; i < a.n * a.out_dim; i += 256) a.logits_out[(size_t)t * a.n * a.out_dim + i] = a.logits[i];
  if (a.mode == VQVAE_GEN_SOFTMAX) {
    for (int b = 0; b < a.n; ++b) {
      const float* l = a.logits + b * a.out_dim;
      // chainer.functions.softmax: exp(l - max) / sum, fp32 (generate.py:138)
      float m = -INFINITY;
      for (int i = tid; i < a.out_dim; i += 256) m = fmaxf(m, l[i]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      if (lane == 0) red_s[wave] = m;
      __syncthreads();
      m = fmaxf(fmaxf(red_s[0], red_s[1]), fmaxf(red_s[2], red_s[3]));
      __syncthreads();
      float s = 0.f;
      for (int i = tid; i < a.out_dim; i += 256) { const float e = expf(l[i] - m); p_s[i] = e; s += e; }
      s = gen_wave_sum(s);
      if (lane == 0) red_s[wave] = s;
      __syncthreads();
      s = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
      for (int i = tid; i < a.out_dim; i += 256) p_s[i] = p_s[i] / s;
      __syncthreads();
      // numpy.random.choice: float64 cumsum, normalised by its last element,
      // searchsorted(u, side='right') == number of cdf entries <= u   (generate.py:136-138)
      if (tid == 0) {
        double c = 0.0;
        for (int i = 0; i < a.out_dim; ++i) { c += (double)p_s[i]; cdf_s[i] = c; }
      }
      __syncthreads();
      const double total = cdf_s[a.out_dim - 1];
      const double u = a.uniforms[((size_t)t * a.n + b) * a.n_uniform];
      int cnt = 0;
      for (int i = tid; i < a.out_dim; i += 256) cnt += (cdf_s[i] / total <= u) ? 1 : 0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
      if (lane == 0) cnt_s[wave] = cnt;
      __syncthreads();
      int idx = cnt_s[0] + cnt_s[1] + cnt_s[2] + cnt_s[3];
      idx = idx < a.out_dim ? idx : a.out_dim - 1;
      if (tid == 0 && a.out) reinterpret_cast<int32_t*>(a.out)[b * a.out_bstride + t] = idx;
      const int nxt = a.forced ? reinterpret_cast<const int32_t*>(a.forced)[(size_t)t * a.n + b] : idx;
      for (int i = tid; i < a.input_dim; i += 256) a.x_cur[b * a.input_dim + i] = (i == nxt) ? 1.f : 0.f;  // generate.py:139-141
      __syncthreads();
    }
  } else if (a.mode == VQVAE_GEN_MOL) {
    if (tid < a.n) {
      const int b = tid, nr = a.out_dim / 3;
      const float* l = a.logits + b * a.out_dim;
      float m = -INFINITY;
      for (int j = 0; j < nr; ++j) m = fmaxf(m, l[j]);
      float s = 0.f;
      for (int j = 0; j < nr; ++j) s += expf(l[j] - m);
      double acc = 0.0;
      for (int j = 0; j < nr; ++j) {
        const float pj = expf(l[j] - m) / s;
        const float sc = expf(fmaxf(l[2 * nr + j], a.log_scale_min));               // generate.py:111-112
        const double u = a.uniforms[((size_t)t * a.n + b) * a.n_uniform + j];
        double r = (double)l[nr + j] + (double)sc * (log(u) - log(1.0 - u));        // generate.py:115-118
        acc += r * (double)pj;                                                      // generate.py:121-122
      }
      float v = (float)acc;
      v = v / 127.5f;                                                               // generate.py:125
      v = fminf(fmaxf(v, -1.f), 1.f);
      if (a.out) reinterpret_cast<float*>(a.out)[b * a.out_bstride + t] = v;
      const float nxt = a.forced ? reinterpret_cast<const float*>(a.forced)[(size_t)t * a.n + b] : v;
      for (int i = 0; i < a.input_dim; ++i) a.x_cur[b * a.input_dim + i] = nxt;     // generate.py:127
    }
  }
  __syncthreads();
  if (tid == 0) *a.step = t + 1;
}

}  // namespace vq

using namespace vq;

extern "C" int vqvae_wavenet_gen_step(const vqvae_gen_desc* d, vqvae_stream_t s) {
  VQ_REQUIRE(d, "gen_step: null descriptor");
  VQ_REQUIRE(d->n >= 1 && d->n <= GEN_NB, "gen_step: n must be 1..%d sequences (got %d)", GEN_NB, d->n);
  VQ_REQUIRE(d->n_blocks >= 1 && d->blocks, "gen_step: no residual blocks");
  VQ_REQUIRE(d->input_dim >= 1 && d->residual >= 1 && d->dilated >= 2 && d->dilated % 2 == 0 && d->skip >= 1 &&
             d->out_dim >= 1 && d->cond_dim >= 0, "gen_step: bad dimensions");
  VQ_REQUIRE(d->embed_W && d->embed_b && d->proj1_W && d->proj1_b && d->proj2_W && d->proj2_b, "gen_step: null head weights");
  VQ_REQUIRE(d->step && d->x_cur && d->x_prev && d->h0 && d->h1 && d->z && d->skip_acc && d->s1 && d->logits, "gen_step: null state buffer");
  VQ_REQUIRE(d->cond_dim == 0 || d->cond, "gen_step: condition missing");
  VQ_REQUIRE(d->sample_mode == VQVAE_GEN_NONE || d->sample_mode == VQVAE_GEN_SOFTMAX || d->sample_mode == VQVAE_GEN_MOL, "gen_step: bad sample_mode");
  if (d->sample_mode != VQVAE_GEN_NONE) {
    VQ_REQUIRE(d->uniforms && d->n_uniform >= 1, "gen_step: sampling needs uniforms");
    if (d->sample_mode == VQVAE_GEN_SOFTMAX) VQ_REQUIRE(d->out_dim <= GEN_MAXOUT, "gen_step: at most %d classes", GEN_MAXOUT);
    else VQ_REQUIRE(d->out_dim % 3 == 0 && d->n_uniform >= d->out_dim / 3, "gen_step: mixture sampling needs out_dim = 3*nr_mix and nr_mix uniforms");
  }
  hipStream_t st = (hipStream_t)s;
  const int n = d->n, R = d->residual, half = d->dilated / 2;

  GenConvArgs e{};   // embed: pad-free 2-tap conv on [previous input, current input] (modules.py:234, 247-248)
  e.W = d->embed_W; e.b = d->embed_b; e.x_old = d->x_prev; e.x_new = d->x_cur; e.out = d->h0; e.step = d->step;
  e.max_steps = d->max_steps; e.n = n; e.rows = R; e.Cin = d->input_dim; e.ring_len = 0; e.gate = 0;
  hipLaunchKernelGGL(gen_conv_kernel, dim3(cdiv(R, 4)), dim3(256), 0, st, e);
  VQ_LAUNCH_CHECK();

  float* h_in = d->h0;
  float* h_out = d->h1;
  for (int l = 0; l < d->n_blocks; ++l) {
    const vqvae_gen_block& bk = d->blocks[l];
    VQ_REQUIRE(bk.conv_W && bk.conv_b && bk.res_W && bk.res_b && bk.skip_W && bk.skip_b && bk.ring && bk.dilation >= 1,
               "gen_step: block %d incomplete", l);
    VQ_REQUIRE(d->cond_dim == 0 || (bk.cond_W && bk.cond_b), "gen_step: block %d has no condition projection", l);
    GenConvArgs c{};
    c.W = bk.conv_W; c.b = bk.conv_b; c.Wc = d->cond_dim ? bk.cond_W : nullptr; c.bc = bk.cond_b;
    c.x_old = bk.ring; c.x_new = h_in; c.cond = d->cond; c.cond_bstride = d->cond_bstride; c.cond_cstride = d->cond_cstride;
    c.cond_step = d->cond_follows_step; c.out = d->z; c.step = d->step; c.max_steps = d->max_steps;
    c.n = n; c.rows = d->dilated; c.Cin = R; c.Cc = d->cond_dim; c.ring_len = bk.dilation; c.gate = 1;
    hipLaunchKernelGGL(gen_conv_kernel, dim3(cdiv(half, 4)), dim3(256), 0, st, c);
    VQ_LAUNCH_CHECK();

    GenDenseArgs g{};
    g.job[0] = GenDenseJob{bk.res_W, bk.res_b, d->z, h_in, h_out, R, half, 0};                       // modules.py:54
    g.job[1] = GenDenseJob{bk.skip_W, bk.skip_b, d->z, nullptr, d->skip_acc, d->skip, half, l ? GEN_ACCUM : 0};
    g.njobs = 2; g.n = n; g.step = d->step; g.max_steps = d->max_steps;
    g.push_src = h_in; g.ring = bk.ring; g.ring_len = bk.dilation; g.push_elems = n * R;
    if (l == 0) { g.shift_src = d->x_cur; g.shift_dst = d->x_prev; g.shift_elems = n * d->input_dim; }
    hipLaunchKernelGGL(gen_dense_kernel, dim3(cdiv(R + d->skip, 4)), dim3(256), 0, st, g);
    VQ_LAUNCH_CHECK();
    float* tmp = h_in; h_in = h_out; h_out = tmp;
  }
  GenDenseArgs p1{};
  p1.job[0] = GenDenseJob{d->proj1_W, d->proj1_b, d->skip_acc, nullptr, d->s1, d->skip, d->skip, GEN_RELU_IN | GEN_RELU_OUT};
  p1.njobs = 1; p1.n = n; p1.step = d->step; p1.max_steps = d->max_steps;
  hipLaunchKernelGGL(gen_dense_kernel, dim3(cdiv(d->skip, 4)), dim3(256), 0, st, p1);
  VQ_LAUNCH_CHECK();
  GenDenseArgs p2{};
  p2.job[0] = GenDenseJob{d->proj2_W, d->proj2_b, d->s1, nullptr, d->logits, d->out_dim, d->skip, 0};
  p2.njobs = 1; p2.n = n; p2.step = d->step; p2.max_steps = d->max_steps;
  hipLaunchKernelGGL(gen_dense_kernel, dim3(cdiv(d->out_dim, 4)), dim3(256), 0, st, p2);
  VQ_LAUNCH_CHECK();

  GenFinishArgs f{};
  f.logits = d->logits; f.uniforms = d->uniforms; f.forced = d->forced_next; f.out = d->out; f.out_bstride = d->out_bstride;
  f.logits_out = d->logits_out; f.x_cur = d->x_cur; f.step = d->step; f.max_steps = d->max_steps; f.n = n;
  f.out_dim = d->out_dim; f.input_dim = d->input_dim; f.mode = d->sample_mode; f.n_uniform = d->n_uniform;
  f.log_scale_min = d->log_scale_min;
  hipLaunchKernelGGL(gen_finish_kernel, dim3(1), dim3(256), 0, st, f);
  VQ_LAUNCH_CHECK();
  return 0;
}

// =================================================================================================
// Persistent generation kernel ("megakernel"): the whole loop of generate.py:105-145 in ONE launch.
//
// A step is 2 * n_blocks + 5 all-to-all dependency edges.  As separate kernels an edge costs a
// dependent launch boundary + cold loads (~4.7-7 us measured); inside one launch it is a hand-off
// through tagged 8-byte granules {float, step tag} written with one agent-scope store and gathered
// by spinning loads (~1.8 us measured, tools/ubench/edge_latency.hip), and the weights stay L2-hot
// because the launch never ends.
//   * MEGA_G workgroups of 2 waves.  Wave 0 of every workgroup walks the dependency chain for the
//     rows it owns (row r belongs to workgroup r % MEGA_G: one gate pair, two rows of every 256-row
//     matrix -- a single wave issues one VALU instruction per 4 cycles, so the chain's instruction
//     count IS its latency and is kept to a few hundred per stage); wave 1 computes, a step ahead
//     and off the critical path, everything of the gate pre-activation that does not depend on
//     the current step: old tap W[:,:,0] x[t-d] (from the queue), condition projection and both
//     biases, and hands it over through LDS (ready / consumed tags).
//   * Every block's queue (modules.py:58-62) is a ring of d+1 granule vectors: slot t % (d+1) is
//     written at step t and read as the old tap at step t + d.
//   * All spins are bounded: on timeout the status word is set and every wave leaves.
//   * One edge per block instead of two.  The reference's chain is x_l -> z_l -> x_{l+1} -> z_{l+1}
//     (two all-to-all hand-offs per block).  Because x_{l+1} = Wr_l z_l + br_l + x_l is linear, the
//     new-tap term of block l+1 is  Wc1_{l+1} x_{l+1} = (Wc1_{l+1} Wr_l) z_l + Wc1_{l+1} x_l + Wc1_{l+1} br_l:
//     the composite matrix M_l = Wc1_{l+1} Wr_l (256 x 128, built once in float64 by the pack kernel)
//     takes z_l straight to the next gate, and the term in x_l -- already on its way while z_l is
//     being computed -- is evaluated in the shadow of the z_l hand-off.  The critical path is
//     z_0 -> z_1 -> ... -> z_{L-1}; x_{l+1} is still produced (queues, next block's shadow term) but
//     nobody waits for it.  Same mathematics, different rounding order (~1e-6 relative on the
//     pre-activations; the parity tests hold the 1e-4 fp32 bar against the oracle's plain order).
// The global sequential dependency (nothing of step t+1 can be produced before the sample of step
// t, which needs every workgroup's step-t rows) makes single-buffered mailboxes race-free.
// =================================================================================================
namespace vq {

constexpr int MEGA_G = 128;         // workgroups; rows are dealt round-robin
constexpr int MEGA_RJ = 2;          // rows per workgroup per 256-row matrix
constexpr int MEGA_PJ = 1;          // gate pairs per workgroup
constexpr int MEGA_NHELP = 1;       // helper waves per workgroup
constexpr int MEGA_THREADS = 64 * (1 + MEGA_NHELP);
constexpr int MEGA_VI = 4;          // vector elements per lane (<= 256 channels)
constexpr int MEGA_MAXL = 64;
constexpr int MEGA_SPIN_LIMIT = 1 << 22;

struct MegaBlock {
  const float *conv_W, *conv_b, *cond_W, *cond_b, *res_W, *res_b, *skip_W, *skip_b;
  long qoff;          // granule offset of this block's queue inside xq
  int dilation, pad_;
};

struct MegaQ { long qoff; int dilation, pad_; };     // what the step loop needs of a block besides its weights

struct MegaArgs {
  const MegaBlock* blk;
  const MegaQ* qt;
  const float4* crec;    // chain-wave weight records  [L][MEGA_G][64 lanes][4 x float4]
  const float4* hrec;    // helper-wave weight records [L][MEGA_G][64 lanes][4 x float4]
  const float4* brec;    // biases [L][MEGA_G][2 x float4]: {bR0, bR1, bS0, bS1}, {conv_b+cond_b of the pair, 0, 0}
  const float *embed_W, *embed_b, *proj1_W, *proj1_b, *proj2_W, *proj2_b;
  const float* cond; long cond_bstride, cond_cstride;
  const double* uniforms; const void* forced; void* out; long out_bstride; float* logits_out;
  unsigned long long *xq, *zbox, *sbox, *s1box, *lgbox, *idxbox;
  int* status;
  int L, n, input_dim, R, D, S, Cc, O, mode, n_uniform, t0, steps;
  float log_scale_min;
};

typedef unsigned long long gran_t;

__device__ __forceinline__ void g_put(gran_t* p, float v, uint32_t tag) {
  __hip_atomic_store(p, ((gran_t)tag << 32) | (gran_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ gran_t g_get(const gran_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool mega_aborted(int* status) {
  return __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
__device__ __forceinline__ void mega_abort(int* status, int code) {
  __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Gathers K (<= 64*MEGA_VI) granules carrying `tag`: element k = lane + 64 i lands in v[i].
__device__ __forceinline__ bool mega_gather(const gran_t* box, int K, uint32_t tag, float (&v)[MEGA_VI], int* status) {
  const int lane = threadIdx.x & 63;
  unsigned pending = 0;
#pragma unroll
  for (int i = 0; i < MEGA_VI; ++i) { v[i] = 0.f; if (lane + 64 * i < K) pending |= 1u << i; }
  int spins = 0;
  for (;;) {
    gran_t g[MEGA_VI];
#pragma unroll
    for (int i = 0; i < MEGA_VI; ++i) if (pending & (1u << i)) g[i] = g_get(box + lane + 64 * i);
#pragma unroll
    for (int i = 0; i < MEGA_VI; ++i)
      if ((pending & (1u << i)) && (uint32_t)(g[i] >> 32) == tag) { v[i] = __uint_as_float((uint32_t)g[i]); pending &= ~(1u << i); }
    if (__all(pending == 0)) return true;
    if ((++spins & 255) == 0) {
      if (mega_aborted(status)) return false;
      if (spins > MEGA_SPIN_LIMIT) { mega_abort(status, 1); return false; }
    }
  }
}

__device__ __forceinline__ bool lds_wait_eq(int* flag, int want, int* status) {
  int spins = 0;
  // LDS is coherent inside the CU and a wave's LDS operations execute in order: relaxed accesses
  // plus compiler fences are enough (a release/acquire here would also drain the global stores).
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != want) {
    __builtin_amdgcn_s_sleep(1);
    if ((++spins & 1023) == 0) {
      if (mega_aborted(status)) return false;
      if (spins > (MEGA_SPIN_LIMIT << 2)) { mega_abort(status, 2); return false; }
    }
  }
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  return true;
}
__device__ __forceinline__ void lds_post(int* flag, int v) {
  __atomic_signal_fence(__ATOMIC_SEQ_CST);
  __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ double wave_incl_scan(double v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  return v;
}

#ifdef MEGA_PROF
#define MP_T(i) do { const long long _n = wall_clock64(); prof_r[i] += _n - mp_last; mp_last = _n; } while (0)
#else
#define MP_T(i) do { } while (0)
#endif

// Wave-wide sum, result in every lane: four DPP adds inside each row of 16 lanes (quad_perm
// [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then the four row totals through SGPRs.
// ~12 instructions instead of six dependent ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float mega_wave_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float mega_gate(float h0, float h1) {          // tanh(h0) * sigmoid(h1)
  const float th = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * h0));
  return th * __builtin_amdgcn_rcpf(1.f + __expf(-h1));
}

// Weights and biases of one residual block for the rows this workgroup owns, as the chain wave
// holds them in registers (element k = lane + 64 i of a row).  Loaded one block ahead.
typedef const __attribute__((address_space(1))) float* gfp_t;    // pointers read from the block table are
#define GF(p) ((gfp_t)(p))                                          // generic: make the loads global_load again

struct MegaBlkW {                    // record l: what the chain wave needs once z_l has arrived
  float wU[2][MEGA_VI];              // conv tap 1 of block l+1, its gate pair rows (shadow term Wc1_{l+1} x_l)
  float wM[2][2];                    // M_l = Wc1_{l+1} Wr_l, the same two rows, k = lane + 64 i
  float wR[MEGA_RJ][2], wS[MEGA_RJ][2];
  float bR[MEGA_RJ], bS[MEGA_RJ], cb[2];   // cb = Wc1_{l+1} br_l
};
constexpr int MEGA_CREC = 5;         // float4 per lane in a chain record
static_assert(MEGA_PJ == 1 && MEGA_RJ == 2 && MEGA_VI == 4, "the packed weight records are laid out for 1 pair / 2 rows / 4 elements per lane");
__device__ __forceinline__ void mega_load_blk(MegaBlkW& w, const MegaArgs& a, int l, int g, int lane) {
  const float4* rec = a.crec + ((size_t)(l * MEGA_G + g) * 64 + lane) * MEGA_CREC;
  const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4];
  const float4 bq = a.brec[(size_t)(l * MEGA_G + g) * 2], bc = a.brec[(size_t)(l * MEGA_G + g) * 2 + 1];
  w.wU[0][0] = r0.x; w.wU[0][1] = r0.y; w.wU[0][2] = r0.z; w.wU[0][3] = r0.w;
  w.wU[1][0] = r1.x; w.wU[1][1] = r1.y; w.wU[1][2] = r1.z; w.wU[1][3] = r1.w;
  w.wM[0][0] = r2.x; w.wM[0][1] = r2.y; w.wM[1][0] = r2.z; w.wM[1][1] = r2.w;
  w.wR[0][0] = r3.x; w.wR[0][1] = r3.y; w.wR[1][0] = r3.z; w.wR[1][1] = r3.w;
  w.wS[0][0] = r4.x; w.wS[0][1] = r4.y; w.wS[1][0] = r4.z; w.wS[1][1] = r4.w;
  w.bR[0] = bq.x; w.bR[1] = bq.y; w.bS[0] = bq.z; w.bS[1] = bq.w;
  w.cb[0] = bc.z; w.cb[1] = bc.w;
}

// One-off repack of the Chainer-layout weights into per-(block, workgroup, lane) records (four or
// five coalesced 16-byte loads per lane fetch a block), including the composite matrices.
__global__ __launch_bounds__(64) void gen_mega_pack_kernel(const MegaBlock* blk, int L, int R, int half, int S, int Cc,
                                                           float4* crec, float4* hrec, float4* brec) {
  const int l = blockIdx.x / MEGA_G, g = blockIdx.x % MEGA_G, lane = threadIdx.x;
  const MegaBlock bk = blk[l];
  const bool last = l == L - 1;
  const MegaBlock bn = blk[last ? l : l + 1];
  float c[20], h[16];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = lane + 64 * i;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int row = g + hh * half;
      const bool ok = g < half && k < R;
      c[hh * 4 + i] = (ok && !last) ? bn.conv_W[((size_t)row * R + k) * 2 + 1] : 0.f;   // tap 1 of block l+1
      h[hh * 4 + i] = ok ? bk.conv_W[((size_t)row * R + k) * 2] : 0.f;                  // tap 0 of block l
      h[8 + hh * 4 + i] = (g < half && k < Cc) ? bk.cond_W[(size_t)row * Cc + k] : 0.f;
    }
  }
  double cbv[2] = {0.0, 0.0};
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int row = g + hh * half;
    double m0 = 0.0, m1 = 0.0;
    if (!last && g < half) {
      for (int cc = 0; cc < R; ++cc) {
        const double w = (double)bn.conv_W[((size_t)row * R + cc) * 2 + 1];
        if (lane < half) m0 += w * (double)bk.res_W[(size_t)cc * half + lane];
        if (lane + 64 < half) m1 += w * (double)bk.res_W[(size_t)cc * half + lane + 64];
        cbv[hh] += w * (double)bk.res_b[cc];
      }
    }
    c[8 + hh * 2] = (float)m0;
    c[8 + hh * 2 + 1] = (float)m1;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = g + MEGA_G * j, k = lane + 64 * i;
      c[12 + j * 2 + i] = (!last && r < R && k < half) ? bk.res_W[(size_t)r * half + k] : 0.f;
      c[16 + j * 2 + i] = (r < S && k < half) ? bk.skip_W[(size_t)r * half + k] : 0.f;
    }
  float4* cr = crec + ((size_t)blockIdx.x * 64 + lane) * MEGA_CREC;
  float4* hr = hrec + ((size_t)blockIdx.x * 64 + lane) * 4;
#pragma unroll
  for (int q = 0; q < MEGA_CREC; ++q) cr[q] = make_float4(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3]);
#pragma unroll
  for (int q = 0; q < 4; ++q) hr[q] = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
  if (lane == 0) {
    const int r0 = g, r1 = g + MEGA_G;
    brec[(size_t)blockIdx.x * 2] = make_float4((!last && r0 < R) ? bk.res_b[r0] : 0.f, (!last && r1 < R) ? bk.res_b[r1] : 0.f,
                                               r0 < S ? bk.skip_b[r0] : 0.f, r1 < S ? bk.skip_b[r1] : 0.f);
    brec[(size_t)blockIdx.x * 2 + 1] = make_float4(g < half ? bk.conv_b[g] + bk.cond_b[g] : 0.f,
                                                   g < half ? bk.conv_b[g + half] + bk.cond_b[g + half] : 0.f,
                                                   (float)cbv[0], (float)cbv[1]);
  }
}

// All sequences' vectors of one mailbox in a single polling loop (element k = lane + 64 i of sequence b).
template <int NB>
__device__ __forceinline__ bool mega_gather_n(const gran_t* box, size_t bstride, int n, int K, uint32_t tag,
                                              float (&v)[NB][MEGA_VI], int* status) {
  const int lane = threadIdx.x & 63;
  unsigned pending = 0;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < MEGA_VI; ++i) { v[b][i] = 0.f; if (b < n && lane + 64 * i < K) pending |= 1u << (b * MEGA_VI + i); }
  int spins = 0;
  for (;;) {
    gran_t gr[NB][MEGA_VI];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < MEGA_VI; ++i) if (pending & (1u << (b * MEGA_VI + i))) gr[b][i] = g_get(box + b * bstride + lane + 64 * i);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int i = 0; i < MEGA_VI; ++i) {
        const unsigned bit = 1u << (b * MEGA_VI + i);
        if ((pending & bit) && (uint32_t)(gr[b][i] >> 32) == tag) { v[b][i] = __uint_as_float((uint32_t)gr[b][i]); pending &= ~bit; }
      }
    if (__all(pending == 0)) return true;
    if ((++spins & 255) == 0) {
      if (mega_aborted(status)) return false;
      if (spins > MEGA_SPIN_LIMIT) { mega_abort(status, 1); return false; }
    }
  }
}

// NB = sequences the code is unrolled for; EXACT: n == NB is known at compile time (no per-sequence tests)
template <int NB, bool EXACT>
__global__ __launch_bounds__(MEGA_THREADS) void gen_mega_kernel(MegaArgs a) {
  __shared__ float pre_s[MEGA_MAXL][MEGA_PJ][2][NB];
  __shared__ int ready_s[MEGA_MAXL];
  __shared__ int consumed_s[MEGA_MAXL];
  __shared__ float lg_s[NB][64 * MEGA_VI];
  const int g = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = EXACT ? NB : a.n, R = a.R, half = a.D >> 1, S = a.S, O = a.O;
  for (int i = threadIdx.x; i < MEGA_MAXL; i += MEGA_THREADS) { ready_s[i] = a.t0; consumed_s[i] = a.t0; }
  __syncthreads();

  if (wave != 0) {
    // ---------------- helper waves: step-independent part of the gate pre-activation -------------
    for (int s = a.t0; s < a.t0 + a.steps; ++s) {
      float cv[NB][MEGA_VI];
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < MEGA_VI; ++i) {
          const int k = lane + 64 * i;
          cv[b][i] = (b < n && k < a.Cc) ? a.cond[b * a.cond_bstride + k * a.cond_cstride + s] : 0.f;
        }
      for (int l = wave - 1; l < a.L; l += MEGA_NHELP) {
        const MegaQ bk = a.qt[l];
        const int d = bk.dilation;
        // weights and biases first: they do not depend on anything
        float w0[MEGA_PJ][2][MEGA_VI], wc[MEGA_PJ][2][MEGA_VI], bb[MEGA_PJ][2];
        {
          const float4* rec = a.hrec + ((size_t)(l * MEGA_G + g) * 64 + lane) * 4;
          const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
          const float4 bq = a.brec[(size_t)(l * MEGA_G + g) * 2 + 1];
          w0[0][0][0] = r0.x; w0[0][0][1] = r0.y; w0[0][0][2] = r0.z; w0[0][0][3] = r0.w;
          w0[0][1][0] = r1.x; w0[0][1][1] = r1.y; w0[0][1][2] = r1.z; w0[0][1][3] = r1.w;
          wc[0][0][0] = r2.x; wc[0][0][1] = r2.y; wc[0][0][2] = r2.z; wc[0][0][3] = r2.w;
          wc[0][1][0] = r3.x; wc[0][1][1] = r3.y; wc[0][1][2] = r3.z; wc[0][1][3] = r3.w;
          bb[0][0] = bq.x; bb[0][1] = bq.y;
        }
        float xo[NB][MEGA_VI];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
          for (int i = 0; i < MEGA_VI; ++i) xo[b][i] = 0.f;
          if (b < n && s - d >= 0) {
            const gran_t* q = a.xq + bk.qoff + ((size_t)((s - d) % (d + 1)) * n + b) * R;
            if (!mega_gather(q, R, (uint32_t)(s - d + 1), xo[b], a.status)) return;
          }
        }
        if (!lds_wait_eq(&consumed_s[l], s, a.status)) return;
#pragma unroll
        for (int j = 0; j < MEGA_PJ; ++j)
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
              if (b >= n || g + MEGA_G * j >= half) continue;
              float acc = 0.f;
#pragma unroll
              for (int i = 0; i < MEGA_VI; ++i) acc += w0[j][h][i] * xo[b][i] + wc[j][h][i] * cv[b][i];
              acc = mega_wave_sum(acc) + bb[j][h];
              if (lane == 0) pre_s[l][j][h][b] = acc;
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): pre_s writes have reached the LDS
        if (lane == 0) lds_post(&ready_s[l], s + 1);
      }
    }
    return;
  }

  // ---------------- wave 0: the dependency chain ------------------------------------------------
#ifdef MEGA_PROF
  long long prof_r[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long mp_last = wall_clock64();
#endif
  // feedback values of the two previous steps = the embed queue (modules.py:236-237, 247)
  int ip[NB], ic[NB];
  float fp[NB], fc[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    ip[b] = ic[b] = -1; fp[b] = fc[b] = 0.f;
    if (b >= n) continue;
    for (int back = 1; back <= 2; ++back) {
      const int ts = a.t0 - back;
      if (ts < 0) continue;
      int iv = -1; float fv = 0.f;
      if (a.mode == VQVAE_GEN_SOFTMAX)
        iv = a.forced ? reinterpret_cast<const int32_t*>(a.forced)[(size_t)ts * n + b] : reinterpret_cast<const int32_t*>(a.out)[b * a.out_bstride + ts];
      else
        fv = a.forced ? reinterpret_cast<const float*>(a.forced)[(size_t)ts * n + b] : reinterpret_cast<const float*>(a.out)[b * a.out_bstride + ts];
      if (back == 1) { ic[b] = iv; fc[b] = fv; } else { ip[b] = iv; fp[b] = fv; }
    }
  }
  // step-invariant registers: embed bias, head weights and biases, block 0's tap-1 rows
  float bE[MEGA_RJ], b1[MEGA_RJ], b2[MEGA_RJ], w1[MEGA_RJ][MEGA_VI], w2[MEGA_RJ][MEGA_VI], wU0[2][MEGA_VI], bEf[MEGA_VI];
  {
    const MegaBlock bk0 = a.blk[0];
#pragma unroll
    for (int i = 0; i < MEGA_VI; ++i) {
      const int k = lane + 64 * i;
      bEf[i] = k < R ? a.embed_b[k] : 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) wU0[h][i] = (g < half && k < R) ? GF(bk0.conv_W)[((size_t)(g + h * half) * R + k) * 2 + 1] : 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < MEGA_RJ; ++j) {
    const int r = g + MEGA_G * j;
    bE[j] = r < R ? a.embed_b[r] : 0.f;
    b1[j] = r < S ? a.proj1_b[r] : 0.f;
    b2[j] = r < O ? a.proj2_b[r] : 0.f;
#pragma unroll
    for (int i = 0; i < MEGA_VI; ++i) {
      const int k = lane + 64 * i;
      w1[j][i] = (r < S && k < S) ? a.proj1_W[(size_t)r * S + k] : 0.f;
      w2[j][i] = (r < O && k < S) ? a.proj2_W[(size_t)r * S + k] : 0.f;
    }
  }
  MegaBlkW cur, nxt;
  mega_load_blk(cur, a, 0, g, lane);

  for (int t = a.t0; t < a.t0 + a.steps; ++t) {
    const uint32_t tag = (uint32_t)t + 1u;
    float xown[MEGA_RJ][NB], sacc[MEGA_RJ][NB], xf[NB][MEGA_VI];
    // ---- embed (modules.py:247-248) on [input(t-1), input(t)]: a function of the two fed-back
    //      values only, so every workgroup rebuilds the whole x_0 itself -- no hand-off ----
    {
      const MegaQ bk0 = a.qt[0];
      gran_t* q = a.xq + bk0.qoff + (size_t)(t % (bk0.dilation + 1)) * n * R;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int i = 0; i < MEGA_VI; ++i) {
          const int k = lane + 64 * i;
          float e0 = 0.f, e1 = 0.f;
          if (b < n && k < R) {
            if (a.mode == VQVAE_GEN_SOFTMAX) {
              if (ip[b] >= 0) e0 = a.embed_W[((size_t)k * a.input_dim + ip[b]) * 2];
              if (ic[b] >= 0) e1 = a.embed_W[((size_t)k * a.input_dim + ic[b]) * 2 + 1];
            } else {
              for (int c = 0; c < a.input_dim; ++c) {
                e0 += a.embed_W[((size_t)k * a.input_dim + c) * 2] * fp[b];
                e1 += a.embed_W[((size_t)k * a.input_dim + c) * 2 + 1] * fc[b];
              }
            }
          }
          xf[b][i] = (b < n && k < R) ? (e0 + e1) + bEf[i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < MEGA_RJ; ++j) {
          const int r = g + MEGA_G * j;             // x_0[r] sits in lane r % 64, element r / 64
          float v = 0.f;
#pragma unroll
          for (int i = 0; i < MEGA_VI; ++i) if (r / 64 == i) v = __shfl(xf[b][i], r & 63, 64);
          xown[j][b] = v; sacc[j][b] = 0.f;
          if (r < R && b < n && lane == 0) g_put(q + (size_t)b * R + r, v, tag);      // queue entry for the helpers
        }
      }
    }
    // ---- z_0: gate of block 0 (modules.py:40-48) ----
    if (!lds_wait_eq(&ready_s[0], t + 1, a.status)) return;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (b >= n || g >= half) continue;
      float h0 = 0.f, h1 = 0.f;
#pragma unroll
      for (int i = 0; i < MEGA_VI; ++i) { h0 += wU0[0][i] * xf[b][i]; h1 += wU0[1][i] * xf[b][i]; }
      h0 = mega_wave_sum(h0) + pre_s[0][0][0][b];
      h1 = mega_wave_sum(h1) + pre_s[0][0][1][b];
      if (lane == 0) g_put(a.zbox + (size_t)b * half + g, mega_gate(h0, h1), tag);
    }
    if (lane == 0) lds_post(&consumed_s[0], t + 1);
    MP_T(0);
    // ---- residual blocks (modules.py:102-110), one critical hand-off each: z_l ----
    for (int l = 0; l < a.L; ++l) {
      const bool has_next = l + 1 < a.L;
      mega_load_blk(nxt, a, has_next ? l + 1 : 0, g, lane);       // lands while z_l is awaited
      // shadow of the z_l hand-off: the x_l term of the next gate, u = Wc1_{l+1} x_l
      float u[2][NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) u[0][b] = u[1][b] = 0.f;
      if (has_next) {
        if (l > 0) {
          const MegaQ bk = a.qt[l];
          const gran_t* q = a.xq + bk.qoff + (size_t)(t % (bk.dilation + 1)) * n * R;
          if (!mega_gather_n<NB>(q, R, n, R, tag, xf, a.status)) return;
        }
        MP_T(1);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b >= n || g >= half) continue;
          float u0 = 0.f, u1 = 0.f;
#pragma unroll
          for (int i = 0; i < MEGA_VI; ++i) { u0 += cur.wU[0][i] * xf[b][i]; u1 += cur.wU[1][i] * xf[b][i]; }
          u[0][b] = mega_wave_sum(u0);
          u[1][b] = mega_wave_sum(u1);
        }
      }
      MP_T(2);
      // the critical hand-off
      float zv[NB][MEGA_VI];
      if (!mega_gather_n<NB>(a.zbox + (size_t)l * n * half, half, n, half, tag, zv, a.status)) return;
      MP_T(3);
      if (has_next) {
        // own rows of x_{l+1} = Wr z + br + x_l (modules.py:52-54): published first, they are cheap
        // and every workgroup needs them for its next shadow term; also the queue push (71-74)
        const MegaQ bn = a.qt[l + 1];
        gran_t* qn = a.xq + bn.qoff + (size_t)(t % (bn.dilation + 1)) * n * R;
#pragma unroll
        for (int j = 0; j < MEGA_RJ; ++j) {
          const int r = g + MEGA_G * j;
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            if (b >= n || r >= R) continue;
            const float dot = mega_wave_sum(cur.wR[j][0] * zv[b][0] + cur.wR[j][1] * zv[b][1]);
            xown[j][b] = (dot + cur.bR[j]) + xown[j][b];
            if (lane == 0) g_put(qn + (size_t)b * R + r, xown[j][b], tag);
          }
        }
        // z_{l+1} through the composite matrix
        if (!lds_wait_eq(&ready_s[l + 1], t + 1, a.status)) return;
        gran_t* zb = a.zbox + (size_t)(l + 1) * n * half;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b >= n || g >= half) continue;
          float h0 = mega_wave_sum(cur.wM[0][0] * zv[b][0] + cur.wM[0][1] * zv[b][1]);
          float h1 = mega_wave_sum(cur.wM[1][0] * zv[b][0] + cur.wM[1][1] * zv[b][1]);
          h0 = ((h0 + u[0][b]) + cur.cb[0]) + pre_s[l + 1][0][0][b];
          h1 = ((h1 + u[1][b]) + cur.cb[1]) + pre_s[l + 1][0][1][b];
          if (lane == 0) g_put(zb + (size_t)b * half + g, mega_gate(h0, h1), tag);
        }
        if (lane == 0) lds_post(&consumed_s[l + 1], t + 1);
      }
      MP_T(4);
      // skip rows (modules.py:55, 105-109): nobody waits for them before the head
#pragma unroll
      for (int j = 0; j < MEGA_RJ; ++j) {
        const int r = g + MEGA_G * j;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (b >= n || r >= S) continue;
          const float sk = mega_wave_sum(cur.wS[j][0] * zv[b][0] + cur.wS[j][1] * zv[b][1]) + cur.bS[j];
          sacc[j][b] = l == 0 ? sk : sacc[j][b] + sk;
        }
      }
      cur = nxt;
      MP_T(5);
    }
    // ---- head: relu, proj1, relu, proj2 (modules.py:249-255) ----
#pragma unroll
    for (int j = 0; j < MEGA_RJ; ++j) {
      const int r = g + MEGA_G * j;
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (r < S && b < n && lane == 0) g_put(a.sbox + (size_t)b * S + r, fmaxf(sacc[j][b], 0.f), tag);
    }
    float sv[NB][MEGA_VI];
    if (!mega_gather_n<NB>(a.sbox, S, n, S, tag, sv, a.status)) return;
    MP_T(6);
#pragma unroll
    for (int j = 0; j < MEGA_RJ; ++j) {
      const int r = g + MEGA_G * j;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b >= n || r >= S) continue;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < MEGA_VI; ++i) acc += w1[j][i] * sv[b][i];
        acc = mega_wave_sum(acc) + b1[j];
        if (lane == 0) g_put(a.s1box + (size_t)b * S + r, fmaxf(acc, 0.f), tag);
      }
    }
    if (!mega_gather_n<NB>(a.s1box, S, n, S, tag, sv, a.status)) return;
    MP_T(7);
#pragma unroll
    for (int j = 0; j < MEGA_RJ; ++j) {
      const int r = g + MEGA_G * j;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b >= n || r >= O) continue;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < MEGA_VI; ++i) acc += w2[j][i] * sv[b][i];
        acc = mega_wave_sum(acc) + b2[j];
        if (lane == 0) g_put(a.lgbox + (size_t)b * O + r, acc, tag);
      }
    }
    MP_T(8);
    // ---- sampler on workgroup 0 (generate.py:109-141) ----
    if (g == 0) {
      for (int b = 0; b < n; ++b) {
        float lv[MEGA_VI];
        if (!mega_gather(a.lgbox + (size_t)b * O, O, tag, lv, a.status)) return;
        if (a.logits_out)
#pragma unroll
          for (int i = 0; i < MEGA_VI; ++i)
            if (lane + 64 * i < O) a.logits_out[((size_t)t * n + b) * O + lane + 64 * i] = lv[i];
        float fb = 0.f;
        if (a.mode == VQVAE_GEN_SOFTMAX) {
          float m = -INFINITY;
#pragma unroll
          for (int i = 0; i < MEGA_VI; ++i) if (lane + 64 * i < O) m = fmaxf(m, lv[i]);
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
          float e[MEGA_VI], s = 0.f;
#pragma unroll
          for (int i = 0; i < MEGA_VI; ++i) { e[i] = lane + 64 * i < O ? expf(lv[i] - m) : 0.f; s += e[i]; }
          s = gen_wave_sum(s);
          // float64 cdf in index order k = lane + 64 i: full rows below + an in-row scan
          const double u = a.uniforms[((size_t)t * n + b) * a.n_uniform];
          double base = 0.0, cdf[MEGA_VI];
#pragma unroll
          for (int i = 0; i < MEGA_VI; ++i) {
            const double pk = (double)(e[i] / s);
            const double sc = wave_incl_scan(pk);
            cdf[i] = base + sc;
            base += __shfl(sc, 63, 64);
          }
          const double total = base;
          int cnt = 0;
#pragma unroll
          for (int i = 0; i < MEGA_VI; ++i) if (lane + 64 * i < O && cdf[i] / total <= u) ++cnt;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
          const int idx = cnt < O ? cnt : O - 1;
          if (lane == 0 && a.out) reinterpret_cast<int32_t*>(a.out)[b * a.out_bstride + t] = idx;
          const int nxt_in = a.forced ? reinterpret_cast<const int32_t*>(a.forced)[(size_t)t * n + b] : idx;
          fb = __int_as_float(nxt_in);
        } else {
#pragma unroll
          for (int i = 0; i < MEGA_VI; ++i) lg_s[b][lane + 64 * i] = lv[i];
          __builtin_amdgcn_wave_barrier();
          float v = 0.f;
          if (lane == 0) {
            const int nr = O / 3;
            const float* lg = lg_s[b];
            float m = -INFINITY;
            for (int j = 0; j < nr; ++j) m = fmaxf(m, lg[j]);
            float s = 0.f;
            for (int j = 0; j < nr; ++j) s += expf(lg[j] - m);
            double acc = 0.0;
            for (int j = 0; j < nr; ++j) {
              const float pj = expf(lg[j] - m) / s;
              const float sc = expf(fmaxf(lg[2 * nr + j], a.log_scale_min));
              const double u = a.uniforms[((size_t)t * n + b) * a.n_uniform + j];
              acc += ((double)lg[nr + j] + (double)sc * (log(u) - log(1.0 - u))) * (double)pj;
            }
            v = (float)acc;
            v = v / 127.5f;
            v = fminf(fmaxf(v, -1.f), 1.f);
            if (a.out) reinterpret_cast<float*>(a.out)[b * a.out_bstride + t] = v;
            if (a.forced) v = reinterpret_cast<const float*>(a.forced)[(size_t)t * n + b];
          }
          fb = __shfl(v, 0, 64);
        }
        if (lane == 0) g_put(a.idxbox + b, fb, tag);
      }
    }
    MP_T(9);
    // ---- feedback: every workgroup learns the next input (generate.py:127, 139-141) ----
    {
      float fv[MEGA_VI];
      if (!mega_gather(a.idxbox, n, tag, fv, a.status)) return;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b >= n) continue;
        const float f = __shfl(fv[0], b, 64);
        ip[b] = ic[b]; fp[b] = fc[b];
        ic[b] = __float_as_int(f); fc[b] = f;
      }
    }
    MP_T(10);
  }
#ifdef MEGA_PROF
  if (g == 0 && lane == 0) {
    long long* prof = reinterpret_cast<long long*>(a.status) + 8;   // 16 counters inside the status block
#pragma unroll
    for (int i = 0; i < 11; ++i) prof[i] += prof_r[i];
  }
#endif
}

}  // namespace vq

static size_t mega_layout(const vqvae_gen_desc* d, size_t* off_tab, size_t* off_xq, size_t* off_z, size_t* off_s,
                          size_t* off_s1, size_t* off_lg, size_t* off_idx, size_t* off_qt, size_t* off_crec,
                          size_t* off_hrec, size_t* off_brec, long* qoffs) {
  size_t o = 256;                                             // status word
  *off_tab = o; o += vq::align_up(sizeof(vq::MegaBlock) * d->n_blocks, 256);
  size_t q = 0;
  for (int l = 0; l < d->n_blocks; ++l) {
    if (qoffs) qoffs[l] = (long)q;
    q += (size_t)(d->blocks[l].dilation + 1) * d->n * d->residual;
  }
  *off_xq = o; o += vq::align_up(q * 8, 256);
  *off_z = o; o += vq::align_up((size_t)d->n_blocks * d->n * (d->dilated / 2) * 8, 256);
  *off_s = o; o += vq::align_up((size_t)d->n * d->skip * 8, 256);
  *off_s1 = o; o += vq::align_up((size_t)d->n * d->skip * 8, 256);
  *off_lg = o; o += vq::align_up((size_t)d->n * d->out_dim * 8, 256);
  *off_idx = o; o += 256;
  *off_qt = o; o += vq::align_up(sizeof(vq::MegaQ) * d->n_blocks, 256);
  const size_t recs = (size_t)d->n_blocks * vq::MEGA_G;
  *off_crec = o; o += recs * 64 * 16 * vq::MEGA_CREC;
  *off_hrec = o; o += recs * 64 * 64;
  *off_brec = o; o += vq::align_up(recs * 32, 256);
  return o;
}

static int mega_check(const vqvae_gen_desc* d) {
  VQ_REQUIRE(d, "gen_run: null descriptor");
  VQ_REQUIRE(d->n >= 1 && d->n <= GEN_NB, "gen_run: n must be 1..%d sequences (got %d)", GEN_NB, d->n);
  VQ_REQUIRE(d->n_blocks >= 1 && d->n_blocks <= MEGA_MAXL && d->blocks, "gen_run: 1..%d residual blocks", MEGA_MAXL);
  const int lim = 64 * MEGA_VI;
  VQ_REQUIRE(d->residual >= 1 && d->residual <= lim && d->skip >= 1 && d->skip <= lim && d->out_dim >= 1 && d->out_dim <= lim &&
             d->cond_dim >= 1 && d->cond_dim <= lim && d->dilated >= 2 && d->dilated % 2 == 0 && d->dilated / 2 <= 128 && d->input_dim >= 1,
             "gen_run: the persistent kernel holds at most %d channels per vector (dilated <= 256)", lim);
  VQ_REQUIRE(d->sample_mode == VQVAE_GEN_SOFTMAX || d->sample_mode == VQVAE_GEN_MOL, "gen_run: sample_mode must be SOFTMAX or MOL");
  VQ_REQUIRE(d->uniforms && d->n_uniform >= 1 && d->cond && d->cond_follows_step, "gen_run: uniforms / per-step condition missing");
  if (d->sample_mode == VQVAE_GEN_MOL) VQ_REQUIRE(d->out_dim % 3 == 0 && d->n_uniform >= d->out_dim / 3, "gen_run: mixture sampling needs out_dim = 3*nr_mix and nr_mix uniforms");
  VQ_REQUIRE(d->out, "gen_run: output array missing");
  for (int l = 0; l < d->n_blocks; ++l) {
    const vqvae_gen_block& bk = d->blocks[l];
    VQ_REQUIRE(bk.conv_W && bk.conv_b && bk.cond_W && bk.cond_b && bk.res_W && bk.res_b && bk.skip_W && bk.skip_b && bk.dilation >= 1,
               "gen_run: block %d incomplete", l);
  }
  return 0;
}

extern "C" size_t vqvae_wavenet_gen_run_workspace_bytes(const vqvae_gen_desc* d) {
  if (mega_check(d) != 0) return 0;
  size_t o[11];
  return mega_layout(d, o, o + 1, o + 2, o + 3, o + 4, o + 5, o + 6, o + 7, o + 8, o + 9, o + 10, nullptr);
}

extern "C" int vqvae_wavenet_gen_run(const vqvae_gen_desc* d, int t0, int steps, void* ws, size_t ws_bytes, vqvae_stream_t s) {
  int rc = mega_check(d);
  if (rc) return rc;
  VQ_REQUIRE(t0 >= 0 && steps >= 0 && ws, "gen_run: bad step range / workspace");
  size_t off_tab, off_xq, off_z, off_s, off_s1, off_lg, off_idx, off_qt, off_crec, off_hrec, off_brec;
  long qoffs[MEGA_MAXL];
  const size_t need = mega_layout(d, &off_tab, &off_xq, &off_z, &off_s, &off_s1, &off_lg, &off_idx, &off_qt, &off_crec,
                                  &off_hrec, &off_brec, qoffs);
  if (ws_bytes < need) { set_error("gen_run: workspace %zu < %zu bytes", ws_bytes, need); return VQVAE_E_WORKSPACE; }
  hipStream_t st = (hipStream_t)s;
  char* base = (char*)ws;
  if (t0 == 0) {      // WaveNet.initialize (modules.py:232-244): zero queues, zero tags, block table
    VQ_CHECK_HIP(hipMemsetAsync(ws, 0, off_crec, st));
    MegaBlock tab[MEGA_MAXL];
    MegaQ qt[MEGA_MAXL];
    for (int l = 0; l < d->n_blocks; ++l) {
      const vqvae_gen_block& bk = d->blocks[l];
      tab[l] = MegaBlock{bk.conv_W, bk.conv_b, bk.cond_W, bk.cond_b, bk.res_W, bk.res_b, bk.skip_W, bk.skip_b, qoffs[l], bk.dilation, 0};
      qt[l] = MegaQ{qoffs[l], bk.dilation, 0};
    }
    VQ_CHECK_HIP(hipMemcpyAsync(base + off_tab, tab, sizeof(MegaBlock) * d->n_blocks, hipMemcpyHostToDevice, st));
    VQ_CHECK_HIP(hipMemcpyAsync(base + off_qt, qt, sizeof(MegaQ) * d->n_blocks, hipMemcpyHostToDevice, st));
    VQ_CHECK_HIP(hipStreamSynchronize(st));      // `tab`, `qt` are stack buffers
    hipLaunchKernelGGL(gen_mega_pack_kernel, dim3(d->n_blocks * MEGA_G), dim3(64), 0, st, (const MegaBlock*)(base + off_tab),
                       d->n_blocks, d->residual, d->dilated / 2, d->skip, d->cond_dim, (float4*)(base + off_crec),
                       (float4*)(base + off_hrec), (float4*)(base + off_brec));
    VQ_LAUNCH_CHECK();
  }
  if (!steps) return 0;
  MegaArgs a{};
  a.blk = (const MegaBlock*)(base + off_tab);
  a.qt = (const MegaQ*)(base + off_qt);
  a.crec = (const float4*)(base + off_crec); a.hrec = (const float4*)(base + off_hrec); a.brec = (const float4*)(base + off_brec);
  a.embed_W = d->embed_W; a.embed_b = d->embed_b; a.proj1_W = d->proj1_W; a.proj1_b = d->proj1_b; a.proj2_W = d->proj2_W; a.proj2_b = d->proj2_b;
  a.cond = d->cond; a.cond_bstride = d->cond_bstride; a.cond_cstride = d->cond_cstride;
  a.uniforms = d->uniforms; a.forced = d->forced_next; a.out = d->out; a.out_bstride = d->out_bstride; a.logits_out = d->logits_out;
  a.xq = (gran_t*)(base + off_xq); a.zbox = (gran_t*)(base + off_z); a.sbox = (gran_t*)(base + off_s);
  a.s1box = (gran_t*)(base + off_s1); a.lgbox = (gran_t*)(base + off_lg); a.idxbox = (gran_t*)(base + off_idx);
  a.status = (int*)base;
  a.L = d->n_blocks; a.n = d->n; a.input_dim = d->input_dim; a.R = d->residual; a.D = d->dilated; a.S = d->skip; a.Cc = d->cond_dim;
  a.O = d->out_dim; a.mode = d->sample_mode; a.n_uniform = d->n_uniform; a.t0 = t0; a.steps = steps; a.log_scale_min = d->log_scale_min;
  if (d->n == 1) hipLaunchKernelGGL((gen_mega_kernel<1, true>), dim3(MEGA_G), dim3(MEGA_THREADS), 0, st, a);
  else if (d->n == 2) hipLaunchKernelGGL((gen_mega_kernel<2, true>), dim3(MEGA_G), dim3(MEGA_THREADS), 0, st, a);
  else if (d->n == 4) hipLaunchKernelGGL((gen_mega_kernel<4, true>), dim3(MEGA_G), dim3(MEGA_THREADS), 0, st, a);
  else hipLaunchKernelGGL((gen_mega_kernel<4, false>), dim3(MEGA_G), dim3(MEGA_THREADS), 0, st, a);
  VQ_LAUNCH_CHECK();
  return 0;
}
