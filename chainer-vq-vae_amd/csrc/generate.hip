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
