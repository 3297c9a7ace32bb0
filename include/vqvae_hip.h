/*
 * vqvae_hip.h -- C ABI of libvqvae_hip.so: the MI355X (gfx950) drop-in for the
 * arithmetic of dhgrs/chainer-VQ-VAE's training hot path.
 *
 * The reference has no FFI of its own: it is pure Python and every multiply-add
 * runs inside Chainer 4.0.0b3 -> NumPy / CuPy+cuDNN.  The entry points below are
 * therefore what a Chainer-shaped `FunctionNode.forward/backward` for this path
 * binds (ctypes).  Each entry cites the reference call site
 * it replaces (file:line under the reference repo).
 *
 * Conventions
 *  - plain C, no C++/torch types; every pointer is a raw DEVICE pointer unless
 *    marked "host"; the library never allocates or frees caller tensors.
 *  - tensors are fp32, Chainer NCHW with W==1, i.e. (B, C, T[,1]) with T
 *    contiguous (net.py:12, modules.py:13-16, utils.py:85-110); labels int32.
 *  - every call is asynchronous on the given stream (hipStream_t as void*),
 *    re-entrant, and returns 0 on success, a positive hipError_t, or a negative
 *    VQVAE_E_* code; vqvae_last_error_string() describes the last failure of the
 *    calling thread.
 *  - scratch memory is passed in by the caller (ws, ws_bytes); the matching
 *    *_workspace_bytes() query gives the required size.
 */
#ifndef VQVAE_HIP_H_
#define VQVAE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQVAE_E_INVALID   (-1)   /* bad argument (shape/alignment/null)            */
#define VQVAE_E_WORKSPACE (-2)   /* workspace too small                            */
#define VQVAE_E_NOLIB     (-3)   /* librccl could not be loaded                    */
#define VQVAE_E_COMM      (-4)   /* RCCL returned an error                         */

typedef void* vqvae_stream_t;

const char* vqvae_last_error_string(void);
int vqvae_abi_version(void);      /* 5 since vqvae_conv1d_amax grew `packed`; 4: vqvae_resblock_amax grew x_max / res_scale / gh_scale (3: vqvae_resblock_desc grew `storage`; bindings must zero what they do not set) */

/* ---- device / memory / stream plumbing (replaces CuPy's allocator + streams,
 *      reached in the reference through model.to_gpu()/converter, updaters.py:8) */
int vqvae_device_count(int* n);
int vqvae_set_device(int dev);
int vqvae_device_info(char* name, int name_cap, int* n_cu, size_t* total_mem);
int vqvae_device_pci_bus_id(char* out, int cap);   /* of the current device, e.g. "0000:c1:00.0" (-> its NUMA node in sysfs) */
int vqvae_malloc(void** p, size_t bytes);
int vqvae_free(void* p);
int vqvae_memcpy_h2d(void* dst, const void* host_src, size_t bytes, vqvae_stream_t s);
int vqvae_memcpy_d2h(void* host_dst, const void* src, size_t bytes, vqvae_stream_t s);
/* page-locked host memory, and a host -> device copy from it that only enqueues (the caller orders its consumers with
 * an event): the converter's copy of updaters.py:8 on a copy stream, under the previous step's kernels            */
int vqvae_host_alloc(void** p, size_t bytes);
int vqvae_host_free(void* p);
int vqvae_memcpy_h2d_async(void* dst, const void* pinned_host_src, size_t bytes, vqvae_stream_t s);
int vqvae_memcpy_d2d(void* dst, const void* src, size_t bytes, vqvae_stream_t s);
/* `height` rows of `width` bytes, row pitches in bytes (a shifted window of every row of a 2-D array) */
int vqvae_memcpy2d_d2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width,
                       size_t height, vqvae_stream_t s);
int vqvae_memset(void* p, int byte_value, size_t bytes, vqvae_stream_t s);
int vqvae_stream_create(vqvae_stream_t* s);
int vqvae_stream_destroy(vqvae_stream_t s);
int vqvae_stream_synchronize(vqvae_stream_t s);
int vqvae_device_synchronize(void);
int vqvae_event_create(void** ev);
int vqvae_event_destroy(void* ev);
int vqvae_event_record(void* ev, vqvae_stream_t s);
int vqvae_event_synchronize(void* ev);
int vqvae_stream_wait_event(vqvae_stream_t s, void* ev);   /* s waits (on the GPU) for ev */
int vqvae_event_elapsed_ms(float* ms, void* ev_start, void* ev_stop);

/* ---- per-kernel timing with HIP events on the launch stream (bench.py roofline).
 *      tag = one of VQVAE_PROF_*; enable(bitmask: bit t = tag t, -1 = all), run, then
 *      read (synchronises the device). */
#define VQVAE_PROF_NONE            0
#define VQVAE_PROF_RESBLOCK_GATE   1   /* dilated conv + cond proj + gate (fwd)       */
#define VQVAE_PROF_RESBLOCK_OUT    2   /* res 1x1 + residual add (fwd)                */
#define VQVAE_PROF_RESBLOCK_BWD_GZ 3   /* bwd: gz + gate derivative                   */
#define VQVAE_PROF_RESBLOCK_BWD_GX 4   /* bwd-data of the dilated conv                */
#define VQVAE_PROF_RESBLOCK_BWD_GC 5   /* bwd-data of the condition projection        */
#define VQVAE_PROF_RESBLOCK_WGRAD  6   /* bwd-weight kernels of the block             */
#define VQVAE_PROF_CONV_FWD        7
#define VQVAE_PROF_CONV_BWD_DATA   8
#define VQVAE_PROF_CONV_WGRAD      9
#define VQVAE_PROF_VQ_NEAREST     10
#define VQVAE_PROF_RESSTACK_SKIP   11  /* the skip sum of a whole ResidualNet as ONE GEMM (fwd)       */
#define VQVAE_PROF_WGRAD_DIL       12  /* bwd-weight of the dilated convs (resstack_dil_wgrad: several blocks per launch, incl. the fixed-order reduce) */
#define VQVAE_PROF_WGRAD_RES_SKIP  13  /* bwd-weight of the res / skip 1x1 convs (resstack_{res,skip}_wgrad, incl. the reduce) */
#define VQVAE_PROF_NTAGS          16
int vqvae_prof_enable(int tag_mask);
int vqvae_prof_reset(void);
int vqvae_prof_read(int tag, double* total_ms, int* launches);

/* ---- arithmetic of every MFMA contraction (convs fwd / bwd-data / bwd-weight, ResidualBlock /
 *      ResidualNet entry points).  Tensors in HBM, accumulators, biases, gates, losses, optimizer and
 *      the vector quantiser are fp32 in every mode.
 *      2 (default): fp32 products on the bf16 matrix pipe -- both operands split EXACTLY into three
 *        bf16 (h + m + l, each the RNE of the remainder), six of the nine products on
 *        v_mfma_f32_32x32x16_bf16 with fp32 accumulate; the dropped three are < 2^-25 of the product.
 *        As accurate as mode 0 (checked against float64), 0.375 of its matrix-pipe time.
 *      0: fp32 operands on v_mfma_f32_32x32x2_f32.
 *      1: operands rounded to bf16 (round-to-nearest-even), v_mfma_f32_32x32x16_bf16, fp32
 *        accumulation -- BASELINE configs[4].
 *      3: fp32 products on the fp16 matrix pipe, THREE v_mfma_f32_32x32x16_f16 per product (`float32x2`): both
 *        operands scaled by a power of two per tensor (taken from the tensor's absolute maximum) and split into
 *        hi + lo fp16; hi*hi + hi*lo + lo*hi, fp32 accumulate.  hi + lo carries 22 significand bits + a sign --
 *        one fp32 rounding per operand -- and a 16-deep K step is rounded once instead of eight times, so against
 *        float64 it is at or below the error of modes 0 and 2 (tests/test_gpu_kernels.py).  The kernels need each
 *        operand's absolute maximum BEFORE they run: ResidualNet's packed chain hands them from launch to launch
 *        (vqvae_resblock_amax; producers publish max |y| from their epilogues), the generic conv entry points
 *        find them with one extra pass over the operand when the launch is large (>= 8 GFLOP) and run mode 2's
 *        kernels otherwise.  Half of mode 2's matrix-pipe time.
 *      Workspace sizes depend on the mode (packed weight slabs are 1.5x larger in modes 2 and 3): query
 *      them after setting it.                                                                    */
int vqvae_set_matmul_dtype(int dtype);
/* An absolute maximum is VQVAE_AMAX_SLOTS device words: the bit patterns of non-negative floats (unsigned order ==
 * float order) whose maximum is max |x[i]| -- producers raise one word per workgroup with atomicMax (spreading the
 * same-address atomics), consumers take the maximum of all.  The form in which matmul mode 3 passes a tensor's
 * scale from its producer to its consumers.  vqvae_absmax zeroes the words, then scans x.       */
#define VQVAE_AMAX_SLOTS 16
int vqvae_absmax(const float* x, size_t n, uint32_t* amax, vqvae_stream_t s);
/* matmul mode 3: smallest generic conv launch (GFLOP = 2 B Tout Cout Cin K / 1e9) that pays for the extra pass over
 * its operand and runs the float32x2 kernels; default 8, 0 = every launch (tests).                */
int vqvae_set_f32x2_min_gflop(double gflop);
int vqvae_get_matmul_dtype(void);
/* weight-gradient kernel choice: 0 = automatic (the 16-byte-LDS fp32 kernel where it applies),
 * 1 = always the generic kernel (A/B checks) */
int vqvae_set_wgrad_impl(int impl);

/* ---- generic 1-D convolution == chainer L.Convolution2D / L.DilatedConvolution2D
 *      with ksize=(K,1), stride=(s,1), pad=(p,0), dilate=(d,1)
 *      (net.py:12-17 encoder, net.py:34-43 condition embed, modules.py:17-22,
 *      127-141 1x1 / embed / proj convs).  W is Chainer's (Cout,Cin,K,1), b (Cout).
 *      Tout may be smaller than the natural output length: the tail is cropped
 *      (modules.py:41 `h[:, :, :length]`, modules.py:152).                       */
typedef struct {
  int B, Cin, Tin, Cout, Tout, K, stride, pad, dil;
  int relu;                 /* fuse F.relu on the output (net.py:20-24, 49-53)   */
} vqvae_conv1d_desc;

size_t vqvae_conv1d_workspace_bytes(const vqvae_conv1d_desc* d);
int vqvae_conv1d_fwd(const vqvae_conv1d_desc* d, const float* x, const float* W,
                     const float* b, float* y, void* ws, size_t ws_bytes, vqvae_stream_t s);
/* gx (B,Cin,Tin) (+)= conv^T(gy) */
int vqvae_conv1d_bwd_data(const vqvae_conv1d_desc* d, const float* W, const float* gy,
                          float* gx, int accumulate, void* ws, size_t ws_bytes,
                          vqvae_stream_t s);
/* gW (Cout,Cin,K) and gb (Cout) (+)= ...; gb may be NULL */
int vqvae_conv1d_bwd_weight(const vqvae_conv1d_desc* d, const float* x, const float* gy,
                            float* gW, float* gb, int accumulate, void* ws,
                            size_t ws_bytes, vqvae_stream_t s);
/* The same three with the operands' absolute maxima handed in and the result's handed out (matmul mode 3; groups of
 * VQVAE_AMAX_SLOTS device words as vqvae_absmax writes them): a maximum that travels with its tensor saves the scan
 * vqvae_conv1d_* would otherwise run over a large operand.  NULL members: scan (inputs) / not wanted (out; zeroed by
 * the caller, raised with atomicMax by the launch's epilogue, any matmul mode).  vqvae_conv1d_uses_f32x2(d) says
 * whether the launch is large enough to run the three-product kernels, i.e. whether handing maxima in matters. */
typedef struct {
  const uint32_t* x;     /* fwd, bwd_weight: max |x|   */
  const uint32_t* gy;    /* bwd_data, bwd_weight: max |gy| */
  uint32_t* out;         /* fwd: max |y|; bwd_data: max |gx| */
  const void* packed;    /* fwd, bwd_data (any matmul mode; NULL: the launch re-lays W into its workspace itself): W already
                          * re-laid by vqvae_conv1d_pack for THIS desc and direction under the current matmul mode and
                          * f32x2 threshold -- see below                                                              */
} vqvae_conv1d_amax;
/* Weights packed ahead of the launches that read them.  vqvae_conv1d_fwd / _bwd_data re-lay W into their workspace in front
 * of every GEMM (one or two small launches); the weights only change in the optimizer, so a caller may pack all of a
 * step's slabs at once -- on another stream, as soon as the optimizer is done -- and hand each launch its slab through
 * vqvae_conv1d_amax::packed (ordering between the streams is the caller's: an event).  packed[i]: device buffer of
 * vqvae_conv1d_packed_bytes(&descs[i], backward[i]) bytes; backward[i] = 0: the slab vqvae_conv1d_fwd* reads, 1: the one
 * vqvae_conv1d_bwd_data* reads.  Jobs share launches (<= 24 per launch).                                             */
size_t vqvae_conv1d_packed_bytes(const vqvae_conv1d_desc* d, int backward);
int vqvae_conv1d_pack(int n, const vqvae_conv1d_desc* descs, const float* const* W, const int* backward,
                      void* const* packed, vqvae_stream_t s);
int vqvae_conv1d_uses_f32x2(const vqvae_conv1d_desc* d);
int vqvae_conv1d_fwd_amax(const vqvae_conv1d_desc* d, const float* x, const float* W,
                          const float* b, float* y, void* ws, size_t ws_bytes,
                          const vqvae_conv1d_amax* amax, vqvae_stream_t s);
int vqvae_conv1d_bwd_data_amax(const vqvae_conv1d_desc* d, const float* W, const float* gy,
                               float* gx, int accumulate, void* ws, size_t ws_bytes,
                               const vqvae_conv1d_amax* amax, vqvae_stream_t s);
/* gx = x_relu > 0 ? conv^T(gy) : 0 -- backward-data of a conv whose INPUT x_relu (B, Cin, Tin) is the output of a ReLU,
 * with that ReLU's backward applied where the gradient is produced (amax may be NULL)                               */
int vqvae_conv1d_bwd_data_relu(const vqvae_conv1d_desc* d, const float* W, const float* gy,
                               const float* x_relu, float* gx, void* ws, size_t ws_bytes,
                               const vqvae_conv1d_amax* amax, vqvae_stream_t s);
int vqvae_conv1d_bwd_weight_amax(const vqvae_conv1d_desc* d, const float* x, const float* gy,
                                 float* gW, float* gb, int accumulate, void* ws, size_t ws_bytes,
                                 const vqvae_conv1d_amax* amax, vqvae_stream_t s);
/* Device-side conditional forms: the whole launch is a no-op when skip_flag != NULL and
 * *skip_flag != 0 at execution time (skip_flag is a device pointer).  Used to pick between two
 * implementations of one operator by a flag computed on the device, without a host round trip
 * (vqvae_embed_onehot_*).                                                                        */
int vqvae_conv1d_fwd_cond(const vqvae_conv1d_desc* d, const float* x, const float* W,
                          const float* b, float* y, void* ws, size_t ws_bytes,
                          const int32_t* skip_flag, vqvae_stream_t s);
int vqvae_conv1d_bwd_weight_cond(const vqvae_conv1d_desc* d, const float* x, const float* gy,
                                 float* gW, float* gb, int accumulate, void* ws, size_t ws_bytes,
                                 const int32_t* skip_flag, vqvae_stream_t s);

/* ---- WaveNet ResidualBlock (WaveNet/modules.py:30-56), dropout_zero_rate == 0:
 *      h = dilconv(x)[:, :, :T] + condition_proj(c); z = tanh(h_a)*sigmoid(h_b);
 *      residual = res(z) + x; skip = skip(z).                                    */
typedef struct {
  int B, T;
  int Cr;      /* residual_channels                                              */
  int Cd;      /* dilated_channels (gate halves Cd/2; Cd/2 % 32 == 0)            */
  int Cs;      /* skip_channels                                                  */
  int Cc;      /* condition_dim                                                  */
  int K;       /* filter_size                                                    */
  int dil;     /* dilation                                                       */
  int storage; /* bit set of VQVAE_STORE_*: tensors of the chain kept in HBM as bf16
                  (0 = all fp32; matmul mode 1 only -- ask vqvae_resblock_bf16_storage) */
} vqvae_resblock_desc;

/* BASELINE configs[4] ("bf16"): tensors the bf16 mode may keep in HBM as bf16 -- same element strides,
 * 2-byte elements, the caller's fp32-sized buffer simply half used.  The caller opts in per tensor through
 * vqvae_resblock_desc::storage and must pass the same bits to every entry point that produces or reads the
 * tensor (resblock_bwd_packed produces gh; resstack_dil_wgrad / resblock_wgrad / upsample_linear_bwd_bf16
 * read it).
 *   GH: gh = [ga; gb], the gate pre-activation gradient (B, Cd, T).  The GEMMs that read it round it to
 *       bf16 anyway; its bias sums and the latent pull-back then see the rounded values.              */
#define VQVAE_STORE_GH_BF16 1
/*   X / RES: the residual stream.  RES: this block's residual output x_{l+1} = bf16((Wr z + br) + x_l) is stored as
 *       bf16; X: this block's input x_l is (i.e. it is the RES output of the block before it).  The first block of a
 *       stack reads the caller's fp32 x (RES only), the last one has no residual output (X only).  The rounding is
 *       part of the result (the oracle's bf16 mode mirrors it); readers: the gate GEMM, the residual add, the dilated
 *       conv's weight gradient (resstack_dil_wgrad: every block of ONE launch shares the flag).                    */
#define VQVAE_STORE_X_BF16 2
#define VQVAE_STORE_RES_BF16 4
/*   GX / GRES: the residual GRADIENT stream, the backward's counterpart.  GX: this block's gx = bf16(g_res + Wd^T gh)
 *       is stored as bf16; GRES: this block's g_res is (it is the GX output of the block above).  The block nearest the
 *       output has no g_res (GX only), the first block's gx leaves the stack as fp32 (GRES only).  Readers: the
 *       gate-derivative GEMM, the backward-data epilogue's add, resstack_res_wgrad (one flag per launch).          */
#define VQVAE_STORE_GX_BF16 8
#define VQVAE_STORE_GRES_BF16 16
/* the bits the library supports for this block shape in the current matmul mode (0 outside mode 1)    */
int vqvae_resblock_bf16_storage(const vqvae_resblock_desc* d);

/* Matmul mode 3 (float32x2): tensors of the packed chain kept PRE-SPLIT.  Every float32x2 GEMM splits each fp32 operand
 * element into fp16 hi + fp16 lo of x * 2^k while staging it; a tensor that three GEMMs read (x_l: gate GEMM + dilated
 * weight gradient; gh_l: backward-data GEMM + dilated weight gradient + pull-back) is stored split by its producer
 * instead: ONE dword per element at the fp32 element's address = hi | lo << 16.  k comes from a rigorous bound on the
 * tensor's absolute maximum that is known before the producer runs (weight norms found by vqvae_resstack_pack x the
 * ACTUAL maxima of the producer's inputs); the producer writes that bound into the tensor's SCALE words (a zeroed
 * VQVAE_AMAX_SLOTS group, vqvae_resblock_amax::res_scale / gh_scale) and every reader is handed those words where it
 * used to get the maximum.  The values are fp32-accurate (an element within 2^-10 of the bound keeps 24 significant bits,
 * smaller ones an absolute 2^-39 of the bound: DESIGN.md 3a); they are opaque to the caller.
 *   GH : gh (B, Cd, T), written by resblock_bwd_packed; readers resstack_dil_wgrad, upsample_linear_bwd_f16x2.
 *   X / RES: the residual stream, as VQVAE_STORE_X_BF16 / RES_BF16 (first block: RES only; last block: X only).   */
#define VQVAE_STORE_GH_F16X2 32
#define VQVAE_STORE_X_F16X2 64
#define VQVAE_STORE_RES_F16X2 128
/*   GATES_SIG: of the three tensors the gate kernel saves per block -- tanh, sigmoid (`gates`) and z = tanh * sigmoid -- the
 *   backward needs two: resblock_fwd_packed leaves the tanh half of `gates` UNWRITTEN (a third of the gate launch's 189 MB
 *   of stores: 106 -> 88 us per launch) and resblock_bwd_packed takes tanh = z / sigmoid from the z it is handed (z is
 *   fl(tanh * sigmoid): tanh to 2^-23 relative; where sigmoid underflows to 0, z is 0 and both derivatives vanish).
 *   Same flag in the forward and the backward call of a block.  `gates` keeps its (B, Cd, T) shape.                  */
#define VQVAE_STORE_GATES_SIG 256
/* the pre-split bits the library supports for this block shape in the current matmul mode (0 outside mode 3)      */
int vqvae_resblock_f16x2_storage(const vqvae_resblock_desc* d);
/* which tensors vqvae_resblock_f16x2_storage may offer: bit 0 = gh, bit 1 = the residual stream, bit 2 = GATES_SIG
 * (default 7; 0 = every tensor of the chain stays fp32 -- the A/B switch of the pre-split tests) */
int vqvae_set_presplit(int mask);

typedef struct {            /* parameters, Chainer layouts (modules.py:13-22)     */
  const float *Wd, *bd;     /* conv            (Cd, Cr, K, 1), (Cd)               */
  const float *Wc, *bc;     /* condition_proj  (Cd, Cc, 1, 1), (Cd)               */
  const float *Wr, *br;     /* res             (Cr, Cd/2, 1, 1), (Cr)             */
  const float *Ws, *bs;     /* skip            (Cs, Cd/2, 1, 1), (Cs)             */
} vqvae_resblock_params;

typedef struct {            /* gradients, same layouts; any pair may be NULL      */
  float *gWd, *gbd, *gWc, *gbc, *gWr, *gbr, *gWs, *gbs;
} vqvae_resblock_grads;

/* Optional: the condition projection computed ONCE at the latent rate.  Up-sampling
 * (net.py:54-55) is linear and acts per channel, so condition_proj(upsample(c)) ==
 * upsample(condition_proj(c)): ResidualNet projects the (B,Cc,Tl) latent-rate
 * condition through all blocks' 1x1 convs in one small GEMM and each block adds the
 * align-corners lerp of its (B,Cd,Tl) slice P in the gate epilogue (v0/w0/w1 are the
 * resize tables for Tl -> T, v1 == v0+1).  Removes Cc of K*Cr+Cc contraction steps. */
typedef struct {
  const float* P;        /* (B, Cd, Tl) incl. the bias bc; batch stride P_bstride      */
  long P_bstride;
  int Tl;
  const int32_t* v0; const float* w0; const float* w1;
  /* ABI 4.  P_has_bd != 0: P also includes the dilated conv's bias bd (the caller added bd to the projection's bias: the
   * lerp weights sum to one), so the gate kernel adds no bias at all.  P_amax: max |P| as VQVAE_AMAX_SLOTS words (as
   * vqvae_conv1d_fwd_amax publishes them; NULL = unknown).  With both, and T >= 26 Tl, the two-tap gate GEMM adds the
   * lerp as ONE more step of its contraction -- an (M x 8) x (8 x 128) product of P's columns under the tile with the
   * lerp coefficients -- instead of fetching four values of P per output element in its epilogue.                  */
  int P_has_bd; const uint32_t* P_amax;
} vqvae_resblock_cproj;

size_t vqvae_resblock_workspace_bytes(const vqvae_resblock_desc* d);
/* res may be NULL (last block: residual unused, modules.py:89-96).  skip is
 * overwritten, or accumulated into when skip_accumulate != 0 (modules.py:92-95);
 * skip may be NULL too (ResidualNet computes the skip sum with vqvae_resstack_skip_fwd).
 * gates (B,Cd,T) = [tanh(h_a) | sigmoid(h_b)] and z (B,Cd/2,T) are saved for bwd.  z is OPAQUE to
 * the caller: it is only ever handed back to this library (skip sum, weight gradients), and in matmul
 * mode 1 (bf16 operands) the configs-sized blocks keep it as bf16 inside the same buffer.
 * With cproj != NULL, cond / Wc / bc are ignored (may be NULL).                       */
int vqvae_resblock_fwd(const vqvae_resblock_desc* d, const vqvae_resblock_params* p,
                       const float* x, const float* cond, const vqvae_resblock_cproj* cproj,
                       float* res, float* skip,
                       int skip_accumulate, float* gates, float* z, void* ws,
                       size_t ws_bytes, vqvae_stream_t s);
/* g_res may be NULL; gx may be NULL; cond may be NULL (cproj mode: then gcond, gWc, gbc
 * must be NULL); gcond (B,Cc,T) may be NULL and is accumulated
 * into when gcond_accumulate != 0; gh_out (B,Cd,T), if not NULL, receives the
 * gradient w.r.t. the pre-gate activation h (kept by ResidualNet for the stack-level
 * condition gradient); parameter grads are accumulated when grads_accumulate. */
int vqvae_resblock_bwd(const vqvae_resblock_desc* d, const vqvae_resblock_params* p,
                       const float* x, const float* cond, const float* gates,
                       const float* z, const float* g_res, const float* g_skip,
                       float* gx, float* gcond, int gcond_accumulate, float* gh_out,
                       const vqvae_resblock_grads* g, int grads_accumulate, void* ws,
                       size_t ws_bytes, vqvae_stream_t s);

/* ---- pack once per step (ResidualNet's chain, WaveNet/modules.py:89-96).
 *      The GEMM kernels read weights as re-laid "slabs"; vqvae_resblock_fwd / _bwd re-lay them into the
 *      workspace on every call.  The weights only change in the optimizer, so ResidualNet packs the
 *      slabs of ALL its blocks once per training step (forward + backward forms: gated dilated conv, res
 *      1x1, gz from g_res / g_skip, dilated-conv backward-data) and hands every block its slice:
 *        packed = nblocks x vqvae_resstack_packed_bytes(d) bytes of device memory;
 *        params = HOST array of nblocks vqvae_resblock_params (device pointers), has_res[l] = 0 for a
 *        block whose residual output is unused (the last one);
 *        block l's slice starts at packed + l * vqvae_resstack_packed_bytes(d).
 *      The _packed entry points are the latent-rate-condition chain only (cproj given, no per-block
 *      skip output, no per-block condition / parameter gradients: ResidualNet batches those).   */
/* matmul mode 3 only (NULL / ignored otherwise): where the absolute maxima of the chain's tensors live -- device
 * groups of VQVAE_AMAX_SLOTS uint32 as written by vqvae_absmax.  Inputs must be final when the launch runs; outputs
 * are raised with atomicMax by the producing kernel's epilogue and must have been zeroed by the caller.      */
typedef struct {
  const uint32_t* x;        /* fwd in : max |x|      (the block's input)                               */
  uint32_t* res;            /* fwd out: max |res|    (the next block's x); NULL with res == NULL       */
  const uint32_t* g_res;    /* bwd in : max |g_res|  ; NULL with g_res == NULL                         */
  const uint32_t* g_skip;   /* bwd in : max |g_skip|                                                  */
  uint32_t* gh;             /* bwd out: max |gh_out| (also read by the backward-data launch)           */
  uint32_t* gx;             /* bwd out: max |gx|     (the previous block's g_res); may be NULL         */
  /* pre-split storage (VQVAE_STORE_*_F16X2; NULL / ignored otherwise).  With VQVAE_STORE_X_F16X2 `x` above holds the
   * SCALE words of x (the res_scale of the block before) and x_max its actual maximum (that block's `res`); with an
   * fp32 x both are the same group.  With VQVAE_STORE_GH_F16X2 `gh` still receives the actual maximum.              */
  const uint32_t* x_max;    /* fwd in : actual max |x| (enters the bound of the residual output)              */
  uint32_t* res_scale;      /* fwd out: scale words of the pre-split residual output (zeroed by the caller)    */
  uint32_t* gh_scale;       /* bwd out: scale words of the pre-split gh_out (zeroed by the caller)             */
  /* the latent pull-back of gh (the adjoint of the forward's condition lerp) inside resblock_bwd_packed: with pb_part !=
   * NULL (Cd == 256, T % 128 == 0, T >= 64 pb_Tl) the gate-derivative launch leaves, per 128-column tile, the sums of
   * gh over the tile's columns for the four latent positions under it in pb_part[b][tile][Cd][4] -- the caller does not
   * call vqvae_upsample_linear_bwd* for this block and finishes all blocks at once with vqvae_pullback_reduce.
   * pb_v0 / pb_w0 / pb_w1: the Tl -> T resize tables (as vqvae_resblock_cproj).                                        */
  float* pb_part; const int32_t* pb_v0; const float* pb_w0; const float* pb_w1; int pb_Tl;
} vqvae_resblock_amax;
/* gP[b][l * Cd + c][v] = sum over the tiles n whose four positions v0[128 n] .. + 3 hold v of part[l][b][n][c][v - v0[128 n]]
 * (ascending n: deterministic); part = nblocks consecutive pb_part buffers of B * (T / 128) * Cd * 4 floats          */
int vqvae_pullback_reduce(const float* part, const int32_t* v0, int nblocks, int B, int T, int Cd, int Tl,
                          float* gP, vqvae_stream_t s);
/* the same for a GROUP of blocks inside a larger gP: gP points at the group's first row (l = 0 of `part`), gP_bstride =
 * elements between batch items of the whole tensor (rows of all blocks * Tl).  ResidualNet's backward finishes the blocks
 * it has passed five at a time, between the chain's big launches, instead of all twenty in the latency-bound tail. */
int vqvae_pullback_reduce_into(const float* part, const int32_t* v0, int nblocks, int B, int T, int Cd, int Tl,
                               float* gP, size_t gP_bstride, vqvae_stream_t s);
/* The run-time guard of float32x2's pre-split storage (VQVAE_STORE_*_F16X2): pair i = (scale[i], amax[i]) names one
 * pre-split tensor by its SCALE words (the a-priori bound it was split under, word 0) and the words its producer raised
 * to its ACTUAL maximum.  A bound 2^m above the maximum costs m bits of the mode's absolute floor; the check counts, on
 * the device and without a host round trip, every tensor with bound / max > 2^log2_limit (or a bound below the maximum):
 *   report[0] += violations, report[1] = max(report[1], float bits of the largest bound / max seen), report[2] += tensors
 * checked (report: 3 device words the caller zeroes when it starts counting).  NULL pairs, groups nobody wrote and
 * all-zero tensors are skipped.  There is no counterpart in the reference (Chainer computes in fp32, modules.py:40-55). */
int vqvae_f32x2_contract_check(int n, const uint32_t* const* scale, const uint32_t* const* amax, int log2_limit,
                               uint32_t* report, vqvae_stream_t s);

/* ---- a stack of latent-rate convolutions in one launch per direction (csrc/latent.hip) -------------------------------
 * L "same"-padded dilated 3-tap convs + ReLU over (B, C, T) tensors, layer l: h_l = relu(conv(h_{l-1}; W_l (C, C, 3), pad =
 * dil_l, dilate = dil_l) + b_l), h_0 = x -- ConditionEmbed's five local convs (net.py:34-53).  One workgroup per sample
 * keeps the sample's state in LDS; every h_l is written once for the backward.  Served: L <= 8, C in {32, 64}, T <= 128,
 * dilations <= 16 (vqvae_convstack_supported); fp32 MFMA arithmetic in every matmul mode.
 *   fwd:  h[l] (B, C, T) receives h_{l+1}, l = 0 .. L-1
 *   bwd:  gy = gradient of h_L; gx (nullable) = gradient of x; gW[l] / gb[l] (nullable) receive (accumulate != 0: are added)
 *         the parameter gradients, summed over the samples in ascending order (deterministic); ws: convstack_workspace_bytes */
int vqvae_convstack_supported(int L, int C, int T, const int* dil);
size_t vqvae_convstack_workspace_bytes(int L, int B, int C);
int vqvae_convstack_fwd(int L, int B, int C, int T, const int* dil, const float* x, const float* const* W,
                        const float* const* b, float* const* h, vqvae_stream_t s);
int vqvae_convstack_bwd(int L, int B, int C, int T, const int* dil, const float* x, const float* const* W,
                        const float* const* h, const float* gy, float* gx, float* const* gW, float* const* gb,
                        int accumulate, void* ws, size_t ws_bytes, vqvae_stream_t s);
/* The whole backward of ONE encoder stage -- Convolution2D(C, C, (4, 1), stride 2, pad 1), net.py:12-17, C = 64 -- in one
 * launch + a reduce instead of four (backward-data GEMM, phase split, weight-gradient GEMM, reduce): gx (nullable; with
 * mask_by_x != 0 multiplied by x > 0, the backward of the ReLU that produced x), gW (C, C, 4) and gb (C,) (nullable;
 * accumulate != 0: added to).  Workgroups own 128 input columns each; their weight-gradient shares are summed in ascending
 * order (deterministic).  fp32 MFMA arithmetic in every matmul mode.  ws: vqvae_conv_s2_bwd_workspace_bytes. */
int vqvae_conv_s2_bwd_supported(int Cin, int Cout, int K, int stride, int pad, int dil, int Tin, int Tout);
size_t vqvae_conv_s2_bwd_workspace_bytes(int B, int C, int Tin);
int vqvae_conv_s2_bwd(int B, int C, int Tin, int Tout, const float* x, const float* W, const float* gy, int mask_by_x,
                      float* gx, float* gW, float* gb, int accumulate, void* ws, size_t ws_bytes, vqvae_stream_t s);
size_t vqvae_resstack_packed_bytes(const vqvae_resblock_desc* d);
int vqvae_resstack_pack(const vqvae_resblock_desc* d, int nblocks,
                        const vqvae_resblock_params* params, const int* has_res, void* packed,
                        size_t packed_bytes, vqvae_stream_t s);
int vqvae_resblock_fwd_packed(const vqvae_resblock_desc* d, const vqvae_resblock_params* p,
                              const float* x, const vqvae_resblock_cproj* cproj, float* res,
                              float* gates, float* z, void* ws, size_t ws_bytes,
                              const void* packed, const vqvae_resblock_amax* amax, vqvae_stream_t s);
/* gz = Wr^T g_res + Ws^T g_skip, gate derivative -> gh_out (B,Cd,T), gx = g_res + conv^T(gh_out);
 * g_res / gx may be NULL as in vqvae_resblock_bwd.                                            */
int vqvae_resblock_bwd_packed(const vqvae_resblock_desc* d, const vqvae_resblock_params* p,
                              const float* x, const float* gates, const float* z,
                              const float* g_res, const float* g_skip, float* gx, float* gh_out,
                              void* ws, size_t ws_bytes, const void* packed,
                              const vqvae_resblock_amax* amax, vqvae_stream_t s);

/* Weight gradients of the dilated conv only: gWd (+)= gh x_taps^T, gbd (+)= rowsum(gh), from
 * the gh that vqvae_resblock_bwd wrote to gh_out.  Split out so ResidualNet can run it on a
 * second stream, concurrently with the next block's backward-data chain.              */
int vqvae_resblock_wgrad(const vqvae_resblock_desc* d, const float* x, const float* gh,
                         float* gWd, float* gbd, int accumulate, void* ws, size_t ws_bytes,
                         vqvae_stream_t s);

/* ---- ResidualNet-level contractions (WaveNet/modules.py:89-96): the arguments are
 *      HOST arrays of nblocks (<= 24) device pointers.
 *      skip_fwd : skip (B,Cs,T) (+)= sum_l (Ws_l z_l + bs_l) -- one GEMM, K = nblocks*Cd/2
 *                 (deeper stacks are fed in groups of <= 24 blocks with accumulate)
 *      gcond_bwd: gcond (B,Cc,T) (+)= sum_l Wc_l^T gh_l     -- one GEMM, K = nblocks*Cd
 *      skip_wgrad: gWs_l (+)= g_skip z_l^T, gbs_l (+)= rowsum(g_skip) for every l     */
size_t vqvae_resstack_workspace_bytes(const vqvae_resblock_desc* d, int nblocks);
/*      skip_amax_out (nullable): max |skip| is raised there (VQVAE_AMAX_SLOTS zeroed words)         */
int vqvae_resstack_skip_fwd(const vqvae_resblock_desc* d, int nblocks,
                            const float* const* Ws, const float* const* bs,
                            const float* const* z, float* skip, int accumulate, int relu, void* ws,
                            size_t ws_bytes, uint32_t* skip_amax_out, vqvae_stream_t s);
/* the same in two halves: _prepare re-lays the skip weights and sums the biases into ws (parameters only: a caller may run
 * it ahead, on another stream, as soon as the optimizer is done), _fwd_prepared runs the GEMM over a ws so prepared for the
 * same desc, nblocks and matmul mode                                                                               */
int vqvae_resstack_skip_prepare(const vqvae_resblock_desc* d, int nblocks, const float* const* Ws,
                                const float* const* bs, void* ws, size_t ws_bytes, vqvae_stream_t s);
int vqvae_resstack_skip_fwd_prepared(const vqvae_resblock_desc* d, int nblocks, const float* const* z, float* skip,
                                     int accumulate, int relu, const void* ws, size_t ws_bytes,
                                     uint32_t* skip_amax_out, vqvae_stream_t s);
/*      relu != 0: skip = max(skip (+ old skip), 0) -- WaveNet's F.relu(resnet(...)) (modules.py:158) in this epilogue; with
 *      more than one group of blocks pass it with the LAST group only                                                   */
int vqvae_resstack_gcond_bwd(const vqvae_resblock_desc* d, int nblocks,
                             const float* const* Wc, const float* const* gh, float* gcond,
                             int accumulate, void* ws, size_t ws_bytes, vqvae_stream_t s);
/*      The trailing *_amax arguments of the three weight-gradient entries: matmul mode 3's operand maxima
 *      (device uint32 slots / HOST arrays of such pointers); NULL keeps mode 2's kernels.      */
int vqvae_resstack_skip_wgrad(const vqvae_resblock_desc* d, int nblocks, const float* g_skip,
                              const float* const* z, float* const* gWs, float* const* gbs,
                              int accumulate, void* ws, size_t ws_bytes, const uint32_t* g_skip_amax,
                              vqvae_stream_t s);
/*      res_wgrad : gWr_l (+)= g_res_l z_l^T, gbr_l (+)= rowsum(g_res_l) for every l with
 *      g_res[l] != NULL (the last block has none) -- one launch                      */
int vqvae_resstack_res_wgrad(const vqvae_resblock_desc* d, int nblocks,
                             const float* const* g_res, const float* const* z,
                             float* const* gWr, float* const* gbr, int accumulate, void* ws,
                             size_t ws_bytes, const uint32_t* const* g_res_amax, vqvae_stream_t s);
/* dilated-conv weight / bias gradients of nblocks blocks (nblocks * K <= 24) in one launch: block l
 * contributes K segments (x[l] shifted by -(K-1-j)*dils[l], output gradient gh[l]); gWd[l] (Cd,Cr,K),
 * gbd[l] (Cd) -- NULL entries are skipped.  Workspace from the _workspace_bytes query.            */
size_t vqvae_resstack_dil_wgrad_workspace_bytes(const vqvae_resblock_desc* d, int nblocks);
int vqvae_resstack_dil_wgrad(const vqvae_resblock_desc* d, int nblocks, const int* dils,
                             const float* const* x, const float* const* gh, float* const* gWd,
                             float* const* gbd, int accumulate, void* ws, size_t ws_bytes,
                             const uint32_t* const* x_amax, const uint32_t* const* gh_amax,
                             vqvae_stream_t s);

/* ---- vector quantiser: StraightThrough.forward / backward (utils.py:176-231).
 *      z (B,d,T) [T contiguous], W (k,d).  idx (B,T) int32 is bit-exact with
 *      numpy.argmin(numpy.sum((xs-W)**2, axis=2), axis=1): first minimum wins,
 *      distances evaluated in the reference's fp32 operation order
 *      (sequential over d, sub/mul/add individually rounded).
 *      mode 0: MFMA pairwise distances + wavefront argmin + exact re-check of every
 *      row whose runner-up is inside the rounding band; mode 1: exact evaluation
 *      of all k codes for every row (test cross-check).  e (B,d,T) = W[idx] may be
 *      NULL.  n_rechecked (device int32, may be NULL) receives the number of rows
 *      that took the exact re-check path.                                        */
size_t vqvae_vq_workspace_bytes(int B, int d, int T, int k);
int vqvae_vq_nearest_fwd(const float* z, const float* W, int B, int d, int T, int k,
                         int mode, int32_t* idx, float* e, int32_t* n_rechecked,
                         void* ws, size_t ws_bytes, vqvae_stream_t s);
/* gW (k,d) (+)= onehot(idx)^T . gy, accumulated in float64 then rounded
 * (utils.py:227-228).  gy (B,d,T).                                               */
int vqvae_vq_grad_w(const int32_t* idx, const float* gy, int B, int d, int T, int k,
                    float* gW, int accumulate, void* ws, size_t ws_bytes,
                    vqvae_stream_t s);

/* ---- F.resize_images along T, align-corners (net.py:54-55): tables computed by
 *      the host in float64 exactly as Chainer does (v0/v1 int32[Tout], w0/w1 fp32[Tout]);
 *      y[b,c,i] = w0[i]*x[b,c,v0[i]] + w1[i]*x[b,c,v1[i]].  y has batch stride
 *      y_bstride elements (writes a channel slice of the concat, net.py:63).
 *      bwd uses host tables of contributing output ranges per input position
 *      (lo0/hi0: outputs with v0==v; lo1/hi1: outputs with v0+1==v).            */
int vqvae_upsample_linear_fwd(const float* x, int B, int C, int Tin, int Tout,
                              const int32_t* v0, const int32_t* v1, const float* w0,
                              const float* w1,
                              float* y, long y_bstride, vqvae_stream_t s);
int vqvae_upsample_linear_bwd(const float* gy, long gy_bstride, int B, int C, int Tin,
                              int Tout, const float* w0, const float* w1,
                              const int32_t* lo0, const int32_t* hi0,
                              const int32_t* lo1, const int32_t* hi1,
                              float* gx, long gx_bstride, vqvae_stream_t s);
/* the pull-back of L equally shaped tensors in one launch (ratios Tout >= 8 Tin only): tensor l starts gy_lstride elements
 * behind tensor l - 1 (bf16 != 0: 2-byte elements), its result gx_lstride floats behind the previous one -- every block's gh
 * of a ResidualNet once its backward chain has run, instead of a launch per block                                  */
int vqvae_upsample_linear_bwd_blocks(const void* gy, int bf16, long gy_lstride, long gy_bstride, int L, int B, int C,
                                     int Tin, int Tout, const float* w0, const float* w1, const int32_t* lo0,
                                     const int32_t* hi0, const int32_t* lo1, const int32_t* hi1, float* gx,
                                     long gx_lstride, long gx_bstride, vqvae_stream_t s);
/* the same with gy stored as bf16 (VQVAE_STORE_GH_BF16; gy_bstride in elements): ratios Tout >= 8 Tin only */
int vqvae_upsample_linear_bwd_bf16(const void* gy, long gy_bstride, int B, int C, int Tin,
                                   int Tout, const float* w0, const float* w1,
                                   const int32_t* lo0, const int32_t* hi0,
                                   const int32_t* lo1, const int32_t* hi1,
                                   float* gx, long gx_bstride, vqvae_stream_t s);
/* the same with gy stored pre-split (VQVAE_STORE_GH_F16X2; `scale` = its scale words): ratios Tout >= 8 Tin only */
int vqvae_upsample_linear_bwd_f16x2(const void* gy, long gy_bstride, int B, int C, int Tin,
                                    int Tout, const float* w0, const float* w1,
                                    const int32_t* lo0, const int32_t* hi0,
                                    const int32_t* lo1, const int32_t* hi1,
                                    float* gx, long gx_bstride, const uint32_t* scale, vqvae_stream_t s);

/* ---- L.EmbedID + broadcast along T (net.py:57-61): y[b,c,t] = E[id[b],c]      */
int vqvae_embed_broadcast_fwd(const float* E, const int32_t* ids, int B, int G, int T,
                              float* y, long y_bstride, vqvae_stream_t s);
/* gE (n_id,G) (+)= scatter of sum_t gy[b,c,t]; ws >= B*G floats                  */
int vqvae_embed_broadcast_bwd(const float* gy, long gy_bstride, const int32_t* ids,
                              int B, int G, int T, int n_id, float* gE, int accumulate,
                              void* ws, size_t ws_bytes, vqvae_stream_t s);

/* ---- chainer.functions.softmax_cross_entropy (train.py:95, net.py:89):
 *      y (B,q,T), t (B,T) int32; loss = -mean_{b,t} log softmax(y)[t].
 *      fwd writes lse (B,T) and the scalar loss; bwd writes
 *      gy = (softmax - onehot) * (*gloss) / (B*T)   (gloss: device scalar or NULL=1) */
size_t vqvae_softmax_xent_workspace_bytes(int B, int q, int T);
int vqvae_softmax_xent_fwd(const float* y, const int32_t* t, int B, int q, int T,
                           float* lse, float* loss, void* ws, size_t ws_bytes,
                           vqvae_stream_t s);
int vqvae_softmax_xent_bwd(const float* y, const int32_t* t, const float* lse,
                           const float* gloss, int B, int q, int T, float* gy,
                           vqvae_stream_t s);
/* the same, and amax_out (VQVAE_AMAX_SLOTS words, matmul mode 3) receives an upper bound of max |gy|: |*gloss| / (B*T), within
 * a few 1e-3 of the maximum -- the conv that reads gy then needs no scan of it                                   */
int vqvae_softmax_xent_bwd_amax(const float* y, const int32_t* t, const float* lse,
                                const float* gloss, int B, int q, int T, float* gy, uint32_t* amax_out,
                                vqvae_stream_t s);

/* ---- WaveNet.calculate_logistic_loss (WaveNet/modules.py:169-230): discretised
 *      mixture-of-logistics NLL.  y (B, 3*n_mixture, T) = [logit_probs | means |
 *      log_scales], t (B,1,T) fp32 in [-1,1]; loss = -mean_{b,t} logsumexp_k(...).
 *      n_mixture here is the number of logistics (the reference's `n_mixture` argument
 *      is 3x that, params.py:38).  ws >= 4096 floats.                               */
int vqvae_mol_nll_fwd(const float* y, const float* t, int B, int n_mixture, int T, int quantize,
                      float log_scale_min, float* loss, void* ws, size_t ws_bytes,
                      vqvae_stream_t s);
int vqvae_mol_nll_bwd(const float* y, const float* t, const float* gloss, int B, int n_mixture,
                      int T, int quantize, float log_scale_min, float* gy, vqvae_stream_t s);

/* ---- element-wise helpers behind Variable arithmetic (net.py:90-92) and F.relu */
#define VQVAE_EW_ADD       0   /* out = a + b                  */
#define VQVAE_EW_SUB       1   /* out = a - b                  */
#define VQVAE_EW_MUL       2   /* out = a * b                  */
#define VQVAE_EW_AXPBY     3   /* out = alpha*a + beta*b       */
#define VQVAE_EW_SCALE     4   /* out = alpha*a                */
#define VQVAE_EW_SQUARE    5   /* out = a*a                    */
#define VQVAE_EW_RELU      6   /* out = max(a,0)               */
#define VQVAE_EW_RELU_BWD  7   /* out = a * (b > 0)   (a=gy,b=y) */
#define VQVAE_EW_FILL      8   /* out = alpha                  */
#define VQVAE_EW_MUL_SCALAR_DEV 9 /* out = a * b[0] * alpha    */
int vqvae_elementwise(int op, size_t n, const float* a, const float* b, float* out,
                      float alpha, float beta, vqvae_stream_t s);
/* out[0] = scale * sum(x[0..n)) (deterministic two-stage); ws >= 4096 floats     */
int vqvae_sum(const float* x, size_t n, float scale, float* out, void* ws,
              size_t ws_bytes, vqvae_stream_t s);
/* out[0] = mean((a - b)^2) over n elements (net.py:90-91: codebook / commitment loss), with the roundings of the
 * Variable-arithmetic chain it replaces (sub, square, two-stage sum); ws >= 4096 floats.  _bwd: ga = 2 (a - b) gloss[0] / n,
 * gb = -ga (either may be NULL)                                                                              */
int vqvae_sqdiff_mean(const float* a, const float* b, size_t n, float* out, void* ws, size_t ws_bytes, vqvae_stream_t s);
int vqvae_sqdiff_mean_bwd(const float* a, const float* b, const float* gloss, size_t n, float* ga, float* gb, vqvae_stream_t s);

/* ---- device-side input pipeline (the step right before the hot path; utils.py:18-23, 85-110).
 *      mulaw_bins: q[i] = MuLaw(mu).transform(x[i]) by searching host-computed thresholds
 *                  (thresholds[j-1] = smallest fp32 x with transform(x) >= j; bit-exact with NumPy).
 *      onehot    : fp32 (B,q,T) one-hot of idx (B rows of idx_bstride int32).
 *      embed_gather_fwd: the decoder's causal embed conv (modules.py:127-128,151-152) applied to
 *                  an INDEX input instead of the 125.8 MB one-hot tensor:
 *                  y[b,co,t] = b[co] + sum_tap W[co, idx[b,t-(K-1-tap)], tap]; bit-identical to
 *                  vqvae_conv1d_fwd on the one-hot tensor.                                   */
int vqvae_mulaw_bins(const float* x, size_t n, const float* thresholds, int n_thresholds,
                     int32_t* q, vqvae_stream_t s);
int vqvae_onehot(const int32_t* idx, long idx_bstride, int B, int q, int T, float* out,
                 vqvae_stream_t s);
int vqvae_embed_gather_fwd(const int32_t* idx, long idx_bstride, int B, int T, const float* W,
                           const float* b, int Cout, int q, int K, float* y, vqvae_stream_t s);
/*      embed_gather_bound: amax_out (VQVAE_AMAX_SLOTS words) = an upper bound of max |y| of embed_gather_fwd from the
 *                  weights alone, max_co (|b[co]| + sum_tap max_q |W[co, q, tap]|): the scale of the first gate GEMM's
 *                  operand in matmul mode 3 without a scan of y.                                               */
int vqvae_embed_gather_bound(const float* W, const float* b, int Cout, int q, int K, uint32_t* amax_out,
                             vqvae_stream_t s);
/*      embed_onehot_fwd / _wgrad: the SAME conv applied to the reference's one-hot float input
 *                  x (B,q,T) (utils.py:85-87, modules.py:151-152).  fwd scans x on the device
 *                  (idx (B,T) int32 out; *flag = 1 iff every column is exactly one 1.0 and q-1 zeros),
 *                  then launches the gather form (runs iff *flag != 0; bit-identical to the dense
 *                  conv) and the dense conv (runs iff *flag == 0): no host round trip, any other
 *                  input silently takes the dense path.  wgrad likewise: a weighted bincount of the
 *                  output-gradient rows by class (deterministic; LDS histograms, fixed-order batch
 *                  sum; gb = row sums) iff *flag != 0, else the dense weight gradient.  flag == NULL
 *                  (with x == NULL): indices are authoritative (index-fed input), bincount only.
 *                  gW (Cout,q,K), workspace from the query.                                     */
size_t vqvae_embed_onehot_workspace_bytes(int B, int Cout, int q, int K, int T);
int vqvae_embed_onehot_fwd(const float* x, const float* W, const float* b, int B, int Cout, int q,
                           int K, int T, float* y, int32_t* idx, int32_t* flag, void* ws,
                           size_t ws_bytes, vqvae_stream_t s);
int vqvae_embed_onehot_wgrad(const float* x, const int32_t* idx, const int32_t* flag, const float* gy,
                             int B, int Cout, int q, int K, int T, float* gW, float* gb,
                             int accumulate, void* ws, size_t ws_bytes, vqvae_stream_t s);

/* ---- gather n (<= 32) equally sized arrays (host array of device pointers) into one
 *      contiguous array and back (NULL destinations are skipped): lets ResidualNet
 *      treat its blocks' condition_proj parameters as one (n*Cd, Cc) 1x1 conv.       */
int vqvae_concat(float* dst, const float* const* srcs, int n, size_t count, vqvae_stream_t s);
int vqvae_split(const float* src, float* const* dsts, int n, size_t count, int accumulate,
                vqvae_stream_t s);
/* dst[i][0 .. count[i]) = src[i][...] for n independent fp32 arrays (non-overlapping), 64 per launch: the arena adoption of
 * a model's parameters without a copy dispatch per parameter                                                      */
int vqvae_copy_list(int n, float* const* dst, const float* const* src, const size_t* count, vqvae_stream_t s);

/* ---- chainer.optimizers.Adam update rule (train.py:101-102) over a flat arena:
 *      m += (1-b1)(g-m); v += (1-b2)(g*g-v); p -= lr_t * m/(sqrt(v)+eps)
 *      with lr_t = alpha*sqrt(1-b2^t)/(1-b1^t) computed by the host in double;
 *      scalars are rounded to fp32 exactly where NumPy would (weak Python scalars). */
int vqvae_adam_step(float* p, const float* g, float* m, float* v, size_t n,
                    double lr_t, double beta1, double beta2, double eps, vqvae_stream_t s);
/* the same update with lr_t = lr_table[*step] (entries already rounded to fp32 by the host), *step advanced by one
 * behind it: the form a captured training step (vqvae_graph_*) is replayed with -- nothing about the schedule
 * crosses the host-device boundary per step                                                                */
int vqvae_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* lr_table,
                        int32_t* step, double beta1, double beta2, double eps, vqvae_stream_t s);
/* ExponentialMovingAverage (utils.py:151-155): ema = decay*target + (1-decay)*ema */
int vqvae_ema_step(float* ema, const float* target, size_t n, double decay,
                   vqvae_stream_t s);

/* ---- data-parallel gradient exchange: replaces Link.addgrads (sum to main,
 *      updaters.py:71-72) + Link.copyparams (updaters.py:76-77) by one RCCL
 *      all-reduce(sum) of the flat gradient arena on every rank.                */
#define VQVAE_COMM_ID_BYTES 128
int vqvae_comm_unique_id(char id[VQVAE_COMM_ID_BYTES]);          /* host buffer     */
int vqvae_comm_init(void** comm, int nranks, int rank, const char id[VQVAE_COMM_ID_BYTES]);
int vqvae_comm_allreduce_sum_f32(void* comm, float* buf, size_t n, vqvae_stream_t s);
int vqvae_comm_allreduce_max_f32(void* comm, float* buf, size_t n, vqvae_stream_t s);
int vqvae_comm_count(void* comm, int* nranks);   /* ranks RCCL sees in this communicator (ncclCommCount) */
int vqvae_comm_destroy(void* comm);

/* ---- incremental generation: WaveNet.initialize / WaveNet.generate /
 *      ResidualNet.generate / ResidualBlock.push+pop (WaveNet/modules.py:58-74, 98-110,
 *      232-255) and the sampling loop of generate.py:105-145.
 *      One call enqueues ONE audio-sample step for n sequences in lockstep: embed on the
 *      2-sample embed queue, per block the pad-free dilated conv on the queue ends + condition
 *      projection + gate, res/skip 1x1s + queue push, relu/proj1/relu/proj2, then (optionally)
 *      the sampler, the feedback of the sample into the input vector and the advance of the
 *      device-side step counter.  Because the step index lives in device memory the identical
 *      launch sequence serves every step: capture it once with vqvae_graph_* and replay.
 *      "initialize" (modules.py:58-67, 232-244) = the caller zero-fills step, x_cur, x_prev and
 *      every ring.  All weights in their Chainer layout.                                     */
#define VQVAE_GEN_NONE    0   /* no sampling: logits only, the caller writes x_cur (modules.py:246) */
#define VQVAE_GEN_SOFTMAX 1   /* numpy.random.choice over softmax(y) (generate.py:135-141)          */
#define VQVAE_GEN_MOL     2   /* softmax-weighted logistic samples (generate.py:113-133)            */
#define VQVAE_GEN_MAX_N   4
typedef struct {
  const float* conv_W; const float* conv_b;   /* (dilated, residual, 2), (dilated)    modules.py:13-16 */
  const float* cond_W; const float* cond_b;   /* (dilated, cond_dim), (dilated)       modules.py:17-18 */
  const float* res_W;  const float* res_b;    /* (residual, dilated/2), (residual)    modules.py:19-20 */
  const float* skip_W; const float* skip_b;   /* (skip, dilated/2), (skip)            modules.py:21-22 */
  float* ring;       /* (dilation, n, residual): the block's queue minus its newest entry (58-62)   */
  int dilation;
} vqvae_gen_block;
typedef struct {
  int n;                     /* sequences in lockstep, 1..VQVAE_GEN_MAX_N (generate.py:42: 1)        */
  int n_blocks;
  int input_dim, residual, dilated, skip, cond_dim, out_dim;
  int sample_mode;           /* VQVAE_GEN_*                                                          */
  float log_scale_min;       /* params.py:39, generate.py:111                                        */
  const float* embed_W; const float* embed_b;   /* (residual, input_dim, 2)   modules.py:127-128     */
  const float* proj1_W; const float* proj1_b;   /* (skip, skip)               modules.py:135         */
  const float* proj2_W; const float* proj2_b;   /* (out_dim, skip)            modules.py:141         */
  const vqvae_gen_block* blocks;                /* HOST array [n_blocks]                             */
  /* device state, caller-owned */
  int* step;                 /* current step t; advanced by the call                                 */
  float* x_cur;              /* (n, input_dim) input of this step (x_dec, generate.py:52, 127, 139)   */
  float* x_prev;             /* (n, input_dim) older half of the embed queue (modules.py:236-237)     */
  float* h0; float* h1;      /* (n, residual) ping-pong block input/output                           */
  float* z;                  /* (n, dilated/2)                                                       */
  float* skip_acc;           /* (n, skip)                                                            */
  float* s1;                 /* (n, skip)                                                            */
  float* logits;             /* (n, out_dim) = WaveNet.generate's return value                       */
  /* per-step inputs */
  const float* cond;         /* element (b, c) of step t at cond[b*cond_bstride + c*cond_cstride + (cond_follows_step ? t : 0)] */
  long cond_bstride, cond_cstride;
  int cond_follows_step;
  const double* uniforms;    /* (max_steps, n, n_uniform) doubles in [0,1): what numpy.random would draw */
  int n_uniform;
  const void* forced_next;   /* NULL, or (max_steps, n) int32 [softmax] / float [mol]: fed back instead of
                                the sample (teacher forcing; index -1 = all-zero vector)              */
  /* outputs */
  void* out;                 /* NULL, or sample of step t at out[b*out_bstride + t] (int32 / float)  */
  long out_bstride;
  float* logits_out;         /* NULL, or (max_steps, n, out_dim)                                     */
  int max_steps;             /* calls with *step >= max_steps are no-ops on the device               */
} vqvae_gen_desc;
int vqvae_wavenet_gen_step(const vqvae_gen_desc* d, vqvae_stream_t s);

/* The same loop (generate.py:105-145) as ONE persistent launch: steps [t0, t0+steps) for fresh
 * queues at t0 == 0 (the call zero-fills `ws`: that is WaveNet.initialize) or continuing a
 * previous call on the same `ws`, `out` and `forced_next`.  Uses the weights, cond, uniforms,
 * forced_next, out, logits_out, sample_mode (SOFTMAX or MOL) fields of `d`; the device state
 * fields are not used (queues and hand-off mailboxes live in `ws`).  Channel counts <= 256.
 * After the stream has drained, the first int32 of `ws` is 0 on success, non-zero when a bounded
 * device-side wait timed out (the launch then ended early).                                    */
size_t vqvae_wavenet_gen_run_workspace_bytes(const vqvae_gen_desc* d);
int vqvae_wavenet_gen_run(const vqvae_gen_desc* d, int t0, int steps, void* ws, size_t ws_bytes,
                          vqvae_stream_t s);

/* ---- hipGraph capture/replay of a launch sequence on one stream (launch-bound inner loops:
 *      the per-sample chain of generate.py:105-145)                                          */
int vqvae_graph_capture_begin(vqvae_stream_t s);
/* the same with hipStreamCaptureModeRelaxed: the host side may allocate while it records (a whole training step,
 * updaters.py:13-19, driven by the Python-level graph of FunctionNodes)                                    */
int vqvae_graph_capture_begin_relaxed(vqvae_stream_t s);
int vqvae_graph_capture_end(vqvae_stream_t s, void** graph_exec);
int vqvae_graph_launch(void* graph_exec, vqvae_stream_t s);
int vqvae_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* VQVAE_HIP_H_ */
