#!/usr/bin/env python
"""bench.py -- audio samples/s of the VQ-VAE training hot path on MI355X.

One "step" = one VQVAE updater iteration over one synthetic minibatch: VAE
forward (encoder -> VQ -> condition embed -> WaveNet -> softmax-CE), the
three-loss backward of updaters.py:13-19, the EMA blend (utils.py:146-155), the
RCCL gradient all-reduce when N > 1, and the Adam update.  Nothing is skipped in
the timed region.  Inputs are resident in HBM before the timed region starts.

Workload (N=1): BASELINE.json configs[1] -- batch 16, length 7680, mu-law q=256,
d=64, k=512, n_loop=2, n_layer=10, residual=dilated=skip=256, condition 64+128,
fp32.  N>1: weak scaling, 16 samples per GPU (configs[2]), one process per GPU.
`python bench.py --gpus N` spawns its own N ranks (RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment); when it finds itself
already launched (`python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N`, WORLD_SIZE set) it is one of the ranks.  torch is never imported.

--workload c4 is BASELINE configs[3] (VQ stress, k=8192 d=128; SURVEY 8d inputs):
a "step" is one nearest-codebook search + gather over N latent rows; the line
reports the expansion-form MFMA rate and the algorithmic HBM GB/s side by side.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     -- the north star's kernel (ResidualBlock forward: dilated conv +
                  condition step + gate): algorithmic FLOPs per launch / its mean
                  launch time from the dispatch's own HIP events.  It is no longer
                  the largest consumer of the step: roofline.kernels lists EVERY
                  kernel family of ResidualNet's chain (gate, res 1x1, skip sum,
                  gate derivative, backward-data, both weight-gradient families)
                  with launches per step, mean launch time, algorithmic work per
                  launch, its bound (mfma / hbm) and fraction, sorted by time per step.
  cpu_baseline -- the NumPy oracle (a port of the Chainer-CPU algorithm) timed on
                  this box's host cores on a bounded sample (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'chainer-vq-vae_amd'))

import numpy as np  # noqa: E402

CFG = dict(d=64, k=512, n_loop=2, n_layer=10, filter_size=2, input_dim=256, quantize=256,
           residual=256, dilated=256, skip=256, local_dim=64, global_dim=128, n_speaker=109,
           length=7680, beta=0.25, lr=2e-4, ema_mu=0.9999, batch_per_gpu=16)
PEAK_FP32_MFMA_TFLOPS = 157.3       # /opt/skills/guides/MI355X_MICROARCH.md chip table

# The north star's kernel (`roofline`) and what one launch of it computes.  While the
# residual 1x1 conv is a separate launch the extra term is 0.
ROOFLINE_KERNEL = ('conv_gemm_kernel<EPI_GATE> (ResidualBlock fwd: dilated causal conv k=2 as MFMA GEMM '
                   '+ latent-rate condition lerp + tanh*sigmoid gate)')
ROOFLINE_KERNEL_BF16 = ('conv_gemm_x3_kernel<EPI_GATE, 256 x 128 tiles, two 8-wave workgroups per CU, NP = 1> '
                        '(ResidualBlock fwd: dilated causal conv k=2 as MFMA GEMM on bf16-rounded operands, fp32 accumulate '
                        '+ latent-rate condition lerp + tanh*sigmoid gate epilogue; x, z, gates stored as bf16)')
ROOFLINE_KERNEL_X3 = ('conv_gemm_x3_kernel<EPI_GATE, 256 x 128 tiles, two 8-wave workgroups per CU, NP = 3> '
                      '(ResidualBlock fwd: dilated causal conv k=2 as MFMA GEMM, '
                      'every fp32 product = 6 bf16 MFMA products of an exact 3-way operand split, fp32 accumulate '
                      '+ the latent-rate condition'"'"'s lerp as one more K step + tanh*sigmoid gate epilogue; weights by LDS-DMA)')
ROOFLINE_KERNEL_X2 = ('conv_gemm_x3_kernel<EPI_GATE, 256 x 128 tiles, two 8-wave workgroups per CU, NP = 2> '
                      '(ResidualBlock fwd: dilated causal conv k=2 as MFMA GEMM, '
                      'every fp32 product = 3 fp16 MFMA products of a scaled hi + lo operand split, fp32 accumulate '
                      '+ the latent-rate condition'"'"'s lerp as one more K step + tanh*sigmoid gate epilogue; weights by LDS-DMA)')
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense (bf16 and fp16 run at the same rate)
# what the whole chip's matrix pipe sustains on the product stream alone with REAL operands (N(0,1) values split into
# their pieces; registers only, no LDS / memory): tools/ubench/mfma_power.hip (six bf16 products, profiles/r3/
# ubench_mfma_power.txt) and tools/ubench/f16x2_probe.hip (three fp16 products, profiles/r4/ubench_f16x2_probe.txt);
# zeros: 2 130-2 180 / 1 830: the power limit.  Reported beside the nominal peak, never instead of it.
SUSTAINED_MFMA_TFLOPS_REAL_DATA = {'float32x3': 1670.0, 'float32x2': 1577.0}
PRODUCTS_PER_FP32 = {'float32x3': 6, 'float32x2': 3}     # 16-bit MFMA products per algorithmic fp32 product


def FUSED_RES_FLOP_PER_POS(cfg):
    return 0.0


def kernel_roofline_table(upd, backend, cfg, B, mode, k_eager):
    """SURVEY 8(d) for every kernel family of ResidualNet's chain, not only the gate kernel: `k_eager` eager steps of the
    same job with the library's per-launch profiler on for every chain tag (dispatch events for the GEMM launches,
    stream events around a weight-gradient launch and its fixed-order reduce), then per family: launches per step, mean
    launch time, ALGORITHMIC work per launch (family total per step / launches per step, so a family whose launches
    differ -- the first / last block -- is averaged, not idealised), the roofline that bounds it and the fraction.
    Runs outside the timed region."""
    import ctypes as C
    from vqvae_amd import _lib
    lib = _lib.load()
    T = cfg['length']
    N = float(B * T)
    nb = cfg['n_loop'] * cfg['n_layer']
    Cr, Cd, Cs, Kf = cfg['residual'], cfg['dilated'], cfg['skip'], cfg['filter_size']
    Ch = Cd // 2
    mfma_peak = (PEAK_BF16_MFMA_TFLOPS / PRODUCTS_PER_FP32[mode]) if mode in PRODUCTS_PER_FP32 else PEAK_FP32_MFMA_TFLOPS
    fam = [
        # (name, tag, bound, family total per step [FLOP or bytes], what the total is)
        ('ResidualBlock fwd: dilated conv + condition step + gate (conv_gemm_x3_kernel<EPI_GATE>)', _lib.PROF_RESBLOCK_GATE, 'mfma',
         nb * 2.0 * N * Cd * Cr * Kf, 'n_blocks * 2 N Cd Cr K'),
        ('ResidualBlock fwd: res 1x1 + residual add (lin128_stream_kernel)', _lib.PROF_RESBLOCK_OUT, 'hbm',
         (nb - 1) * 4.0 * N * (Ch + Cr + Cr), '(n_blocks - 1) * 4 N (Ch + Cr + Cr): z and x_l read, x_{l+1} written'),
        ('ResidualNet fwd: skip sum over all blocks as one GEMM (conv_gemm_x3_kernel<EPI_LINEAR>)', _lib.PROF_RESSTACK_SKIP, 'mfma',
         2.0 * N * Cs * Ch * nb, '2 N Cs Ch n_blocks'),
        ('ResidualBlock bwd: gz = Wr^T g_res + Ws^T g_skip, gate derivative -> gh, latent pull-back (conv_gemm_x3_kernel<EPI_GATE_BWD>)',
         _lib.PROF_RESBLOCK_BWD_GZ, 'hbm',
         4.0 * N * (nb * (Cs + Ch + Ch + Cd) + (nb - 1) * Cr),
         '4 N (n_blocks (Cs + Ch + Ch + Cd) + (n_blocks - 1) Cr): g_skip, sigmoid, z read, gh written; g_res read by all but the last block'),
        ('ResidualBlock bwd: backward-data of the dilated conv + g_res (conv_gemm_x3_kernel<EPI_LINEAR>, two taps)', _lib.PROF_RESBLOCK_BWD_GX, 'mfma',
         nb * 2.0 * N * Cr * Cd * Kf, 'n_blocks * 2 N Cr Cd K'),
        ('ResidualNet bwd: weight gradients of the dilated convs, several blocks per launch, incl. the fixed-order reduce (wgrad3_dma_kernel: both operands stored pre-split / as bf16, global -> LDS by LDS-DMA; wgrad3_kernel otherwise)',
         _lib.PROF_WGRAD_DIL, 'mfma', nb * 2.0 * N * Cd * Cr * Kf, 'n_blocks * 2 N Cd Cr K'),
        ('ResidualNet bwd: weight gradients of the res / skip 1x1 convs, incl. the reduce (wgrad3_kernel)', _lib.PROF_WGRAD_RES_SKIP, 'mfma',
         (2 * nb - 1) * 2.0 * N * Cr * Ch, '(2 n_blocks - 1) * 2 N 256 Ch'),
    ]
    mask = 0
    for f in fam:
        mask |= 1 << f[1]
    lib.vqvae_prof_reset()
    lib.vqvae_prof_enable(mask)
    t0 = time.perf_counter()
    for _ in range(k_eager):
        upd.update()
    backend.synchronize()
    ms_step = 1e3 * (time.perf_counter() - t0) / k_eager
    lib.vqvae_prof_enable(0)
    rows = []
    for name, tag, bound, total, what in fam:
        tot, cnt = C.c_double(0), C.c_int(0)
        _lib.call('vqvae_prof_read', tag, C.byref(tot), C.byref(cnt))
        if not cnt.value:
            continue
        per_step = cnt.value / float(k_eager)
        avg_ms = tot.value / cnt.value
        work = total / per_step
        if bound == 'mfma':
            ach = work / (avg_ms * 1e-3) / 1e12
            row = {'flop_per_launch': work, 'achieved': ach, 'peak': mfma_peak, 'unit': 'TFLOP/s', 'frac': ach / mfma_peak}
        else:
            ach = work / (avg_ms * 1e-3) / 1e9
            row = {'bytes_per_launch': work, 'achieved': ach, 'peak': 8000.0, 'unit': 'GB/s', 'frac': ach / 8000.0}
        row.update({'name': name, 'bound': bound, 'launches_per_step': per_step, 'avg_launch_ms': avg_ms,
                    'ms_per_step': avg_ms * per_step, 'share_of_eager_step': avg_ms * per_step / ms_step, 'work_is': what})
        rows.append(row)
    lib.vqvae_prof_reset()
    rows.sort(key=lambda r: -r['ms_per_step'])
    return rows, ms_step


def synth_examples(B, cfg, seed):
    """Preprocess's output contract (utils.py:99-110) on synthetic audio:
    (raw (1,L+1,1) f32, one_hot[:, :-1] (q,L,1) f32, speaker () i32, quantized[1:] (L,1) i32)."""
    from vqvae_amd.utils import MuLaw
    rs = np.random.RandomState(seed)
    L = cfg['length'] + 1
    n = np.arange(L) / 16000.0
    mu = MuLaw(cfg['quantize'])
    ex = []
    eye = np.identity(cfg['quantize'], dtype=np.float32)
    for _ in range(B):
        f = rs.uniform(80, 4000, 3)
        ph = rs.uniform(0, 2 * np.pi, 3)
        a = rs.uniform(0.2, 1.0, 3)
        raw = sum(a[i] * np.sin(2 * np.pi * f[i] * n + ph[i]) for i in range(3))
        raw = raw + 0.05 * rs.standard_normal(L)
        raw = (raw / np.abs(raw).max()).astype(np.float32)
        q = mu.transform(raw)
        spk = np.array(rs.randint(0, cfg['n_speaker']), np.int32)
        raw3 = raw[None, :, None]
        if cfg.get('use_logistic', False):          # utils.py:104, 107: raw in, raw target
            ex.append((raw3, raw3[:, :-1], spk, raw3[:, 1:]))
        else:
            one_hot = np.expand_dims(eye[q].T, 2)
            ex.append((raw3, one_hot[:, :-1], spk, np.expand_dims(q, 1)[1:]))
    return ex


class ResidentIterator(object):
    """Yields this rank's shard as device-resident arrays (already concatenated):
    the timed region starts with inputs in HBM."""

    yields_rank_shard = True      # VQVAE_ParallelUpdater takes next() as this rank's shard, no batch[rank::n] on a global batch

    class _Shard(object):
        def __init__(self, arrays):
            self.arrays = arrays

    def __init__(self, shards):
        self.shards = [self._Shard(s) for s in shards]
        self.i = 0

    def next(self):
        s = self.shards[self.i % len(self.shards)]
        self.i += 1
        return s


def resident_converter(batch, device):
    return batch.arrays


def build(cfg, n_gpus):
    import vqvae_amd as V
    from vqvae_amd import functions as F
    from vqvae_amd.optimizers import Adam
    V.core.seed_initializers(0)
    encoder = V.Encoder(cfg['d'])
    logistic = cfg.get('use_logistic', False)
    wavenet = V.WaveNet(cfg['n_loop'], cfg['n_layer'], cfg['filter_size'], cfg['input_dim'],
                        cfg['residual'], cfg['dilated'], cfg['skip'], cfg['quantize'], logistic,
                        cfg.get('n_mixture', 30), -40, cfg['local_dim'] + cfg['global_dim'], 0)
    cond = V.ConditionEmbed(cfg['n_speaker'], cfg['global_dim'], cfg['local_dim'])
    decoder = V.ExponentialMovingAverage(wavenet, cfg['ema_mu'])       # train.py:87-90
    loss_fun = wavenet.calculate_logistic_loss if logistic else F.softmax_cross_entropy   # train.py:92-95
    model = V.VAE(encoder, decoder, cond, cfg['d'], cfg['k'], cfg['beta'], loss_fun)
    return model, Adam(cfg['lr'] / n_gpus)                              # train.py:101


def cpu_baseline(cfg):
    """Times the NumPy oracle (port of the Chainer-CPU algorithm) at the
    reference's CPU-runnable shape (configs[0]: batch 1, length 7680)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import vqvae_oracle as O
    rs = np.random.RandomState(0)
    P = O.make_params(rs, d=cfg['d'], k=cfg['k'], n_loop=cfg['n_loop'], n_layer=cfg['n_layer'],
                      residual=cfg['residual'], dilated=cfg['dilated'], skip=cfg['skip'],
                      local_dim=cfg['local_dim'], global_dim=cfg['global_dim'],
                      n_speaker=cfg['n_speaker'])
    batch = O.synth_batch(1, length=cfg['length'], n_speaker=cfg['n_speaker'], seed=71)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count()
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:
        threadpool_limits = None
    # BLAS with every core is not the CPU's best on these mid-sized GEMMs: try a few thread
    # counts (one warm-up + two timed steps each) and report the fastest
    candidates = sorted({c for c in (cores, 64, 32, 16) if c <= cores}, reverse=True)
    if threadpool_limits is None:
        candidates = [cores]

    def timed(nthr, n_timed):
        state = {}
        ctx = threadpool_limits(limits=nthr) if threadpool_limits else None
        try:
            if ctx is not None:
                ctx.__enter__()
            O.train_step(P, state, batch, cfg['n_loop'], cfg['n_layer'])    # warm-up
            times = []
            for _ in range(n_timed):
                t0 = time.time()
                O.train_step(P, state, batch, cfg['n_loop'], cfg['n_layer'])
                times.append(time.time() - t0)
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
        return times
    # pick the thread count on one timed step each, then 1 warm-up + 3 timed steps at the best
    # (median of >= 3 steps after a warm-up)
    probe = {nthr: timed(nthr, 1)[0] for nthr in candidates}
    nthr = min(probe, key=probe.get)
    times = timed(nthr, 3)
    med = float(np.median(times))
    model = 'unknown'
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.lower().startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return {'value': cfg['length'] / med, 'unit': 'samples/s', 'cores': nthr, 'cores_is': 'BLAS threads used',
            'cores_visible': cores, 'cpu_model': model, 'kind': 'port',
            'sample': '%d full training steps (fwd + 3-loss bwd + Adam + EMA) at batch 1, length %d '
                      '(BASELINE configs[0]) after 1 warm-up, median %.2f s/step with %d BLAS threads '
                      '(thread count chosen from %s on one step each; %d cores visible); NumPy '
                      'restatement of the Chainer-CPU algorithm'
                      % (len(times), cfg['length'], med, nthr, candidates, cores)}


def kernel_source_hash():
    """sha256 over the HIP sources whose kernels the roofline/traffic figures describe."""
    import hashlib
    h = hashlib.sha256()
    for name in ('gemm_common.h', 'conv_gemm_x3.hip', 'vq.hip', 'common.h'):
        with open(os.path.join(ROOT, 'chainer-vq-vae_amd', 'csrc', name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def GATES_SIG_ACTIVE(cfg, B):
    """Does ResidualNet's chain save sigmoid and z only (vqvae_resblock_desc.storage & VQVAE_STORE_GATES_SIG: float32x2,
    configs-sized blocks)?  Asked of the library, as wavenet.py does."""
    import ctypes as C
    from vqvae_amd import _lib
    d = _lib.ResblockDesc(B, cfg['length'], cfg['residual'], cfg['dilated'], cfg['skip'], 192, cfg['filter_size'], 1, 0)
    return bool(_lib.load().vqvae_resblock_f16x2_storage(C.byref(d)) & _lib.STORE_GATES_SIG)


def GATE_BYTES(cfg, B, bf16=False):
    """Algorithmic HBM bytes of one gate launch: 4 * (N Cr + N 1.5 Cd + K Cr Cd + B Cd T'); in
    the bf16 mode of the configs-sized blocks the gate values, z and the residual stream x are kept as bf16: 2 bytes each.
    Where the chain saves sigmoid and z only (float32x2, round 5) the stores are N Cd instead of N 1.5 Cd."""
    N = B * cfg['length']
    out_bytes = 2.0 if (bf16 and cfg['dilated'] == 256 and cfg['residual'] == 256 and cfg['length'] % 64 == 0) else 4.0
    x_bytes = 2.0 if (out_bytes == 2.0 and cfg['length'] % 128 == 0) else 4.0      # the bf16 residual stream (every block but the first)
    saved = 1.0 if (not bf16 and GATES_SIG_ACTIVE(cfg, B)) else 1.5
    return (x_bytes * N * cfg['residual'] + out_bytes * N * saved * cfg['dilated']
            + 4.0 * (cfg['filter_size'] * cfg['residual'] * cfg['dilated'] + B * cfg['dilated'] * (cfg['length'] // 64)))


def measured_traffic(key):
    """HBM bytes per launch of the gate kernel from the committed rocprofv3 --pmc summary
    (profiles/roofline_traffic.json, written by tools/pmc_traffic.py from the counter CSVs of
    THIS command).  PMC counters cannot be read from inside the process, so the figure is only
    reported while the summary was taken on the same kernel sources (hash stamp); otherwise null."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'roofline_traffic.json')) as fh:
            rec = json.load(fh)[key]
    except Exception:
        return None, 'no rocprofv3 --pmc summary for this workload under profiles/'
    if rec.get('kernel_source_sha256_16') != kernel_source_hash():
        return None, ('profiles/roofline_traffic.json[%s] was measured on other kernel sources '
                      '(stale): re-run tools/pmc_run.sh + tools/pmc_traffic.py' % key)
    return rec['hbm_traffic_bytes_per_launch'], rec.get('source', 'profiles/roofline_traffic.json')


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` from a bare shell: start N ranks of this script (one per GPU),
    hand them RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* and a private rendezvous id, relay their
    output, and fail if any rank fails.  No torch, no external launcher."""
    import socket
    import subprocess
    import uuid
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    rdzv = uuid.uuid4().hex
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), VQVAE_RDZV_ID=rdzv,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                r = p.poll()
                if r is None:
                    continue
                pending.remove(p)
                if r != 0:
                    rc = rc or r
                    for q in pending:           # one rank died: the others would hang in the collective
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raise SystemExit(rc)


NO_AFFINITY = [False]      # --no-affinity


def pin_rank_to_cores(local_rank, n_local):
    """One contiguous slice of the visible cores per rank (the host side of a step is ~500 Python-driven
    kernel launches: eight unpinned ranks migrating across sockets show up as rank skew).  Returns the
    (first, last, count) it pinned to, or None when the platform has no sched_setaffinity / one rank."""
    if n_local <= 1 or not hasattr(os, 'sched_setaffinity') or NO_AFFINITY[0]:
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = len(cores) // n_local
        if per < 1:
            return None
        mine = cores[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, mine)
        return [mine[0], mine[-1], len(mine)]
    except OSError:
        return None


def vq_stress_inputs(N, d, k):
    """SURVEY 8d C4 inputs: half the rows N(0,1), half W[j] + 0.5 N(0,1); W ~ N(0, 1/d); seeds 1/2."""
    rw = np.random.RandomState(2)
    W = (rw.standard_normal((k, d)) / np.sqrt(d)).astype(np.float32)
    rz = np.random.RandomState(1)
    rows = rz.standard_normal((N, d)).astype(np.float32)
    j = rz.randint(0, k, size=N - N // 2)
    rows[N // 2:] = W[j] + np.float32(0.5) * rz.standard_normal((N - N // 2, d)).astype(np.float32)
    return rows, W


def run_c4(args, rank, n, local):
    """BASELINE configs[3]: large-codebook VQ stress, k=8192 d=128.  One step = one
    StraightThrough.forward (nearest code + gather, utils.py:176-211) over N latent rows laid out
    (B, d, T'=120) like the encoder output.  Ranks are independent (each quantises its own rows)."""
    import ctypes as C
    from vqvae_amd import _lib, backend
    from vqvae_amd.backend import DeviceArray
    from vqvae_amd.comm import RcclCommunicator, SingleCommunicator
    backend.init(local)
    if getattr(args, 'matmul', None):
        backend.set_matmul_dtype(args.matmul)
    comm = RcclCommunicator(rank, n, local) if (n > 1 or args.force_comm) else SingleCommunicator()
    d, k, T = 128, 8192, 120
    N = args.vq_rows
    if N % T:
        raise SystemExit('--vq-rows must be a multiple of 120 (latents per 7680-sample crop)')
    B = N // T
    rows, W = vq_stress_inputs(N, d, k)
    z = np.ascontiguousarray(rows.reshape(B, T, d).transpose(0, 2, 1))
    dz, dW = backend.to_device(z), backend.to_device(W)
    idx = DeviceArray((B, T), np.int32)
    e = DeviceArray((B, d, T), np.float32)
    nre = DeviceArray((1,), np.int32)
    ws = backend.workspace(_lib.load().vqvae_vq_workspace_bytes(B, d, T, k))

    def step():
        _lib.call('vqvae_vq_nearest_fwd', dz.ptr, dW.ptr, B, d, T, k, 0, idx.ptr, e.ptr, nre.ptr,
                  ws.ptr, ws.nbytes, backend.stream())
    for _ in range(args.warmup):
        step()
    backend.synchronize()
    lib = _lib.load()
    tag = _lib.PROF_VQ_NEAREST
    lib.vqvae_prof_reset()
    lib.vqvae_prof_enable(1 << tag)
    comm.barrier()
    backend.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    backend.synchronize()
    comm.barrier()
    dt = comm.max_scalar(time.perf_counter() - t0)
    lib.vqvae_prof_enable(0)
    tot, cnt = C.c_double(0), C.c_int(0)
    _lib.call('vqvae_prof_read', tag, C.byref(tot), C.byref(cnt))
    seen = comm.ranks_seen()
    if rank == 0:
        mode = backend.default_matmul_dtype() if not getattr(args, 'matmul', None) else args.matmul
        x3 = mode in ('float32x3', 'float32x2')
        # 'float32x2': three fp16 products of the scaled two-piece split since round 5 ('float32x3': six bf16 products)
        X3_PRODUCTS = 3 if mode == 'float32x2' else 6
        peak = PEAK_BF16_MFMA_TFLOPS / X3_PRODUCTS if x3 else PEAK_FP32_MFMA_TFLOPS
        flop = 2.0 * N * k * d                              # expansion form, re-check not counted
        byts = 4.0 * (N * d + k * d + N + N * d)            # z read, codebook once, idx write, e write
        avg_ms = tot.value / max(cnt.value, 1)
        ach = flop / (avg_ms * 1e-3) / 1e12 if cnt.value else None
        gbs = byts / (avg_ms * 1e-3) / 1e9 if cnt.value else None
        traffic, tsrc = measured_traffic('c4_N%d' % N)
        # sanity of the result itself on a bounded sample: exact reference distances for 64 rows
        got = idx.get().reshape(-1)
        pick = np.linspace(0, N - 1, 64).astype(np.int64)
        ok = True
        for r in pick:
            dist = np.zeros(k, np.float32)
            for c in range(d):
                dist = dist + (rows[r, c] - W[:, c]) ** 2   # utils.py:189-203 summation order
            ok = ok and int(np.argmin(dist)) == int(got[r])
        out = {
            'metric': 'VQ nearest-codebook lookups/sec (k=8192, d=128), whole job',
            'value': n * N * args.steps / dt, 'unit': 'latent rows/s', 'n_gpus': n,
            'ranks_seen_by_rccl': seen, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[3]: VQ stress k=8192 d=128, N=%d latent rows/GPU '
                                   '(B=%d x T\'=120), half N(0,1) half near-codebook rows' % (N, B),
                       'parallelism': 'dp%d (independent rows per rank, no data-path collective)' % n},
            'rows_rechecked_exactly': int(nre.get()[0]),
            'indices_match_reference_order_distance_on_64_rows': bool(ok),
            'matmul': mode,
            'roofline': {'bound': 'mfma', 'kernel': 'vqvae_vq_nearest_fwd (vq_wnorm + %s + vq_exact_batched_kernel '
                         '+ gather): MFMA pairwise distance, wavefront argmin, exact re-check of ambiguous rows'
                         % ('vq_wsplit + vq_mfma_x3_kernel' if x3 else 'vq_mfma_reg_kernel'),
                         'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                         'peak_is': ('dense 16-bit MFMA peak / %d MFMA products per fp32 product; achieved counts '
                                     'algorithmic fp32 FLOPs' % X3_PRODUCTS if x3 else 'fp32 MFMA peak'),
                         'frac': (ach / peak) if ach else None,
                         'achieved_vs_fp32_mfma_peak': (ach / PEAK_FP32_MFMA_TFLOPS) if ach else None,
                         'traffic': traffic, 'traffic_source': tsrc,
                         'launches': cnt.value, 'avg_launch_ms': avg_ms, 'flop_per_launch': flop,
                         'hbm': {'achieved_algorithmic': gbs, 'peak': 8000.0, 'unit': 'GB/s',
                                 'frac': (gbs / 8000.0) if gbs else None,
                                 'algorithmic_bytes_per_launch': byts}},
        }
        print(json.dumps(out))
    if n > 1 or args.force_comm:
        comm.barrier()
        comm.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--batch', type=int, default=CFG['batch_per_gpu'])
    ap.add_argument('--workload', choices=['c2', 'c4', 'c5'], default='c2',
                    help='c2: BASELINE configs[1]/[2] (softmax, 20 blocks, fp32) -- the metric config; '
                         'c4: configs[3] (VQ stress k=8192 d=128; see --vq-rows); '
                         'c5: configs[4] (mixture of logistics, input_dim=1, n_loop=4 -> 40 blocks)')
    ap.add_argument('--vq-rows', type=int, default=1048560,
                    help='c4: latent rows N per GPU (multiple of 120); SURVEY 8d sizes are 1920 (training '
                         'shape, latency-bound) and ~1 M (default 8738 x 120 = 1 048 560)')
    ap.add_argument('--bf16', action='store_true',
                    help='bf16 MFMA operands with fp32 accumulation (configs[4] precision)')
    ap.add_argument('--matmul', choices=['float32x2', 'float32x3', 'float32'], default=None,
                    help="fp32 matmul mode: 'float32x2' (default; fp32 products as three fp16 MFMA products of scaled two-piece operands), 'float32x3' (fp32 products as six bf16 MFMA products of an "
                         "exact three-way operand split) or 'float32' (v_mfma_f32_32x32x2_f32)")
    ap.add_argument('--index-input', action='store_true', help='(the default since round 4; kept for the scripts)')
    ap.add_argument('--onehot-input', action='store_true',
                    help="feed x_dec as the reference's resident fp32 one-hot (utils.py:85-87) instead of device-computed "
                         "bin indices")
    ap.add_argument('--graph', action='store_true', help='replay the step as a hipGraph also with a communicator (n > 1)')
    ap.add_argument('--no-graph', action='store_true', help='eager launches every step')
    ap.add_argument('--no-fresh-input', action='store_true',
                    help='skip the second timed region (ms_per_step_with_input: a fresh host minibatch every step)')
    ap.add_argument('--no-kernel-table', action='store_true', help='skip roofline.kernels (the eager pass with every chain kernel timed)')
    ap.add_argument('--no-overlap', action='store_true',
                    help='single stream (the default since the float32x3 kernels; kept for the profile scripts)')
    ap.add_argument('--overlap', action='store_true',
                    help='weight gradients of the backward pass on a second stream')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help="N > 1: 'weak' = --batch samples per GPU (configs[2]: 16/GPU, global 128 at N = 8); "
                         "'strong' = global batch 128 split over the N ranks (128/N per GPU)")
    ap.add_argument('--overlap-comm', action='store_true',
                    help='exchange the decoder / condition-embed gradient bucket on the side stream while the '
                         'codebook and commitment losses still back-propagate (VQVAE_ParallelUpdater(overlap_comm=True)); '
                         'the default with more than one rank')
    ap.add_argument('--no-overlap-comm', action='store_true', help='N > 1: the whole arena in one all-reduce on the main stream')
    ap.add_argument('--no-affinity', action='store_true', help='N > 1: leave CPU affinity and NUMA policy alone')
    ap.add_argument('--force-comm', action='store_true',
                    help='use the RCCL communicator even with one rank (bootstrap self-test)')
    args = ap.parse_args()
    world_env = os.environ.get('WORLD_SIZE')
    if args.gpus > 1 and world_env is None:
        spawn_ranks(args.gpus, sys.argv[1:])           # does not return
    cfg = dict(CFG)
    cfg['batch_per_gpu'] = args.batch
    if args.workload == 'c5':
        cfg.update(n_loop=4, input_dim=1, use_logistic=True, n_mixture=30)
    rank = int(os.environ.get('RANK', 0))
    world = int(world_env or 1)
    local = int(os.environ.get('VQVAE_LOCAL_DEVICE', os.environ.get('LOCAL_RANK', 0)))   # override: multi-rank dry runs on one GPU
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    n = world
    try:
        cpus_at_start = set(os.sched_getaffinity(0))
    except AttributeError:
        cpus_at_start = None
    NO_AFFINITY[0] = args.no_affinity
    affinity = pin_rank_to_cores(local, n)       # (replaced below by the GPU's NUMA node when sysfs names one)
    if args.scaling == 'strong':
        if 128 % n:
            raise SystemExit('bench.py: --scaling strong splits a global batch of 128: --gpus must divide it')
        cfg['batch_per_gpu'] = 128 // n
    if args.workload == 'c4':
        return run_c4(args, rank, n, local)

    import vqvae_amd as V
    from vqvae_amd import _lib, backend
    from vqvae_amd.comm import RcclCommunicator, SingleCommunicator
    backend.init(local)
    # N > 1: this rank's host memory and threads next to its GPU (the page-locked input buffers, the Python heap)
    numa = None
    if n > 1 and not args.no_affinity:
        import ctypes as C
        from vqvae_amd.comm import bind_to_numa_node, gpu_numa_node
        bus = C.create_string_buffer(32)
        if _lib.load().vqvae_device_pci_bus_id(bus, 32) == 0:
            numa = bind_to_numa_node(gpu_numa_node(bus.value.decode()), cpus_at_start)
            numa['pci_bus_id'] = bus.value.decode()
    args.overlap_comm = (args.overlap_comm or n > 1) and not args.no_overlap_comm
    if args.bf16:
        backend.set_matmul_dtype('bfloat16')
    elif args.matmul:
        backend.set_matmul_dtype(args.matmul)
    mode = 'bfloat16' if args.bf16 else (args.matmul or backend.default_matmul_dtype())
    if args.no_overlap:
        backend.set_overlap(False)
    if args.overlap:
        backend.set_overlap(True)
    comm = RcclCommunicator(rank, n, local) if (n > 1 or args.force_comm) else SingleCommunicator()

    # train.py's order (train.py:76-102): construct -> to_gpu -> optimizer.setup -> train.  The
    # condition embed's lazily shaped convs (net.py:34-43) appear at the first forward and are
    # adopted by the optimizer there (optimizers.Adam.adopt_new_params) -- the warm-up steps.
    model, opt = build(cfg, n)
    model.to_gpu(local)
    opt.setup(model)
    # start-up self-test of the exchange (rank ids through sum / max all-reduces, identical parameters on every rank)
    comm_self_test = comm.self_test(opt.params) if hasattr(comm, 'self_test') else None

    B = cfg['batch_per_gpu']
    # The feed.  Default: the device-side input pipeline's form (raw crops -> bin INDICES on the device);
    # --onehot-input: the reference's call surface (a 125.8 MB fp32 one-hot x_dec, utils.py:85-87), which the device
    # scans back into indices every step.  The mixture-of-logistics workload feeds raw samples either way.
    index_input = (not args.onehot_input) and not cfg.get('use_logistic', False)
    shards, host_batches = [], []
    for s in range(2):                      # two distinct resident minibatches, alternated
        ex = synth_examples(B, cfg, seed=71 + 1000 * rank + s)
        raw = np.stack([e[0][0, :, 0] for e in ex])
        host_batches.append((raw, np.array([e[2] for e in ex], np.int32)))
        if index_input:
            from vqvae_amd.inputs import DeviceInputPipeline
            shards.append(DeviceInputPipeline(cfg['quantize'])(raw, host_batches[-1][1]))
        else:
            shards.append(V.concat_examples(ex, device=local))
    it = ResidentIterator(shards)
    # The step as ONE hipGraph launch (updaters.GraphedStep: recorded after two eager steps, bit-identical to the eager
    # step -- tests/test_gpu_model.py::test_graphed_step_equals_eager_step).  Single-rank runs by default; --graph forces
    # it with a communicator too (RCCL inside a capture has not run on this pool), --no-graph keeps every step eager.
    use_graph = (not args.no_graph) and (args.graph or (n == 1 and not args.force_comm))
    upd = V.VQVAE_ParallelUpdater(it, opt, comm=comm, converter=resident_converter, device=local,
                                  overlap_comm=args.overlap_comm, graph=use_graph)

    for _ in range(max(args.warmup, 1)):
        upd.update()
    if use_graph:
        try:
            for _ in range(6):                  # (lazily shaped parameters appear at step 1, the recording at step 3)
                if upd._graphed is not None:
                    break
                upd.update()
            if upd._graphed is None:
                raise RuntimeError('no recording after the warm-up steps')
        except Exception as e:                  # never lose the measurement to the capture: eager steps
            sys.stderr.write('bench: hipGraph capture of the step failed (%s); running eager steps\n' % e)
            use_graph = upd.graph = False
    backend.synchronize()
    if opt.uninitialized_params():
        raise SystemExit('bench: parameters without storage after warm-up: %s' % opt.uninitialized_params())
    n_params = opt.n_train

    tag = _lib.PROF_RESBLOCK_GATE
    lib = _lib.load()
    lib.vqvae_prof_reset()
    if not use_graph:
        lib.vqvae_prof_enable(1 << tag)      # (a replayed recording carries no per-launch events: timed below, eagerly)
    timed_comm = hasattr(comm, 'time_comm')
    if timed_comm:
        comm.time_comm = True            # HIP events around every all-reduce, on the stream it runs on
    comm.barrier()
    backend.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        upd.update()
    backend.synchronize()
    dt_local = time.perf_counter() - t0  # this rank alone (before the closing barrier): rank skew shows here
    comm.barrier()
    dt = time.perf_counter() - t0
    lib.vqvae_prof_enable(0)
    roofline_pass = 'the timed steps'
    if use_graph:
        # per-launch time of the gate kernel: the same job, the same kernels, launched eagerly with the dispatch's own
        # start / stop events (hipExtLaunchKernelGGL) right after the timed region
        upd.graph = False
        upd.update()
        backend.synchronize()
        lib.vqvae_prof_enable(1 << tag)
        k_eager = min(args.steps, 10)
        for _ in range(k_eager):
            upd.update()
        backend.synchronize()
        lib.vqvae_prof_enable(0)
        upd.graph = True
        roofline_pass = ('%d eager steps of the same job right after the timed region (the timed steps are hipGraph replays, '
                         'which carry no per-launch events)' % k_eager)
    comm_ms, comm_calls = comm.comm_time_ms() if timed_comm else (0.0, 0)
    if timed_comm:
        comm.time_comm = False
    dt = comm.max_scalar(dt)
    # diagnostics of a multi-rank run: slowest / fastest rank's own step time, slowest rank's all-reduce time
    rank_ms = 1e3 * dt_local / args.steps
    rank_ms_max, rank_ms_min = comm.max_scalar(rank_ms), -comm.max_scalar(-rank_ms)
    comm_ms_step_max = comm.max_scalar(comm_ms / args.steps)
    comm_ms_step_min = -comm.max_scalar(-comm_ms / args.steps)

    import ctypes as C
    tot = C.c_double(0)
    cnt = C.c_int(0)
    _lib.call('vqvae_prof_read', tag, C.byref(tot), C.byref(cnt))
    losses = [float(l.data.get()) for l in upd.last_losses]
    seen = comm.ranks_seen()
    # every kernel family of the chain against ITS roofline (same job, eager, per-launch events; outside the timed region)
    ktable = None
    if rank == 0 and n == 1 and not args.bf16 and not args.no_kernel_table:
        upd.graph = False
        upd.update()
        backend.synchronize()
        ktable = kernel_roofline_table(upd, backend, cfg, B, mode, min(args.steps, 10))
        upd.graph = use_graph
    # The same job with the input leg INSIDE the step (updaters.py:8): every step consumes a minibatch that was in host
    # memory when the previous step started -- page-locked double buffer, copy stream, binning on the device
    # (inputs.StreamingInputIterator).  Timed like the main region, reported beside it, never as `value`.
    fresh_ms = None
    if index_input and not args.no_fresh_input:
        from vqvae_amd.inputs import StreamingInputIterator
        for s in range(2, 8):               # eight distinct host minibatches, cycled
            ex = synth_examples(B, cfg, seed=71 + 1000 * rank + s)
            host_batches.append((np.stack([e[0][0, :, 0] for e in ex]), np.array([e[2] for e in ex], np.int32)))
        cyc = [0]

        def source():
            cyc[0] += 1
            return host_batches[cyc[0] % len(host_batches)]
        upd._iterators['main'] = StreamingInputIterator(source, B, cfg['length'], cfg['quantize'])
        for _ in range(2):
            upd.update()
        comm.barrier()
        backend.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            upd.update()
        backend.synchronize()
        comm.barrier()
        fresh_ms = 1e3 * comm.max_scalar(time.perf_counter() - t1) / args.steps
        upd._iterators['main'] = it
    ref_ms = None
    if n == 1 and mode in PRODUCTS_PER_FP32 and not args.no_cpu_baseline:
        # the same step with the fp32 MFMA kernels, for reference (not part of the timed region above)
        upd.graph = False
        backend.set_matmul_dtype('float32')
        for _ in range(2):
            upd.update()
        backend.synchronize()
        k = min(args.steps, 5)
        t1 = time.perf_counter()
        for _ in range(k):
            upd.update()
        backend.synchronize()
        ref_ms = 1e3 * (time.perf_counter() - t1) / k
        backend.set_matmul_dtype(mode)
        upd.graph = use_graph

    if rank == 0:
        T = cfg['length']
        samples = n * B * T * args.steps
        value = samples / dt
        # algorithmic FLOPs of one launch of the gate kernel (ResidualBlock forward), SURVEY 8(d):
        # `dilconv1d` fwd = 2*B*T*Cout*Cin*K, plus -- now that the residual 1x1 conv runs inside the
        # same launch -- its 2*B*T*Cr*(Cd/2); the condition projection is not in this kernel's
        # contraction (computed once at the latent rate and lerped in the epilogue)
        flop_conv = 2.0 * B * T * cfg['dilated'] * cfg['filter_size'] * cfg['residual']
        flop = flop_conv + FUSED_RES_FLOP_PER_POS(cfg) * B * T
        x3 = mode in PRODUCTS_PER_FP32
        X3_PRODUCTS = PRODUCTS_PER_FP32.get(mode, 1)
        # the pipe that bounds the kernel: fp32 MFMA; the 16-bit MFMA pipe ('bfloat16'); or that pipe doing
        # X3_PRODUCTS products per algorithmic fp32 product ('float32x3': 6 bf16, 'float32x2': 3 fp16)
        peak = PEAK_BF16_MFMA_TFLOPS if args.bf16 else (PEAK_BF16_MFMA_TFLOPS / X3_PRODUCTS if x3 else PEAK_FP32_MFMA_TFLOPS)
        avg_ms = tot.value / max(cnt.value, 1)
        ach = flop / (avg_ms * 1e-3) / 1e12 if cnt.value else None
        key = ('c2' if args.workload == 'c2' else 'c5') + ('_bf16' if args.bf16 else {'float32x2': '', 'float32x3': '_x3', 'float32': '_fp32mfma'}[mode]) + '_B%d' % B
        traffic, tsrc = measured_traffic(key)
        out = {
            'metric': 'audio samples/sec, VQ-VAE fwd+bwd+Adam step, 16 kHz mu-law (whole job)',
            'value': value, 'unit': 'samples/s', 'n_gpus': n, 'ranks_seen_by_rccl': seen,
            'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'step_execution': ('one hipGraph launch per step (recorded after two eager steps; bit-identical to the eager step)'
                               if use_graph else 'eager launches'),
            'ms_per_step_with_input': fresh_ms,
            'with_input_is': (None if fresh_ms is None else
                              'the same %d steps with the input leg inside the step: each minibatch starts in host memory '
                              '(8 distinct host minibatches, cycled), page-locked double buffer -> copy stream -> mu-law '
                              'binning on the device (0.5 MB of waveform per step instead of the 125.8 MB one-hot)' % args.steps),
            'backward': ('loss1 + loss3 back-propagated in one sweep (the encoder walked once with g1 + g3 at z), then vq.cleargrads() and '
                         'loss2: the gradients of updaters.py:14-18, encoder to fp32 rounding, the rest bit for bit (DESIGN.md section 1)'
                         if V.updaters.MERGED_BACKWARD else "the reference's three sweeps (updaters.py:14-18)"),
            'comm_ms_per_step': comm_ms_step_max if timed_comm else None,
            'multi_rank_diagnostics': None if not timed_comm else {
                'allreduce_ms_per_step_max_over_ranks': comm_ms_step_max,
                'allreduce_ms_per_step_min_over_ranks': comm_ms_step_min,
                'allreduce_calls_per_step': comm_calls / float(args.steps),
                'allreduce_bytes_per_step': 4 * int(opt.n_train),
                'allreduce_overlapped_with_backward': bool(args.overlap_comm),
                'rank_ms_per_step_max': rank_ms_max, 'rank_ms_per_step_min': rank_ms_min,
                'note': 'all-reduce time = HIP events around ncclAllReduce on its stream (includes waiting for the '
                        'slowest rank to arrive); rank_ms = each rank\'s own wall time per step before the closing barrier',
                'cpu_affinity_rank0': affinity, 'numa_binding_rank0': numa, 'startup_self_test': comm_self_test},
            'dtype': ('bf16 operands, f32 accumulate' if args.bf16 else
                      {'float32x3': 'f32 (fp32 tensors; each fp32 product = 6 bf16 MFMA products of an exact 3-way operand split, fp32 accumulate)',
                       'float32x2': 'f32 (fp32 tensors; each fp32 product = 3 fp16 MFMA products of a power-of-two-scaled hi + lo operand split, fp32 accumulate)',
                       'float32': 'f32'}[mode]),
            'matmul': {'bfloat16': 'operands rounded to bf16 (RNE), v_mfma_f32_32x32x16_bf16, fp32 accumulate',
                       'float32': 'v_mfma_f32_32x32x2_f32',
                       'float32x3': 'fp32 operands, fp32 results: each operand split EXACTLY into three bf16 '
                                    '(h + m + l), six of the nine products on v_mfma_f32_32x32x16_bf16 with fp32 '
                                    'accumulate (the dropped three are < 2^-25 relative); error against float64 is at '
                                    'or below the fp32 MFMA path\'s (tests/test_gpu_kernels.py::'
                                    'test_float32x3_is_as_accurate_as_fp32_mfma)',
                       'float32x2': 'fp32 operands, fp32 results: each operand scaled by a power of two per tensor and split '
                                    'into two fp16 (hi + lo), three products on v_mfma_f32_32x32x16_f16 with fp32 accumulate; '
                                    'error against float64 at or below the fp32 MFMA path\'s (same test)'}[mode],
            'data': 'synthetic' + (' (x_dec as device-computed mu-law bin indices)' if index_input else
                                   (' (x_dec as the reference\'s fp32 one-hot)' if not cfg.get('use_logistic', False) else '')),
            'samples_per_sec_per_gpu': value / n,
            'config': {'workload': ('BASELINE configs[%d]: batch %d/GPU, length 7680, mu-law q=256, '
                                    'd=64 k=512, n_loop=2 n_layer=10, residual=dilated=skip=256, '
                                    'cond 64+128, EMA 0.9999, Adam lr=2e-4/N' % (1 if n == 1 else 2, B))
                       if args.workload == 'c2' else
                       ('BASELINE configs[4]: mixture-of-logistics decoder (use_logistic, input_dim=1, '
                        '10 logistics), n_loop=4 n_layer=10 (40 blocks), batch %d/GPU, length 7680' % B),
                       'global_batch': n * B, 'length': T, 'scaling': args.scaling,
                       'parallelism': 'dp%d (one process/GPU, RCCL all-reduce of the flat grad arena)' % n},
            'trainable_params': int(n_params),
            'losses_last_step': losses,
            'roofline': {'bound': 'mfma', 'kernel': {'float32x3': ROOFLINE_KERNEL_X3, 'float32x2': ROOFLINE_KERNEL_X2, 'bfloat16': ROOFLINE_KERNEL_BF16}.get(mode, ROOFLINE_KERNEL),
                         'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s',
                         'peak_is': ('dense bf16 MFMA peak' if args.bf16 else
                                     ('dense 16-bit MFMA peak (2500) / %d MFMA products per fp32 product; achieved counts '
                                      'algorithmic fp32 FLOPs' % X3_PRODUCTS if x3 else 'fp32 MFMA peak')),
                         'frac': (ach / peak) if ach else None,
                         'achieved_vs_fp32_mfma_peak': (ach / PEAK_FP32_MFMA_TFLOPS) if (ach and not args.bf16) else None,
                         'achieved_vs_measured_sustained_mfma': ((ach * X3_PRODUCTS / SUSTAINED_MFMA_TFLOPS_REAL_DATA[mode])
                                                                 if (ach and x3 and not args.bf16) else None),
                         'sustained_is': ('%.0f TFLOP/s: the %d-product MFMA stream alone on real operands, whole chip '
                                          '(tools/ubench/mfma_power.hip, f16x2_probe.hip; %.0f nominal)'
                                          % (SUSTAINED_MFMA_TFLOPS_REAL_DATA[mode], X3_PRODUCTS, PEAK_BF16_MFMA_TFLOPS))
                                         if (x3 and not args.bf16) else None,
                         'hbm': {'achieved_algorithmic': (GATE_BYTES(cfg, B, args.bf16) / (avg_ms * 1e-3) / 1e9) if cnt.value else None,
                                 'peak': 8000.0, 'unit': 'GB/s',
                                 'frac': (GATE_BYTES(cfg, B, args.bf16) / (avg_ms * 1e-3) / 1e9 / 8000.0) if cnt.value else None,
                                 'algorithmic_bytes_per_launch': GATE_BYTES(cfg, B, args.bf16),
                                 'note': 'x read once, the saved gate values + z written once (float32x2: sigmoid and z, the backward takes tanh = z / sigmoid; other fp32 modes: tanh, sigmoid, z; --bf16: all as bf16), weights, latent-rate condition slice (DESIGN.md sections 3, 3a, 3c)'},
                         'traffic': traffic, 'traffic_source': tsrc,
                         'launches': cnt.value, 'avg_launch_ms': avg_ms, 'avg_launch_ms_measured_over': roofline_pass,
                         'flop_per_launch': flop, 'flop_per_launch_dilconv1d_only': flop_conv,
                         'kernels': None if ktable is None else ktable[0],
                         'kernels_measured_over': None if ktable is None else (
                             '%d eager steps of the same job (%.3f ms per step with the per-launch events on), outside the timed region; '
                             'dispatch events per GEMM launch, stream events around a weight-gradient launch + its reduce; '
                             'mfma peak as in roofline.peak_is, hbm peak 8000 GB/s; work = algorithmic FLOPs / bytes (SURVEY 8d)'
                             % (min(args.steps, 10), ktable[1]))},
        }
        if mode == 'float32x2':
            # the run-time guard of the mode's pre-split storage: every backward sweep of this process so far --
            # warm-up, timed, with-input and eager steps -- compared each pre-split tensor's bound with its actual maximum on
            # the device; `violations` = tensors whose bound was more than 2^log2_limit above it (0 = the contract held)
            out['f32x2_contract'] = backend.f32x2_contract_violations()
        if ref_ms is not None:
            out['fp32_mfma_reference'] = {'ms_per_step': ref_ms, 'value': B * T / (ref_ms * 1e-3), 'unit': 'samples/s',
                                          'note': "the same job with --matmul float32 (v_mfma_f32_32x32x2_f32), "
                                                  "%d steps after 2 warm-up steps, outside the timed region" % min(args.steps, 5)}
        if n == 1 and not args.no_cpu_baseline and args.workload == 'c2' and not args.bf16:
            out['cpu_baseline'] = cpu_baseline(cfg)
        print(json.dumps(out))
    if n > 1 or args.force_comm:
        comm.barrier()
        comm.close()


if __name__ == '__main__':
    main()
