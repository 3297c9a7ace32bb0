"""Incremental WaveNet generation -- the queue machinery of WaveNet/modules.py:58-74, 98-110,
232-255 and the sampling loop of generate.py:101-145, on the device.

``WaveNet.initialize(n)`` / ``WaveNet.generate(x, condition)`` keep the reference's step-by-step
call surface (the host supplies the input vector and the condition column, gets the logits).
``WaveNet.generate_sequence`` is the MI355X-native form of generate.py's ``for i in range(...)``:
the whole loop -- network step, sampler, feedback of the sample, step counter -- lives in device
memory, one step's ~45 kernel launches are captured once into a hipGraph and the graph is
replayed, so the host does nothing per sample.
"""
import ctypes as C

import numpy as np

from . import _lib, backend, core
from .backend import DeviceArray

_S = backend.stream
_INT_MAX = 2 ** 31 - 1


class GenerationState(object):
    """Device state behind WaveNet.initialize(n): zero queues (modules.py:58-67, 232-244) as
    per-block rings of ``dilation`` slots, the 2-sample embed queue, the step counter."""

    def __init__(self, wavenet, n):
        if not 1 <= n <= _lib.GEN_MAX_N:
            raise ValueError('generation runs 1..%d sequences in lockstep (generate.py:42 runs 1), got %d'
                             % (_lib.GEN_MAX_N, n))
        blocks = list(wavenet.resnet.children())
        for p in wavenet.params():
            if p.data is None:
                raise RuntimeError('WaveNet.initialize: parameters are not initialised')
            backend.require_device(p.data)
        b0 = blocks[0]
        if b0.filter_size != 2:
            raise NotImplementedError('incremental generation supports filter_size == 2 (params.py:31)')
        self.n = n
        # the state holds raw device pointers of the weights: keep their buffers alive and
        # remember which parameter layout / contents they belong to
        self._weights = [p.data for p in wavenet.params()]
        self._epochs = (core.param_epoch('layout'), core.param_epoch('load'))
        self.input_dim = wavenet.embed.W.shape[1]
        self.residual = wavenet.embed.W.shape[0]
        self.dilated = b0.conv.W.shape[0]
        self.skip = b0.skip.W.shape[0]
        self.cond_dim = b0.condition_proj.W.shape[1]
        self.out_dim = wavenet.proj2.W.shape[0]
        f32 = np.float32
        self.step = backend.zeros((1,), np.int32)
        self.x_cur = backend.zeros((n, self.input_dim), f32)
        self.x_prev = backend.zeros((n, self.input_dim), f32)
        self.h0 = backend.zeros((n, self.residual), f32)
        self.h1 = backend.zeros((n, self.residual), f32)
        self.z = backend.zeros((n, self.dilated // 2), f32)
        self.skip_acc = backend.zeros((n, self.skip), f32)
        self.s1 = backend.zeros((n, self.skip), f32)
        self.logits = backend.zeros((n, self.out_dim), f32)
        self.rings = [backend.zeros((b.dilation, n, self.residual), f32) for b in blocks]
        self._blocks = (_lib.GenBlock * len(blocks))()
        for gb, b, ring in zip(self._blocks, blocks, self.rings):
            gb.conv_W, gb.conv_b = b.conv.W.data.ptr, b.conv.b.data.ptr
            gb.cond_W, gb.cond_b = b.condition_proj.W.data.ptr, b.condition_proj.b.data.ptr
            gb.res_W, gb.res_b = b.res.W.data.ptr, b.res.b.data.ptr
            gb.skip_W, gb.skip_b = b.skip.W.data.ptr, b.skip.b.data.ptr
            gb.ring, gb.dilation = ring.ptr, b.dilation
        d = _lib.GenDesc()
        d.n, d.n_blocks = n, len(blocks)
        d.input_dim, d.residual, d.dilated, d.skip = self.input_dim, self.residual, self.dilated, self.skip
        d.cond_dim, d.out_dim = self.cond_dim, self.out_dim
        d.log_scale_min = float(wavenet.log_scale_min)
        d.embed_W, d.embed_b = wavenet.embed.W.data.ptr, wavenet.embed.b.data.ptr
        d.proj1_W, d.proj1_b = wavenet.proj1.W.data.ptr, wavenet.proj1.b.data.ptr
        d.proj2_W, d.proj2_b = wavenet.proj2.W.data.ptr, wavenet.proj2.b.data.ptr
        d.blocks = self._blocks
        for name in ('step', 'x_cur', 'x_prev', 'h0', 'h1', 'z', 'skip_acc', 's1', 'logits'):
            setattr(d, name, getattr(self, name).ptr)
        d.sample_mode = _lib.GEN_NONE
        d.max_steps = _INT_MAX
        self.desc = d

    # ---- reference-shaped single step: WaveNet.generate(x, condition) (modules.py:246-255) ----
    def step_logits(self, x, condition):
        n = self.n
        if x.size != n * self.input_dim:
            raise ValueError('generate: x must hold (n, input_dim, 1, 1) = (%d, %d, 1, 1) values, got %s'
                             % (n, self.input_dim, x.shape))
        if condition.size != n * self.cond_dim:
            raise ValueError('generate: condition must hold (n, condition_dim, 1, 1) = (%d, %d, 1, 1) '
                             'values, got %s' % (n, self.cond_dim, condition.shape))
        self.x_cur.copy_from(x)
        d = self.desc
        d.sample_mode = _lib.GEN_NONE
        d.cond, d.cond_bstride, d.cond_cstride, d.cond_follows_step = condition.ptr, self.cond_dim, 1, 0
        d.uniforms = d.forced_next = d.out = d.logits_out = None
        d.max_steps = _INT_MAX
        _lib.call('vqvae_wavenet_gen_step', C.byref(d), _S())
        self._keep = (x, condition)
        return self.logits.copy().reshape(n, self.out_dim, 1, 1)

    # ---- the whole loop of generate.py:105-145 on the device ----------------------------------
    def run(self, condition, uniforms, mode, n_steps=None, forced=None, return_logits=False,
            graph_steps=8, persistent=False, chunk=4096):
        n = self.n
        if condition.ndim == 4:
            condition = condition.reshape(condition.shape[:3])
        if condition.shape[:2] != (n, self.cond_dim):
            raise ValueError('generate_sequence: condition must be (n, condition_dim, T[, 1]) = (%d, %d, T), got %s'
                             % (n, self.cond_dim, condition.shape))
        T = condition.shape[2]
        steps = T - 1 if n_steps is None else int(n_steps)          # generate.py:105
        if not 0 <= steps <= T:
            raise ValueError('generate_sequence: n_steps must be in [0, %d]' % T)
        softmax = mode == _lib.GEN_SOFTMAX
        n_uniform = 1 if softmax else self.out_dim // 3
        u = np.ascontiguousarray(uniforms, dtype=np.float64).reshape(-1)
        if u.size < steps * n * n_uniform:
            raise ValueError('generate_sequence: need %d uniform doubles (steps, n, %d), got %d'
                             % (steps * n * n_uniform, n_uniform, u.size))
        u_dev = backend.to_device(u[:max(1, steps * n * n_uniform)])
        out = backend.zeros((n, T), np.int32 if softmax else np.float32)     # generate.py:103
        forced_dev = None
        if forced is not None:
            forced_dev = backend.to_device(np.ascontiguousarray(
                forced, dtype=np.int32 if softmax else np.float32).reshape(-1)[:steps * n])
        logits_out = DeviceArray((max(steps, 1), n, self.out_dim), np.float32) if return_logits else None
        d = self.desc
        d.sample_mode = mode
        d.cond, d.cond_bstride, d.cond_cstride, d.cond_follows_step = condition.ptr, self.cond_dim * T, T, 1
        d.uniforms, d.n_uniform = u_dev.ptr, n_uniform
        d.forced_next = None if forced_dev is None else forced_dev.ptr
        d.out, d.out_bstride = out.ptr, T
        d.logits_out = None if logits_out is None else logits_out.ptr
        if int(self.step.get()[0]) != 0:
            raise RuntimeError('generate_sequence starts from fresh queues: call initialize(n) first')
        d.max_steps = steps
        nbytes = 0
        if steps and persistent:
            # shapes outside the persistent kernel's limits (> 256 channels per vector) run on the
            # per-step kernels below -- same results, more launches
            nbytes = _lib.load().vqvae_wavenet_gen_run_workspace_bytes(C.byref(d))
        if steps and persistent and nbytes:
            # one persistent launch per `chunk` steps; queues + mailboxes live in its workspace
            ws = DeviceArray((nbytes // 4 + 1,), np.int32)
            for t0 in range(0, steps, chunk):
                _lib.call('vqvae_wavenet_gen_run', C.byref(d), t0, min(chunk, steps - t0), ws.ptr,
                          ws.nbytes, _S())
            backend.synchronize()
            status = int(ws.flat_view(0, 1).get()[0])
            self._last_ws = ws
            if status:
                raise RuntimeError('persistent generation kernel gave up waiting (status %d): are all '
                                   'its workgroups resident?' % status)
            self.step.set(np.array([steps], np.int32))
        elif steps and not graph_steps:                # eager launches (profilers, debugging)
            for _ in range(steps):
                _lib.call('vqvae_wavenet_gen_step', C.byref(d), _S())
            backend.synchronize()
        elif steps:
            per = max(1, min(int(graph_steps), steps))
            graph = C.c_void_p()
            _lib.call('vqvae_graph_capture_begin', _S())
            try:
                for _ in range(per):
                    _lib.call('vqvae_wavenet_gen_step', C.byref(d), _S())
            finally:
                _lib.call('vqvae_graph_capture_end', _S(), C.byref(graph))
            try:
                for _ in range((steps + per - 1) // per):        # surplus steps are device no-ops
                    _lib.call('vqvae_graph_launch', graph, _S())
                backend.synchronize()
            finally:
                _lib.call('vqvae_graph_destroy', graph)
        d.max_steps = _INT_MAX
        return (out, logits_out) if return_logits else out


def run_many(wavenet, condition, uniforms, mode, n_steps=None, group=_lib.GEN_MAX_N, max_streams=8,
             chunk=4096):
    """Batched serving form of the loop: N sequences = ceil(N / group) independent persistent
    launches of <= ``group`` lockstep sequences each, spread over ``max_streams`` HIP streams so
    they run concurrently (a launch occupies 128 two-wave workgroups: a fraction of the GPU).
    condition (N, condition_dim, T) on the device, uniforms (T, N[, nr_mix]) on the host.
    Returns the host array (N, T) of sampled bins / values."""
    if condition.ndim == 4:
        condition = condition.reshape(condition.shape[:3])
    N, Cc, T = condition.shape
    steps = T - 1 if n_steps is None else int(n_steps)
    softmax = mode == _lib.GEN_SOFTMAX
    u = np.asarray(uniforms, np.float64)
    u = u.reshape(u.shape[0], N, -1)
    lib = _lib.load()
    jobs = []
    for g0 in range(0, N, group):
        n = min(group, N - g0)
        st = GenerationState(wavenet, n)
        d = st.desc
        n_uniform = 1 if softmax else st.out_dim // 3
        if u.shape[0] < steps or u.shape[2] < n_uniform:
            raise ValueError('generate_batch: need (steps, N, %d) uniform doubles, got %s' % (n_uniform, u.shape))
        u_dev = backend.to_device(np.ascontiguousarray(u[:max(steps, 1), g0:g0 + n, :n_uniform]).reshape(-1))
        out = backend.zeros((n, T), np.int32 if softmax else np.float32)
        cview = condition.flat_view(g0 * Cc * T, n * Cc * T, (n, Cc, T))
        d.sample_mode = mode
        d.cond, d.cond_bstride, d.cond_cstride, d.cond_follows_step = cview.ptr, Cc * T, T, 1
        d.uniforms, d.n_uniform = u_dev.ptr, n_uniform
        d.forced_next = d.logits_out = None
        d.out, d.out_bstride = out.ptr, T
        d.max_steps = steps
        nbytes = lib.vqvae_wavenet_gen_run_workspace_bytes(C.byref(d))
        if not nbytes:
            raise ValueError('persistent generation: ' + lib.vqvae_last_error_string().decode())
        ws = DeviceArray((nbytes // 4 + 1,), np.int32)
        jobs.append((st, d, out, ws, u_dev, cview))
    backend.synchronize()                  # buffers above were prepared on the main stream
    # every workgroup of a launch must be resident at once (its waves wait for each other): a launch
    # holds 128 two-wave workgroups and a CU takes ~12 such waves (VGPR-limited) -- stay well inside
    resident = backend.device_info()['n_cu'] * 12 // (128 * 2)
    max_streams = max(1, min(max_streams, resident * 2 // 3))
    for t0 in range(0, steps, chunk):
        for i, (st, d, out, ws, _, _) in enumerate(jobs):
            _lib.call('vqvae_wavenet_gen_run', C.byref(d), t0, min(chunk, steps - t0), ws.ptr, ws.nbytes,
                      backend.pool_stream(i % max_streams))
    backend.synchronize()
    res = []
    for st, d, out, ws, _, _ in jobs:
        status = int(ws.flat_view(0, 1).get()[0])
        if status:
            raise RuntimeError('persistent generation kernel gave up waiting (status %d)' % status)
        res.append(out.get())
    return np.concatenate(res, axis=0)
