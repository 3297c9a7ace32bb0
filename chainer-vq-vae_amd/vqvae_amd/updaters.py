"""VQVAE_StandardUpdater / VQVAE_ParallelUpdater -- mirror the reference's
updaters.py (5-19, 22-77) on minimal StandardUpdater / ParallelUpdater bases
that provide the attributes the reference's ``update_core`` bodies use
(``_iterators``, ``converter``, ``_optimizers``, ``loss_func``, ``device``,
``get_optimizer``, ``get_iterator``)."""
import os
import numpy as np

from . import backend, core
from .comm import SingleCommunicator, shard as strided_shard


def concat_examples(batch, device=None):
    """chainer.dataset.concat_examples: list of example tuples -> tuple of stacked
    arrays, moved to the device when ``device >= 0`` (updaters.py:8, 37-38)."""
    first = batch[0]
    n = len(first)
    out = []
    for i in range(n):
        arr = np.stack([np.asarray(ex[i]) for ex in batch])
        if device is not None and device >= 0:
            arr = backend.to_device(arr)
        out.append(arr)
    return tuple(out)


class GraphedStep(object):
    """One training step (updaters.py:13-19: forward, three-loss backward, optimizer update) recorded ONCE into a
    hipGraph and replayed: ~330 kernel launches, their Python-level autograd bookkeeping and ~4 ms of host time per
    step become one hipGraphLaunch.  What makes the recording replayable:
      * the minibatch is copied into staging arrays at fixed addresses before every replay (the recording reads those);
      * every buffer the recording allocates lives in a backend.Arena that nothing else can take from, so the
        addresses baked into the recorded launches stay valid and are reused only by the recording itself;
      * Adam's step size is read from a device-side schedule (optimizers.Adam.sync_schedule, vqvae_adam_step_dev);
      * the losses (and whatever else the step left behind) are arrays of the arena: after a replay they hold the
        new values.
    The recording is keyed by the input signature, the parameter layout and the matmul mode: anything else (another
    batch size, lazily created parameters, a re-laid arena) falls back to an eager step and records again."""

    def __init__(self, key, stage, arena, graph, losses, keep):
        self.key, self.stage, self.arena, self.graph, self.losses, self.keep = key, stage, arena, graph, losses, keep

    def load(self, in_arrays):
        # raw copies: the staging arrays never carry an absolute maximum (the recording scans them itself, so every
        # replay derives its float32x2 scales from the data it was handed -- ADVICE r4)
        from . import _lib
        for dst, src in zip(self.stage, in_arrays):
            if src.nbytes != dst.nbytes:
                raise ValueError('size mismatch in GraphedStep.load')
            _lib.call('vqvae_memcpy_d2d', dst.ptr, src.ptr, dst.nbytes, backend.stream())
            dst.amax = None

    def launch(self):
        from . import _lib
        _lib.call('vqvae_graph_launch', self.graph, backend.stream())

    def __del__(self):            # a dropped recording hands its graph and its arena's blocks back
        try:
            self.release()
        except Exception:
            pass

    def release(self):
        from . import _lib
        if self.graph is not None:
            backend.synchronize()
            _lib.call('vqvae_graph_destroy', self.graph)
            self.graph = None
            self.losses = self.keep = None
            self.arena.release()


def _like(a):
    """A fresh device array of ``a``'s shape, dtype and class (IndexInput keeps its quantize).  NOT its absolute
    maximum: a staging array that inherited one would make the recording skip its scan and bake the pointer to THAT
    batch's maximum into the graph -- every replay would then scale new data by a stale maximum."""
    out = a.__class__.__new__(a.__class__)
    backend.DeviceArray.__init__(out, a.shape, a.dtype)
    for slot in getattr(a.__class__, '__slots__', ()):
        if slot not in ('ptr', 'shape', 'dtype', '_block', 'amax', '__weakref__') and hasattr(a, slot):
            setattr(out, slot, getattr(a, slot))
    return out


class StandardUpdater(object):
    def __init__(self, iterator, optimizer, converter=concat_examples, device=0, loss_func=None, graph=False):
        self._iterators = iterator if isinstance(iterator, dict) else {'main': iterator}
        self._optimizers = optimizer if isinstance(optimizer, dict) else {'main': optimizer}
        self.converter = converter
        self.device = device
        self.loss_func = loss_func
        self.iteration = 0
        # graph=True: record the step into a hipGraph after ``graph_warmup`` eager steps and replay it (GraphedStep)
        self.graph = bool(graph)
        self.graph_warmup = 2
        self._graphed = None
        self._eager_steps = 0

    # ---- captured step ---------------------------------------------------------------------------
    def _step_key(self, in_arrays, optimizer):
        from . import _lib
        sig = tuple((type(a).__name__, tuple(a.shape), str(a.dtype)) for a in in_arrays)
        return (sig, core.param_epoch('layout'), core.param_epoch('init'), _lib.load().vqvae_get_matmul_dtype(),
                backend.overlap_enabled(), id(optimizer),
                optimizer.capture_key() if hasattr(optimizer, 'capture_key') else None)

    def _run_step(self, in_arrays, body):
        """``body(in_arrays)`` = forward + backward (+ exchange) + optimizer.update, eagerly or through the recording."""
        optimizer = self._optimizers['main']
        if not self.graph:
            return body(in_arrays)
        in_arrays = tuple(in_arrays)
        if any(not isinstance(a, backend.DeviceArray) for a in in_arrays):
            raise TypeError('graph=True needs device-resident input arrays')
        key = self._step_key(in_arrays, optimizer)
        g = self._graphed
        if g is not None and g.key == key:
            optimizer.sync_schedule()
            g.load(in_arrays)
            g.launch()
            optimizer.replayed()
            self.last_losses = g.losses
            return
        if g is not None:
            g.release()
            self._graphed = None
        if self._eager_steps < self.graph_warmup or optimizer.uninitialized_params():
            self._eager_steps += 1
            body(in_arrays)
            if self._step_key(in_arrays, optimizer) != key:      # parameters appeared / moved during this step
                self._eager_steps = min(self._eager_steps, self.graph_warmup - 1)
            return
        self._graphed = self._record(in_arrays, key, body, optimizer)

    def _record(self, in_arrays, key, body, optimizer):
        from . import _lib
        import ctypes as C
        stage = [_like(a) for a in in_arrays]
        optimizer.sync_schedule()
        arena = backend.Arena()
        st = backend.stream()
        graph = C.c_void_p()
        backend.synchronize()
        backend.arena_begin(arena)
        optimizer._recording = True
        try:
            _lib.call('vqvae_graph_capture_begin_relaxed', st)
            try:
                body(stage)
            finally:
                _lib.call('vqvae_graph_capture_end', st, C.byref(graph))
        finally:
            optimizer._recording = False
            backend.arena_end()
        # nothing has executed yet: this step runs as the first launch of the recording
        keep = [dict(backend._state['ws'])]          # the scratch buffers the recording reads and writes
        g = GraphedStep(key, stage, arena, graph, self.last_losses, keep)
        g.load(in_arrays)
        g.launch()
        optimizer.replayed()
        return g

    def get_optimizer(self, name):
        return self._optimizers[name]

    def get_iterator(self, name):
        return self._iterators[name]

    def update(self):
        self.update_core()
        self.iteration += 1


# The reference back-propagates its three losses in three sweeps (updaters.py:14-18).  The reconstruction loss and
# the commitment loss both reach the encoder through ONE variable, its output z (loss1 through the quantiser's
# straight-through identity, loss3 = beta * mean((z - e.data)^2) directly), and back-propagation is linear: one sweep from
# loss1 + loss3 hands the encoder g1 + g3 and walks it ONCE -- the same gradients (summation order aside: the two
# contributions are added at z instead of in every encoder parameter), minus one encoder backward per step (6 backward-data
# GEMMs, 6 weight gradients and their ~50 small launches: ~0.5 ms of 17 at configs[1]).  The codebook still receives the
# codebook loss only.  merged=False (or updaters.MERGED_BACKWARD = False) runs the reference's three sweeps.
MERGED_BACKWARD = True


def _merged_sweep(losses, merged=None):
    """Whether loss1 and loss3 are back-propagated in one sweep: the switch, and both must be this package's Variables
    (anything else that merely has .backward() -- a test double, a foreign autograd -- gets the reference's three sweeps)."""
    want = MERGED_BACKWARD if merged is None else merged
    return bool(want) and isinstance(losses[0], core.Variable) and isinstance(losses[2], core.Variable)


class _codebook_share_discarded(object):
    """The reconstruction loss's sweep also reaches the codebook (through e = vq(z)), and the very next statement of the
    reference throws that gradient away (``model.vq.cleargrads()``, updaters.py:16).  While this context is open the
    codebook parameters do not ask for a gradient, so the sweep does not compute what is discarded (the fp64 scatter of
    utils.py:227-228: ~70 us per step at the configs); the result of the three-loss routine is unchanged."""

    def __init__(self, model):
        vq = getattr(model, 'vq', None)
        self.params = [p for p in vq.params()] if isinstance(vq, core.Link) else []

    def __enter__(self):
        self.old = [p.requires_grad for p in self.params]
        for p in self.params:
            p.requires_grad = False

    def __exit__(self, *exc):
        for p, r in zip(self.params, self.old):
            p.requires_grad = r
        return False


def three_loss_backward(model, losses, merged=None):
    """The gradient routing of the reference's updaters (updaters.py:14-18, 58-69):
    clear everything, back-propagate the reconstruction loss, throw away what it put on the
    codebook, then add the codebook loss (-> vq.W only) and the commitment loss (-> encoder)."""
    loss1, loss2, loss3 = losses
    model.cleargrads()
    with backend.deferred_join():       # (weight gradients deferred to the side stream are joined once, behind the last sweep)
        with _codebook_share_discarded(model):
            if _merged_sweep(losses, merged):
                (loss1 + loss3).backward()  # decoder, condition embed, encoder (g1 + g3 at z)
            else:
                loss1.backward()
        model.vq.cleargrads()
        loss2.backward()
        if not _merged_sweep(losses, merged):
            loss3.backward()


class VQVAE_StandardUpdater(StandardUpdater):
    """Single-device step (updaters.py:5-19)."""

    def update_core(self):
        optimizer = self._optimizers['main']
        in_arrays = self.converter(self._iterators['main'].next(), self.device)
        loss_func = self.loss_func or optimizer.target

        def body(arrays):
            self.last_losses = loss_func(*arrays)
            three_loss_backward(optimizer.target, self.last_losses)
            optimizer.update()
        self._run_step(in_arrays, body)


class VQVAE_ParallelUpdater(StandardUpdater):
    """Data-parallel step.  The reference drives all GPUs from one process and
    reduces by device-to-device adds + a parameter broadcast (updaters.py:23-77).
    MI355X-native form: one process per GPU; this rank takes ``batch[rank::n]``
    (updaters.py:37-38), runs the same three-loss backward, then ONE RCCL
    all-reduce(sum) of the flat gradient arena replaces addgrads (sum, not mean,
    updaters.py:71-72) and -- because every rank applies the identical Adam
    update to identical parameters -- copyparams (updaters.py:76-77).  The
    optimizer's alpha must already be lr/n (train.py:101)."""

    def __init__(self, iterator, optimizer, comm=None, converter=concat_examples, device=0,
                 loss_func=None, overlap_comm=None, graph=False):
        super(VQVAE_ParallelUpdater, self).__init__(iterator, optimizer, converter, device,
                                                    loss_func, graph=graph)
        self.comm = comm or SingleCommunicator()
        # exchange the gradients that are final after the reconstruction loss's backward (decoder, condition embed: 95 % of
        # the arena) on the side stream while the codebook / commitment losses still back-propagate into vq.W and the
        # encoder; fixed bucket order, same sums.  Default (None): on whenever there is someone to exchange with
        # (comm.size > 1); False keeps the whole arena in one all-reduce on the main stream.
        self.overlap_comm = (self.comm.size > 1) if overlap_comm is None else bool(overlap_comm)
        self._overlap_defaulted = overlap_comm is None      # (a defaulted request gives way when the model has no .encoder / .vq to bucket by)
        # A recorded step (graph=True) replays whatever the communicator ENQUEUED while it was recorded: a communicator that
        # does host-side work per exchange would run it once, at capture, and never again -- replicas would diverge silently.
        # Only communicators that declare themselves capture-safe (device-side collectives only) may be recorded with n > 1.
        if self.graph and self.comm.size > 1 and not getattr(self.comm, 'capture_safe', False):
            import warnings
            warnings.warn('VQVAE_ParallelUpdater(graph=True): %s does not declare capture_safe; running eager steps'
                          % type(self.comm).__name__)
            self.graph = False
        self._buckets = None

    def _grad_buckets(self, optimizer, model, merged=False):
        """(early, late): contiguous [offset, size) runs of the gradient arena.  `late` = what
        loss2 / loss3 still write after loss1's backward -- the parameters of ``model.vq`` and
        ``model.encoder`` (updaters.py:16-18), found BY IDENTITY in the optimizer's layout, whatever the
        links are called and wherever the VAE sits inside ``optimizer.target`` -- `early` = the rest.
        A model without those two links, or one whose late set comes out empty, has no early bucket at
        all (raises): exchanging a gradient before its last writer has run would be silently wrong."""
        key = (core.param_epoch('layout'), bool(merged))
        if self._buckets is None or self._buckets[0] != key:
            enc, vq = getattr(model, 'encoder', None), getattr(model, 'vq', None)
            if enc is None or vq is None:
                raise RuntimeError('overlap_comm needs a model with .encoder and .vq links (the parameters the '
                                   'codebook / commitment losses still write after the reconstruction loss); got %s'
                                   % type(model).__name__)
            # (merged sweeps, see three_loss_backward: the first sweep already finishes the encoder; only the codebook
            # is still written by the second)
            late_links = (vq,) if merged else (enc, vq)
            late_ids = {id(p) for link in late_links for p in link.params() if p.data is not None}
            by_name = {n: p for n, p in optimizer.target.namedparams()}
            early, late = [], []
            n_late = 0
            for name, off, size in optimizer.layout():
                if off + size > optimizer.n_train:
                    continue                       # EMA shadows: no gradient
                is_late = id(by_name[name]) in late_ids
                n_late += is_late
                dst = late if is_late else early
                if dst and dst[-1][0] + dst[-1][1] == off:
                    dst[-1] = (dst[-1][0], dst[-1][1] + size)
                else:
                    dst.append((off, size))
            if n_late != len(late_ids):
                raise RuntimeError('overlap_comm: %d of the %d encoder / vq parameters are in the optimizer\'s arena -- '
                                   'is optimizer.target the model whose losses are back-propagated?' % (n_late, len(late_ids)))
            self._buckets = (key, early, late)
        return self._buckets[1], self._buckets[2]

    def update_core(self):
        optimizer = self.get_optimizer('main')
        model = optimizer.target
        n = self.comm.size
        it = self.get_iterator('main')
        if getattr(it, 'yields_rank_shard', False):
            shard = it.next()        # the iterator already produces batch[rank::n] only (no rank builds the global batch)
        else:
            shard = strided_shard(it.next(), self.comm.rank, n)   # batch[rank::n], as the reference (updaters.py:37-38)
        in_arrays = self.converter(shard, self.device)
        self._run_step(in_arrays, lambda arrays: self._step_body(arrays, optimizer, model, n))

    def _step_body(self, in_arrays, optimizer, model, n):
        with core.force_backprop_mode():
            self.last_losses = (self.loss_func or model)(*in_arrays)
        exchange = n > 1 or getattr(self.comm, 'always_reduce', False)
        if self.overlap_comm and exchange:
            lm = self.loss_func_model(model)
            if not self._overlap_defaulted or (hasattr(lm, 'encoder') and hasattr(lm, 'vq')):
                return self._update_overlapped(optimizer, model)
        three_loss_backward(model, self.last_losses)

        # parameters created during this forward (lazily shaped links, net.py:34-43) join the
        # flat arenas BEFORE the exchange, so their gradients are summed like everyone else's
        adopt = getattr(optimizer, 'adopt_new_params', None)
        if not getattr(optimizer, '_recording', False) and adopt is not None and adopt() and n > 1:   # (any optimizer with update() / grads serves here: the recording flag is the device Adam's)
            self._check_replicas(optimizer)
        if exchange:
            self.comm.allreduce_grad(optimizer.grads)       # sum over ranks, in place
        optimizer.update()

    def loss_func_model(self, model):
        """The link whose .encoder / .vq the three losses route into: ``loss_func`` when it is such a link (a custom
        loss_func wrapping the VAE must expose them itself), else optimizer.target."""
        lf = self.loss_func
        return lf if (lf is not None and hasattr(lf, 'encoder') and hasattr(lf, 'vq')) else model

    def _check_replicas(self, optimizer):
        """Parameters adopted after setup (lazily shaped links) were initialised by each rank's own
        initializer stream, and nothing broadcasts parameters here (the reference's copyparams,
        updaters.py:76-77, is replaced by "identical update on identical replicas"): verify the
        premise when it can break -- the sum of the parameter arena must be the same on every rank."""
        from . import _lib
        tot = backend.zeros((1,), np.float32)
        ws = backend.workspace(4096 * 4)
        _lib.call('vqvae_sum', optimizer.params.ptr, optimizer.n_train, 1.0, tot.ptr, ws.ptr, ws.nbytes,
                  backend.stream())
        v = float(tot.get()[0])
        hi, lo = self.comm.max_scalar(v), -self.comm.max_scalar(-v)
        if hi != lo:
            raise RuntimeError('data-parallel replicas diverged: parameter checksum %r..%r across ranks after '
                               'lazily shaped parameters were created -- seed the initializers identically on '
                               'every rank (core.seed_initializers) or load the same snapshot' % (lo, hi))

    def _update_overlapped(self, optimizer, model):
        """three_loss_backward with the exchange of the early bucket on the side stream."""
        loss1, loss2, loss3 = self.last_losses
        merged = _merged_sweep(self.last_losses)
        model.cleargrads()
        with _codebook_share_discarded(model):
            if merged:
                (loss1 + loss3).backward()
            else:
                loss1.backward()
        model.vq.cleargrads()
        adopt = getattr(optimizer, 'adopt_new_params', None)
        if adopt is not None and adopt() and self.comm.size > 1:
            self._check_replicas(optimizer)
        early, late = self._grad_buckets(optimizer, self.loss_func_model(model), merged)
        main, side = backend.stream(), backend.side_stream()
        backend.wait_event(side, backend.Event().record(main))      # loss1's gradients are complete
        for off, size in early:
            self.comm.allreduce_grad(optimizer.grads.flat_view(off, size), stream=side)
        loss2.backward()
        if not merged:
            loss3.backward()
        for off, size in late:
            self.comm.allreduce_grad(optimizer.grads.flat_view(off, size))
        backend.wait_event(main, backend.Event().record(side))      # join before the optimizer reads the arena
        optimizer.update()
