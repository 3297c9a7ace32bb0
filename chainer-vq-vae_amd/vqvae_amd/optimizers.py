"""chainer.optimizers.Adam for this path (train.py:101-102), over flat arenas.

``setup(model)`` moves every parameter into one contiguous fp32 arena (in
namedparams order; EMA shadow copies last) and gives each trainable parameter a
slot in one contiguous gradient arena.  ``update()`` is then a single fused Adam
kernel, ``cleargrads()`` a single memset, and the data-parallel exchange a single
RCCL all-reduce of the gradient arena (updaters.py:71-77 replaced).

Lazily shaped parameters.  The reference builds ``ConditionEmbed`` with
``DilatedConvolution2D(None, ...)`` (net.py:34-43) and calls ``optimizer.setup``
before the first forward (train.py:76-102), so five conv weights do not exist at
setup time; Chainer creates them at the first call and trains them like any other
parameter.  Here ``adopt_new_params()`` -- run by the updaters between backward and
the gradient exchange, and by ``update()`` -- notices parameters created since the
last layout (core.param_epoch('init')), rebuilds the arenas with them in
namedparams order and carries over every value, gradient and Adam moment, so the
first step already updates (and all-reduces) them.

Parameters whose gradient is None at update time (the EMA copies; the last
block's ``res`` conv, modules.py:89-96) are skipped by Chainer's update rule;
here their gradient slots are zero and their Adam moments stay exactly zero, so
the update is the identity on them -- the same result.
"""
import numpy as np

from . import _lib, backend, core
from .backend import DeviceArray


class Adam(object):
    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        self.alpha = alpha
        self.beta1 = beta1
        self.beta2 = beta2
        self.eps = eps
        self.t = 0
        self.target = None
        self._layout = []
        self._epoch = -1

    @property
    def lr(self):
        fix1 = 1.0 - self.beta1 ** self.t
        fix2 = 1.0 - self.beta2 ** self.t
        return self.alpha * np.sqrt(fix2) / fix1

    def setup(self, link):
        self.target = link
        self._layout = []
        self.params = self.grads = self.m = self.v = None
        self._build()
        return self

    # ------------------------------------------------------------------ #
    def _build(self):
        """(Re)builds the flat arenas over every parameter that has storage now."""
        self._epoch = core.param_epoch('init')
        named = [(n, p) for n, p in self.target.namedparams() if p.data is not None]
        for n, p in named:
            if not isinstance(p.data, DeviceArray):
                raise ValueError('optimizer.setup: parameter %s is on the host; call '
                                 'model.to_gpu() first (there is no CPU update path)' % n)
            # (checked BEFORE any p.data is re-pointed at an arena view: backend.copy_many refuses other dtypes, and a refusal
            #  half-way through the adoption would leave parameters pointing at views nothing was copied into)
            g = p.grad.data if isinstance(p.grad, core.Variable) else p.grad
            if p.data.dtype != np.float32 or (isinstance(g, DeviceArray) and g.dtype != np.float32):
                raise ValueError('optimizer.setup: parameter %s is %s; the arenas are float32' % (n, p.data.dtype))
        train = [(n, p) for n, p in named if not p._shadow]
        shadow = [(n, p) for n, p in named if p._shadow]
        old = {n: (off, size) for n, off, size in self._layout}
        old_m, old_v, old_n_train = self.m, self.v, getattr(self, 'n_train', 0)
        n_train = sum(p.size for _, p in train)
        n_all = n_train + sum(p.size for _, p in shadow)
        params = backend.empty((n_all,), np.float32)
        grads = backend.zeros((n_train,), np.float32)
        m = backend.zeros((n_train,), np.float32)
        v = backend.zeros((n_train,), np.float32)
        layout = []
        off = 0
        copies = []                           # (dst, src): issued together below (backend.copy_many)
        for n, p in train + shadow:
            view = params.flat_view(off, p.size, p.data.shape)
            copies.append((view, p.data))     # from the old arena or from the parameter's own buffer
            p.data = view
            if not p._shadow:
                slot = grads.flat_view(off, p.size, view.shape)
                if p.grad is not None:        # gradient accumulated before adoption / re-layout
                    copies.append((slot, p.grad.reshape(view.shape) if isinstance(p.grad, DeviceArray)
                                   else p.grad.data.reshape(view.shape)))
                    p.grad = slot
                p._grad_slot = slot
                if n in old and old[n][0] + old[n][1] <= old_n_train:
                    o, sz = old[n]
                    copies.append((m.flat_view(off, sz), old_m.flat_view(o, sz)))
                    copies.append((v.flat_view(off, sz), old_v.flat_view(o, sz)))
            p._owner_step = self._step_count        # lets caches notice parameter updates
            layout.append((n, off, p.size))
            off += p.size
        backend.copy_many(copies)
        self.params, self.grads, self.m, self.v = params, grads, m, v
        self._layout = layout
        self.n_train = n_train
        core.bump_param_epoch('layout')

    def adopt_new_params(self):
        """Re-lays the arenas when parameters were created since the last layout (lazily
        shaped links at their first forward).  Cheap when nothing changed: one integer compare."""
        if self._epoch == core.param_epoch('init'):
            return False
        known = {n for n, _, _ in self._layout}
        fresh = [n for n, p in self.target.namedparams() if p.data is not None and n not in known]
        if not fresh:
            self._epoch = core.param_epoch('init')
            return False
        for n, p in self.target.namedparams():
            if n in fresh:
                p.to_gpu()
        self._build()
        return True

    def uninitialized_params(self):
        return [n for n, p in self.target.namedparams() if p.data is None]

    def _step_count(self):
        return self.t

    def update(self):
        """One Adam step on every trainable parameter (chainer Adam update rule)."""
        if self._recording:
            # inside a hipGraph capture (updaters.GraphedStep): the step size comes from the device-side schedule,
            # nothing host-dependent goes into the recorded launch
            _lib.call('vqvae_adam_step_dev', self.params.ptr, self.grads.ptr, self.m.ptr, self.v.ptr, self.n_train,
                      self._lr_table.ptr, self._step_dev.ptr, float(self.beta1), float(self.beta2), float(self.eps),
                      backend.stream())
            return
        self.adopt_new_params()
        self.t += 1
        _lib.call('vqvae_adam_step', self.params.ptr, self.grads.ptr, self.m.ptr, self.v.ptr,
                  self.n_train, float(self.lr), float(self.beta1), float(self.beta2),
                  float(self.eps), backend.stream())

    # ---- device-side schedule for a captured step ------------------------------------------------
    _recording = False
    _lr_table = None
    SCHEDULE_HORIZON = 4096

    def sync_schedule(self):
        """Makes lr_table[*step] the step size of update number ``self.t + 1`` (what the next replayed step must
        use): a no-op while the device counter is where the host's ``t`` says; otherwise -- first use, the horizon
        reached, eager steps in between -- the table is refilled from ``t`` (one small synchronous upload)."""
        hyper = (float(self.alpha), float(self.beta1), float(self.beta2))
        if (self._lr_table is not None and self._sched_next == self.t + 1
                and self.t + 1 - self._sched_base <= self.SCHEDULE_HORIZON
                and self._sched_hyper == hyper):        # (a changed alpha / beta -- a decay extension, a manual edit -- refills the table: ADVICE r4)
            return
        base, t_save = self.t, self.t
        tbl = np.empty(self.SCHEDULE_HORIZON, np.float32)
        for i in range(self.SCHEDULE_HORIZON):
            self.t = base + 1 + i
            tbl[i] = np.float32(self.lr)             # the same double -> float32 rounding as vqvae_adam_step's cast
        self.t = t_save
        if self._lr_table is None:
            self._lr_table = backend.empty((self.SCHEDULE_HORIZON,), np.float32)
            self._step_dev = backend.empty((1,), np.int32)
        self._lr_table.set(tbl)
        self._step_dev.set(np.zeros(1, np.int32))
        self._sched_base, self._sched_next = base, base + 1
        self._sched_hyper = hyper

    def capture_key(self):
        """What a recorded step bakes into its Adam launch as kernel arguments: a change re-records (updaters._step_key)."""
        return (float(self.beta1), float(self.beta2), float(self.eps))

    def replayed(self):
        """Host-side bookkeeping of one replayed (or just captured and launched) step."""
        self.t += 1
        self._sched_next = self.t + 1

    def layout(self):
        """[(name, offset, size)] of the flat arena (trainable first)."""
        return list(self._layout)
