"""chainer.optimizers.Adam for this path (train.py:101-102), over flat arenas.

``setup(model)`` moves every parameter into one contiguous fp32 arena (in
namedparams order; EMA shadow copies last) and gives each trainable parameter a
slot in one contiguous gradient arena.  ``update()`` is then a single fused Adam
kernel, ``cleargrads()`` a single memset, and the data-parallel exchange a single
RCCL all-reduce of the gradient arena (updaters.py:71-77 replaced).

Parameters whose gradient is None at update time (the EMA copies; the last
block's ``res`` conv, modules.py:89-96) are skipped by Chainer's update rule;
here their gradient slots are zero and their Adam moments stay exactly zero, so
the update is the identity on them -- the same result.
"""
import numpy as np

from . import _lib, backend
from .backend import DeviceArray


class Adam(object):
    def __init__(self, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        self.alpha = alpha
        self.beta1 = beta1
        self.beta2 = beta2
        self.eps = eps
        self.t = 0
        self.target = None

    @property
    def lr(self):
        fix1 = 1.0 - self.beta1 ** self.t
        fix2 = 1.0 - self.beta2 ** self.t
        return self.alpha * np.sqrt(fix2) / fix1

    def setup(self, link):
        self.target = link
        named = [(n, p) for n, p in link.namedparams() if p.data is not None]
        for n, p in named:
            if not isinstance(p.data, DeviceArray):
                raise ValueError('optimizer.setup: parameter %s is on the host; call '
                                 'model.to_gpu() first (there is no CPU update path)' % n)
        train = [(n, p) for n, p in named if not p._shadow]
        shadow = [(n, p) for n, p in named if p._shadow]
        self._layout = []
        n_train = sum(p.size for _, p in train)
        n_all = n_train + sum(p.size for _, p in shadow)
        self.params = backend.empty((n_all,), np.float32)
        self.grads = backend.zeros((n_train,), np.float32)
        self.m = backend.zeros((n_train,), np.float32)
        self.v = backend.zeros((n_train,), np.float32)
        off = 0
        for n, p in train + shadow:
            view = self.params.flat_view(off, p.size, p.data.shape)
            view.copy_from(p.data)
            p.data = view
            if not p._shadow:
                p._grad_slot = self.grads.flat_view(off, p.size, view.shape)
                p.grad = None
            p._owner_step = self._step_count        # lets caches notice parameter updates
            self._layout.append((n, off, p.size))
            off += p.size
        self.n_train = n_train
        return self

    def _step_count(self):
        return self.t

    def update(self):
        """One Adam step on every trainable parameter (chainer Adam update rule)."""
        self.t += 1
        _lib.call('vqvae_adam_step', self.params.ptr, self.grads.ptr, self.m.ptr, self.v.ptr,
                  self.n_train, float(self.lr), float(self.beta1), float(self.beta2),
                  float(self.eps), backend.stream())

    def layout(self):
        """[(name, offset, size)] of the flat arena (trainable first)."""
        return list(self._layout)
