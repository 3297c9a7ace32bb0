"""VQ / StraightThrough / ExponentialMovingAverage / MuLaw -- mirrors the
hot-path half of the reference's utils.py (MuLaw utils.py:12-29,
ExponentialMovingAverage 131-158, StraightThrough 161-236, VQ 239-255)."""
import copy
import ctypes as C
import weakref

import numpy as np

from . import _lib, backend, core
from .backend import DeviceArray
from .core import Chain, FunctionNode, Link, Parameter, Variable, type_expect

_S = backend.stream


class MuLaw(object):
    """mu-law companding + binning (host-side; defines the synthetic inputs).  Arithmetic
    order follows utils.py:18-29 so that bin edges fall on the same samples."""

    def __init__(self, mu=256, int_type=np.int32, float_type=np.float32):
        self.mu = mu
        self.int_type = int_type
        self.float_type = float_type

    def transform(self, x):
        x = x.astype(self.float_type)
        magnitude = np.log(1 + self.mu * np.abs(x))
        companded = np.sign(x) * magnitude / np.log(1 + self.mu)       # in [-1, 1]
        edges = 2 * np.arange(self.mu) / self.mu - 1                   # left bin edges
        return (np.digitize(companded, edges) - 1).astype(self.int_type)

    def itransform(self, y):
        centred = 2 * y.astype(self.float_type) / self.mu - 1
        expanded = np.sign(centred) / self.mu * (self.mu ** np.abs(centred) - 1)
        return expanded.astype(self.float_type)


class ExponentialMovingAverage(Chain):
    """utils.py:131-158.  In train mode: run ``target``, then blend every
    parameter ``ema <- decay*target + (1-decay)*ema`` (the reference's weights,
    utils.py:153-154).  Eval mode runs ``ema``.  When the parameters live in the
    optimizer's flat arena the blend is ONE kernel over the two contiguous,
    identically ordered sub-trees instead of the reference's O(P^2) name match."""

    def __init__(self, target, decay=0.999):
        super(ExponentialMovingAverage, self).__init__()
        self.decay = decay
        with self.init_scope():
            self.target = target
            self.ema = copy.deepcopy(target)
        for p in self.ema.params():
            p._shadow = True

    def __call__(self, *args, **kwargs):
        if core.config.train:
            ys = self.target(*args, **kwargs)
            self._blend()
        else:
            ys = self.ema(*args, **kwargs)
        return ys

    def _blend(self):
        tp = list(self.target.namedparams())
        ep = dict(self.ema.namedparams())
        pairs = [(ep[n], p) for n, p in tp if n in ep and p.data is not None]
        if not pairs:
            return
        # contiguous in the arena? (same order in both sub-trees)
        def contiguous(ps):
            cur = ps[0].data.ptr
            for q in ps:
                if not isinstance(q.data, DeviceArray) or q.data.ptr != cur:
                    return False
                cur += q.data.nbytes
            return True
        es = [e for e, _ in pairs]
        ts = [t for _, t in pairs]
        if contiguous(es) and contiguous(ts):
            n = sum(t.data.size for t in ts)
            _lib.call('vqvae_ema_step', es[0].data.ptr, ts[0].data.ptr, n, float(self.decay), _S())
            return
        for e, t in pairs:
            if not t.requires_grad or e.data is None:
                e.data = t.data
            else:
                backend.require_device(e.data, t.data)
                _lib.call('vqvae_ema_step', e.data.ptr, t.data.ptr, t.data.size,
                          float(self.decay), _S())


class StraightThrough(FunctionNode):
    """utils.py:161-231.  forward: idx = argmin_j sum_c (x - W_j)^2 (bit-exact
    with the NumPy path), e = W[idx] laid out (B,d,T[,1]).  backward: gx = gy
    (identity, same object), gW = onehot^T gy accumulated in float64."""

    mode = 0          # 0: MFMA + exact re-check; 1: exact everywhere (tests)
    preset = None     # (indexes, embeded) of an identical earlier search, or None

    def check_type_forward(self, in_vars):
        type_expect((len(in_vars) == 2, 'StraightThrough takes (x, W)'))
        x, w = in_vars
        type_expect((np.dtype(x.dtype).kind == 'f', 'x must be float'),
                    (np.dtype(w.dtype).kind == 'f', 'W must be float'),
                    (x.ndim >= 3, 'x.ndim >= 3'),
                    (x.ndim <= 4, 'x.ndim <= 4'),
                    (w.ndim == 2, 'W.ndim == 2'),
                    (x.shape[1] == w.shape[1], 'x.shape[1] == W.shape[1]'))

    def forward(self, inputs):
        self.retain_inputs((0, 1))
        xs, W = inputs
        if not (isinstance(xs, DeviceArray) and isinstance(W, DeviceArray)):
            raise ValueError('numpy and device arrays must not be used together\n'
                             'type(W): {0}, type(x): {1}'.format(type(W), type(xs)))
        B, d, T = xs.shape[:3]
        k = W.shape[0]
        self._dims = (B, d, T, k)
        if self.preset is not None:
            self.indexes, embeded = self.preset
            self.n_rechecked = None
            return embeded,
        idx_shape = (B, T, 1) if xs.ndim == 4 else (B, T)
        self.indexes = DeviceArray(idx_shape, np.int32)
        embeded = DeviceArray(xs.shape, np.float32)
        self.n_rechecked = DeviceArray((1,), np.int32)
        ws = backend.workspace(_lib.load().vqvae_vq_workspace_bytes(B, d, T, k))
        _lib.call('vqvae_vq_nearest_fwd', xs.ptr, W.ptr, B, d, T, k, self.mode, self.indexes.ptr,
                  embeded.ptr, self.n_rechecked.ptr, ws.ptr, ws.nbytes, _S())
        self._dims = (B, d, T, k)
        return embeded,

    def backward(self, indexes, grad_outputs):
        xs, W = self.get_retained_inputs()
        gy, = grad_outputs
        ret = []
        if 0 in indexes:
            ret.append(gy)
        else:
            ret.append(None)
        if 1 in indexes:
            B, d, T, k = self._dims
            gW = DeviceArray((k, d), np.float32)
            ws = backend.workspace(_lib.load().vqvae_vq_workspace_bytes(B, d, T, k))
            _lib.call('vqvae_vq_grad_w', self.indexes.ptr, gy.data.ptr, B, d, T, k, gW.ptr, 0,
                      ws.ptr, ws.nbytes, _S())
            ret.append(Variable(gW))
        else:
            ret.append(None)
        return ret


def _straight_through_node(x, W, preset=None):
    node = StraightThrough()
    node.preset = preset
    y, = node.apply((x, W))
    return y, node


def straight_through(x, W):
    """utils.py:234-236."""
    return _straight_through_node(x, W)[0]


class VQ(Link):
    """utils.py:239-255."""

    def __init__(self, k, d=None, initialW=None):
        super(VQ, self).__init__()
        self.k = k
        with self.init_scope():
            W_initializer = core._get_initializer(initialW)
            self.W = Parameter(W_initializer)
            if d is not None:
                self._initialize_params(d)

    def _initialize_params(self, d):
        self.W.initialize((self.k, d))

    def __call__(self, x):
        if self.W.data is None:
            self._initialize_params(x.shape[1])
            if isinstance(x.data, DeviceArray):
                self.W.to_gpu()
        # the reference quantises the same latents twice per step (net.py:82-83); the second
        # application reuses the first search while both arrays are still alive and unchanged
        key = (x.data, self.W.data)
        preset = None
        c = self._cache
        if c is not None and c[0]() is key[0] and c[1]() is key[1] and c[2] == self._w_version():
            preset = c[3]
        y, node = _straight_through_node(x, self.W, preset)
        if preset is None and isinstance(x.data, DeviceArray):
            self._cache = (weakref.ref(key[0]), weakref.ref(key[1]), self._w_version(),
                           (node.indexes, y.data))
        return y

    _cache = None

    def _w_version(self):
        opt = getattr(self.W, '_owner_step', None)
        return (opt() if opt is not None else 0, core.param_epoch('load'))
