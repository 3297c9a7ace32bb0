"""Chainer-shaped micro-runtime: Variable / Parameter / FunctionNode / Link /
Chain / ChainList / configuration / reporter.

Only the part of Chainer's surface that the reference's hot path touches is
provided, with the same names and call semantics:

  FunctionNode   check_type_forward, forward, backward, retain_inputs,
                 get_retained_inputs, apply              (utils.py:161-236)
  Link/Chain     init_scope, add_link, children, namedparams, params, cleargrads,
                 addgrads, copyparams, to_gpu, xp        (net.py:8-17,
                 modules.py:77-90, utils.py:131-148, updaters.py:14-16,42,72,77)
  Variable       data/array/shape/reshape/backward, + - * **, Variable(arr) as a
                 stop-gradient                           (net.py:58-59,83,90-91)

All arithmetic is executed by libvqvae_hip.so on device arrays; host (NumPy)
arrays are only legal as storage before ``to_gpu()``.
"""
import collections
import contextlib
import heapq
import weakref

import numpy as np

from . import backend
from .backend import DeviceArray


# --------------------------------------------------------------------------- #
# configuration (chainer.configuration.config / chainer.using_config)
# --------------------------------------------------------------------------- #
class _Config(object):
    train = True
    enable_backprop = True


config = _Config()
global_config = config


@contextlib.contextmanager
def using_config(name, value):
    old = getattr(config, name)
    setattr(config, name, value)
    try:
        yield
    finally:
        setattr(config, name, old)


def no_backprop_mode():
    return using_config('enable_backprop', False)


def force_backprop_mode():
    return using_config('enable_backprop', True)


# --------------------------------------------------------------------------- #
# reporter (chainer.reporter.report, net.py:93-95)
# --------------------------------------------------------------------------- #
class Reporter(object):
    """chainer.Reporter: observers report into the CURRENT observation dict.  ``scope`` swaps
    the dict for the duration of a block, which is how Chainer's Evaluator keeps a validation
    pass from overwriting the training observations ('main/loss*', net.py:93-95)."""

    def __init__(self):
        self.observation = {}
        self._names = {}

    def add_observer(self, name, observer):
        self._names[id(observer)] = name

    def report(self, values, observer=None):
        prefix = ''
        if observer is not None:
            prefix = self._names.get(id(observer), 'main') + '/'
        for k, v in values.items():
            self.observation[prefix + k] = v

    @contextlib.contextmanager
    def scope(self, observation):
        old = self.observation
        self.observation = observation
        try:
            yield observation
        finally:
            self.observation = old


_reporter = Reporter()


def report(values, observer=None):
    _reporter.report(values, observer)


def get_current_reporter():
    return _reporter


def report_scope(observation):
    """chainer.reporter.report_scope."""
    return _reporter.scope(observation)


# Parameter life-cycle epochs.  'init' advances whenever a Parameter gets (new) host/device
# storage outside an optimizer arena -- lazily shaped convs (net.py:34-43 builds
# DilatedConvolution2D(None, ...)) are created at the first forward, AFTER optimizer.setup in
# train.py's order (train.py:76-102) -- so an optimizer can notice parameters it has not
# adopted yet.  'load' advances when a serializer overwrites parameter values, so value-keyed
# caches (VQ search reuse, packed generation weights) drop what they hold.  'layout' advances
# when an optimizer moves parameters into (new) arenas: raw device pointers captured before
# that (GenerationState) are stale.
_epochs = {'init': 0, 'load': 0, 'layout': 0}


def param_epoch(kind):
    return _epochs[kind]


def bump_param_epoch(kind):
    _epochs[kind] += 1


# --------------------------------------------------------------------------- #
# Variable
# --------------------------------------------------------------------------- #
class Variable(object):
    """chainer.Variable.  ``Variable(arr)`` creates a leaf without history, which
    the reference uses as stop-gradient (net.py:83, 90-91)."""

    def __init__(self, data=None, name=None, requires_grad=True):
        if isinstance(data, Variable):
            data = data.data
        self._data = data
        self.name = name
        self.grad = None
        self.creator = None
        self.rank = 0
        self.requires_grad = requires_grad

    # data / array
    @property
    def data(self):
        return self._data

    @data.setter
    def data(self, d):
        self._data = d

    array = data

    @property
    def shape(self):
        return self._data.shape

    @property
    def ndim(self):
        return len(self._data.shape)

    @property
    def dtype(self):
        return self._data.dtype

    @property
    def size(self):
        return self._data.size

    def __len__(self):
        return self._data.shape[0]

    def __repr__(self):
        return 'variable(shape=%s)' % (self.shape,)

    def cleargrad(self):
        self.grad = None

    def unchain(self):
        self.creator = None

    def reshape(self, *shape):
        from . import functions as F
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return F.reshape(self, shape)

    # arithmetic (net.py:90-92)
    def __add__(self, o):
        from . import functions as F
        return F.add(self, o)

    __radd__ = __add__

    def __sub__(self, o):
        from . import functions as F
        return F.sub(self, o)

    def __mul__(self, o):
        from . import functions as F
        return F.mul(self, o)

    __rmul__ = __mul__

    def __pow__(self, p):
        from . import functions as F
        if p != 2:
            raise NotImplementedError('only ** 2 is on the hot path (net.py:90-91)')
        return F.square(self)

    def __neg__(self):
        from . import functions as F
        return F.mul(self, -1.0)

    # backward -------------------------------------------------------------
    def backward(self, retain_grad=False):
        """Reverse-mode sweep in decreasing rank order; leaf gradients ACCUMULATE
        into ``.grad`` across calls, which the updaters rely on (three backward()
        calls, updaters.py:15-18)."""
        from . import functions as F
        if self.creator is None:
            return
        if self.grad is None:
            if self.size != 1:
                raise RuntimeError('backward() on a non-scalar needs an explicit grad')
            self.grad = F.full_like(self.data, 1.0)
        grads = {id(self): self.grad}
        keep = {id(self): self}
        heap = []
        seen = set()
        counter = [0]

        def push(fn):
            if id(fn) not in seen:
                seen.add(id(fn))
                counter[0] += 1
                heapq.heappush(heap, (-fn.rank, counter[0], fn))

        push(self.creator)
        while heap:
            _, _, fn = heapq.heappop(heap)
            outs = [r() for r in fn._output_refs]
            gys = []
            for o in outs:
                g = grads.pop(id(o), None) if o is not None else None
                gys.append(None if g is None else Variable(g, requires_grad=False))
            if all(g is None for g in gys):
                continue
            idxs = tuple(i for i, x in enumerate(fn.inputs) if x.requires_grad)
            gxs = fn.backward(idxs, tuple(gys))
            if not isinstance(gxs, (tuple, list)):
                gxs = (gxs,)
            if len(gxs) == len(idxs) and len(idxs) != len(fn.inputs):
                full = [None] * len(fn.inputs)
                for i, g in zip(idxs, gxs):
                    full[i] = g
                gxs = full
            for x, gx in zip(fn.inputs, gxs):
                if gx is None or not x.requires_grad:
                    continue
                if isinstance(gx, Variable):
                    gx = gx.data
                if x.creator is None:
                    x._accumulate_grad(gx)          # leaf
                else:
                    cur = grads.get(id(x))
                    grads[id(x)] = gx if cur is None else F.raw_add(cur, gx)
                    keep[id(x)] = x
                    push(x.creator)
        backend.join_side(force=False)        # weight gradients a node deferred to the side stream (wavenet.ResidualStackFunction)

    def _accumulate_grad(self, gx):
        from . import functions as F
        if self.grad is None:
            self.grad = gx
        else:
            masked = getattr(self.grad, 'relu_masked', False) and getattr(gx, 'relu_masked', False)
            if isinstance(self, Parameter) and backend.side_writes_pending(self.grad, gx):
                backend.join_side()      # (one of the two contributions comes from a launch deferred to the side stream)
            self.grad = F.raw_add(self.grad, gx)
            self.grad.relu_masked = masked       # a sum of masked gradients is the masked sum


class Parameter(Variable):
    """chainer.Parameter: a leaf Variable owned by a Link.  ``initializer`` is a
    callable(shape) -> float32 ndarray, or an ndarray.  After
    ``Optimizer.setup`` the data lives in the optimizer's flat parameter arena
    and gradients accumulate in the flat gradient arena (one RCCL all-reduce /
    one Adam kernel per step)."""

    def __init__(self, initializer=None, shape=None, name=None):
        super(Parameter, self).__init__(None, name=name)
        self.initializer = initializer
        self._grad_slot = None        # DeviceArray view into the flat gradient arena
        self._shadow = False          # EMA shadow copy: never receives gradients
        if isinstance(initializer, np.ndarray):
            self._data = np.ascontiguousarray(initializer, dtype=np.float32)
        elif shape is not None:
            self.initialize(shape)

    def initialize(self, shape):
        init = self.initializer
        if isinstance(init, np.ndarray):
            arr = np.ascontiguousarray(init, np.float32).reshape(shape)
        else:
            arr = np.ascontiguousarray(init(shape), np.float32)
        self._data = arr
        bump_param_epoch('init')

    def grad_buffer(self):
        """Where a backward kernel may write this parameter's gradient directly
        (its slot in the flat gradient arena) -- only while no gradient has been
        accumulated yet; otherwise None and the caller allocates."""
        if self._grad_slot is not None and self.grad is None:
            return self._grad_slot
        return None

    def _accumulate_grad(self, gx):
        from . import functions as F
        if self._grad_slot is not None:
            if gx.ptr == self._grad_slot.ptr:        # written in place by the kernel
                self.grad = self._grad_slot
                return
            gx = gx.reshape(self._grad_slot.shape)
            # (only when gx, or an earlier contribution to the slot, comes from a launch deferred to the side stream: joining for
            #  EVERY copied gradient -- the speaker embedding's is the first of the sweep's tail -- made the main stream wait for all
            #  of the deferred weight gradients, 1.4 ms, in front of the condition-embed / encoder backward they were deferred to
            #  run beside)
            if backend.side_writes_pending(self._grad_slot, gx):
                backend.join_side()
            if self.grad is None:
                self._grad_slot.copy_from(gx)
            else:
                F.raw_add(self._grad_slot, gx, out=self._grad_slot)
            self.grad = self._grad_slot
        else:
            Variable._accumulate_grad(self, gx)

    def to_gpu(self):
        if isinstance(self._data, np.ndarray):
            self._data = backend.to_device(self._data, np.float32)


def as_variable(x):
    if isinstance(x, Variable):
        return x
    return Variable(x, requires_grad=False)


# --------------------------------------------------------------------------- #
# FunctionNode
# --------------------------------------------------------------------------- #
class FunctionNode(object):
    """chainer.FunctionNode (utils.py:161-231 shows the contract used)."""

    inputs = ()
    rank = 0
    _retained = ()

    def check_type_forward(self, in_types):
        pass

    def forward(self, inputs):
        raise NotImplementedError

    def backward(self, target_input_indexes, grad_outputs):
        raise NotImplementedError

    def retain_inputs(self, indexes):
        self._retained = tuple(indexes)

    def get_retained_inputs(self):
        return tuple(self.inputs[i] for i in self._retained)

    def apply(self, inputs):
        in_vars = tuple(as_variable(x) for x in inputs)
        in_data = tuple(v.data for v in in_vars)
        self.check_type_forward(in_vars)
        self.inputs = in_vars
        outputs = self.forward(in_data)
        if not isinstance(outputs, tuple):
            outputs = (outputs,)
        requires = config.enable_backprop and any(v.requires_grad for v in in_vars)
        out_vars = tuple(Variable(o, requires_grad=requires) for o in outputs)
        if requires:
            self.rank = max([v.rank for v in in_vars] + [0])
            for o in out_vars:
                o.creator = self
                o.rank = self.rank + 1
            self._output_refs = tuple(weakref.ref(o) for o in out_vars)
        else:
            self.inputs = ()
        return out_vars


class InvalidType(TypeError):
    """chainer.utils.type_check.InvalidType stand-in."""


def type_expect(*conds):
    for ok, msg in conds:
        if not ok:
            raise InvalidType(msg)


# --------------------------------------------------------------------------- #
# Link / Chain / ChainList
# --------------------------------------------------------------------------- #
class Link(object):
    def __init__(self):
        self._params = []
        self._within_init_scope = False
        self.name = None

    @contextlib.contextmanager
    def init_scope(self):
        old = self._within_init_scope
        self._within_init_scope = True
        try:
            yield
        finally:
            self._within_init_scope = old

    def __setattr__(self, name, value):
        if getattr(self, '_within_init_scope', False) and isinstance(value, Parameter):
            value.name = name
            if name not in self._params:
                self._params.append(name)
        object.__setattr__(self, name, value)

    @property
    def xp(self):
        return backend

    def params(self, include_uninit=True):
        for _, p in self.namedparams(include_uninit):
            yield p

    def namedparams(self, include_uninit=True):
        d = self.__dict__
        for name in sorted(self._params):
            if include_uninit or d[name].data is not None:
                yield '/' + name, d[name]

    def links(self, skipself=False):
        if not skipself:
            yield self

    def children(self):
        return iter(())

    def cleargrads(self):
        """Link.cleargrads (updaters.py:14,16).  With a flat gradient arena the
        contiguous run of slots is zeroed by one memset."""
        ps = [p for p in self.params() if p.data is not None]
        slots = [p for p in ps if p._grad_slot is not None]
        dirty = [p for p in slots if p.grad is not None]
        if dirty:
            # the slots of one Link sub-tree are one contiguous run of the arena
            # (namedparams order); zeroing never-written (already zero) slots is harmless
            lo = min(p._grad_slot.ptr for p in slots)
            hi = max(p._grad_slot.ptr + p._grad_slot.nbytes for p in slots)
            tot = sum(p._grad_slot.nbytes for p in slots)
            from . import _lib
            if hi - lo == tot:
                _lib.call('vqvae_memset', lo, 0, tot, backend.stream())
            else:
                for p in dirty:
                    p._grad_slot.fill_zero()
        for p in ps:
            p.grad = None

    zerograds = cleargrads

    def to_gpu(self, device=None):
        backend.init(0 if device is None else device)
        # one transfer for all parameters that are still on the host (they become views of one buffer; the optimizer's
        # arenas adopt them later): a transfer per parameter was ~530 copy dispatches for this model
        host = [p for p in self.params() if isinstance(p._data, np.ndarray)]
        for p, d in zip(host, backend.to_device_many([p._data for p in host]) if host else []):
            p._data = d
        return self

    def addgrads(self, link):
        """Link.addgrads (updaters.py:72): self.grad += link.grad, matched by name."""
        from . import functions as F
        src = dict(link.namedparams())
        for name, p in self.namedparams():
            q = src[name]
            if q.grad is None:
                continue
            p._accumulate_grad(q.grad)

    def copyparams(self, link):
        """Link.copyparams (updaters.py:77)."""
        src = dict(link.namedparams())
        for name, p in self.namedparams():
            q = src[name]
            if isinstance(p.data, DeviceArray):
                if isinstance(q.data, DeviceArray):
                    p.data.copy_from(q.data)
                else:
                    p.data.set(q.data)
            else:
                p.data = np.array(q.data.get() if isinstance(q.data, DeviceArray) else q.data)

    def count_params(self):
        return sum(p.size for p in self.params() if p.data is not None)


class Chain(Link):
    def __init__(self):
        super(Chain, self).__init__()
        self._children = []

    def __setattr__(self, name, value):
        if getattr(self, '_within_init_scope', False) and isinstance(value, Link):
            if name not in self._children:
                self._children.append(name)
            value.name = name
        Link.__setattr__(self, name, value)

    def add_link(self, name, link):
        with self.init_scope():
            setattr(self, name, link)

    def namedparams(self, include_uninit=True):
        for ret in Link.namedparams(self, include_uninit):
            yield ret
        d = self.__dict__
        for name in sorted(self._children):
            prefix = '/' + name
            for path, p in d[name].namedparams(include_uninit):
                yield prefix + path, p

    def links(self, skipself=False):
        if not skipself:
            yield self
        for name in sorted(self._children):
            for l in self.__dict__[name].links():
                yield l

    def children(self):
        for name in sorted(self._children):
            yield self.__dict__[name]

    def __getitem__(self, name):
        return self.__dict__[name]


class ChainList(Link):
    def __init__(self, *links):
        super(ChainList, self).__init__()
        self._children = []
        for l in links:
            self.add_link(l)

    def add_link(self, link):
        link.name = str(len(self._children))
        self._children.append(link)

    def __getitem__(self, i):
        return self._children[i]

    def __len__(self):
        return len(self._children)

    def __iter__(self):
        return iter(self._children)

    def namedparams(self, include_uninit=True):
        for ret in Link.namedparams(self, include_uninit):
            yield ret
        for i, link in enumerate(self._children):
            prefix = '/%d' % i
            for path, p in link.namedparams(include_uninit):
                yield prefix + path, p

    def links(self, skipself=False):
        if not skipself:
            yield self
        for c in self._children:
            for l in c.links():
                yield l

    def children(self):
        for c in self._children:
            yield c


# --------------------------------------------------------------------------- #
# initializers (chainer.initializers; utils.py:244 `_get_initializer(None)`)
# --------------------------------------------------------------------------- #
_init_rng = np.random.RandomState(0)


def seed_initializers(seed):
    global _init_rng
    _init_rng = np.random.RandomState(seed)


class LeCunNormal(object):
    """chainer's default weight initializer: N(0, 1/fan_in), fan_in = prod(shape[1:])."""

    def __init__(self, scale=1.0):
        self.scale = scale

    def __call__(self, shape):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        return (self.scale * _init_rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)


class Normal(object):
    def __init__(self, scale=1.0):
        self.scale = scale

    def __call__(self, shape):
        return (self.scale * _init_rng.standard_normal(shape)).astype(np.float32)


class Constant(object):
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, shape):
        return np.full(shape, self.value, dtype=np.float32)


def _get_initializer(initializer):
    """chainer.initializers._get_initializer (utils.py:244)."""
    if initializer is None:
        return LeCunNormal()
    if np.isscalar(initializer):
        return Constant(initializer)
    if isinstance(initializer, np.ndarray):
        return initializer
    if not callable(initializer):
        raise TypeError('invalid type of initializer: %s' % type(initializer))
    return initializer
