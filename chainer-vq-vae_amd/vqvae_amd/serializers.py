"""Snapshot I/O in the reference's on-disk format.

The reference snapshots the whole Chainer Trainer with ``extensions.snapshot()``
(train.py:134) and ``generate.py`` reads sub-trees back by key prefix,
``load_npz(model, encoder, 'updater/model:main/encoder/')`` (generate.py:67-81).
The ``.npz`` is a flat dict whose keys are '/'-joined paths [chainer-recalled:
DictionarySerializer strips the leading '/' of namedparams paths]:

    updater/model:main/<link path>/<param>            parameters (decoder/{target,ema}/...)
    updater/optimizer:main/<link path>/<param>/{m,v,t}  Adam state per parameter
    updater/optimizer:main/{t,epoch}                  optimizer counters
    updater/iteration                                 updater counter

Parameter shapes equal Chainer's ((Cout,Cin,K,1) convs, (n,G) EmbedID, (k,d) VQ), so
a snapshot written here loads in the reference's generate.py and vice versa.
"""
import numpy as np

from . import core
from .backend import DeviceArray
from .core import Link

MODEL_PREFIX = 'updater/model:main/'
OPT_PREFIX = 'updater/optimizer:main/'


def _host(a):
    return a.get() if isinstance(a, DeviceArray) else np.asarray(a)


def model_state(link, prefix=MODEL_PREFIX):
    return {prefix + name.strip('/'): _host(p.data)
            for name, p in link.namedparams() if p.data is not None}


def optimizer_state(opt, prefix=OPT_PREFIX):
    out = {prefix + 't': np.asarray(opt.t, np.int32), prefix + 'epoch': np.asarray(0, np.int32)}
    m, v = opt.m.get(), opt.v.get()
    for name, off, size in opt.layout():
        if off + size > opt.n_train:        # EMA shadows: no update rule state
            continue
        key = prefix + name.strip('/') + '/'
        shape = dict(opt.target.namedparams())[name].data.shape
        out[key + 'm'] = m[off:off + size].reshape(shape)
        out[key + 'v'] = v[off:off + size].reshape(shape)
        out[key + 't'] = np.asarray(opt.t, np.int32)
    return out


def save_npz(file, obj, compression=True):
    """``obj`` is an updater (model + optimizer + iteration) or a bare Link."""
    state = {}
    if isinstance(obj, Link):
        state.update(model_state(obj, ''))
    else:
        opt = obj.get_optimizer('main')
        state.update(model_state(opt.target))
        state.update(optimizer_state(opt))
        state['updater/iteration'] = np.asarray(getattr(obj, 'iteration', 0), np.int32)
    (np.savez_compressed if compression else np.savez)(file, **state)


def load_npz(file, obj, path='', strict=True):
    """Loads a Link from the sub-tree ``path`` (generate.py:67-81 usage), or -- when
    ``obj`` is an updater -- model, Adam state and the iteration counter."""
    with np.load(file) as f:
        if isinstance(obj, Link):
            _load_link(f, obj, path, strict)
            return
        opt = obj.get_optimizer('main')
        _load_link(f, opt.target, MODEL_PREFIX, strict)
        opt.adopt_new_params()
        m, v = opt.m.get(), opt.v.get()
        for name, off, size in opt.layout():
            key = OPT_PREFIX + name.strip('/') + '/'
            if key + 'm' in f:
                m[off:off + size] = f[key + 'm'].reshape(-1)
                v[off:off + size] = f[key + 'v'].reshape(-1)
        opt.m.set(m)
        opt.v.set(v)
        if OPT_PREFIX + 't' in f:
            opt.t = int(f[OPT_PREFIX + 't'])
        if 'updater/iteration' in f:
            obj.iteration = int(f['updater/iteration'])


def _load_link(f, link, prefix, strict):
    # a model that already lives on the device gets lazily shaped parameters (net.py:34-43:
    # DilatedConvolution2D(None, ...)) created ON the device, and the 'init' epoch tells an
    # optimizer that was set up before to adopt them (optimizers.Adam.adopt_new_params)
    on_device = any(isinstance(p.data, DeviceArray) for p in link.params())
    for name, p in link.namedparams():
        key = prefix + name.strip('/')
        if key not in f:
            if strict:
                raise KeyError('snapshot has no entry %r' % key)
            continue
        arr = f[key]
        if p.data is not None and tuple(p.data.shape) != tuple(arr.shape):
            raise ValueError('shape mismatch for %s: snapshot %s vs link %s'
                             % (key, arr.shape, p.data.shape))
        if isinstance(p.data, DeviceArray):
            p.data.set(arr)
        else:
            created = p.data is None
            p.data = np.ascontiguousarray(arr, np.float32)
            if on_device:
                p.to_gpu()
            if created:
                core.bump_param_epoch('init')
    core.bump_param_epoch('load')       # value-keyed caches (VQ search reuse, generation state) are stale
