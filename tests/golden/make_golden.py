#!/usr/bin/env python
"""Generates tests/golden/*.npz by EXECUTING the reference's own code.

Dev-container tooling only (needs /root/reference, which never travels to the
GPU box).  The reference's ``utils.py`` is imported from /root/reference via
sys.path -- nothing is copied.  ``chainer`` and ``librosa`` are not installed,
so import-time stand-ins are injected first: only the
names ``utils.py`` touches at import time plus the four helpers its NumPy bodies
call (FunctionNode.retain_inputs/get_retained_inputs, cuda.get_array_module,
type_check.same_types, a .data/.transpose/.reshape Variable).  All arithmetic
that ends up in the fixtures is the reference's:

  * utils.MuLaw.transform / itransform           (utils.py:18-29)
  * utils.StraightThrough.forward                (utils.py:176-211)
  * utils.StraightThrough.backward               (utils.py:213-231)

Fixtures hold inputs (or the seed that regenerates them) and expected outputs.

Run:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def _install_stubs():
    ch = types.ModuleType('chainer')

    class FunctionNode(object):
        def retain_inputs(self, idx):
            self._ret = idx

        def get_retained_inputs(self):
            return tuple(self._inputs[i] for i in self._ret)

    class Variable(object):
        def __init__(s, a):
            s.data = s.array = a
        ndim = property(lambda s: s.data.ndim)
        shape = property(lambda s: s.data.shape)
        dtype = property(lambda s: s.data.dtype)

        def transpose(s, ax):
            return Variable(s.data.transpose(ax))

        def reshape(s, sh):
            return Variable(s.data.reshape(sh))

    def mod(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    ch.function_node = mod('chainer.function_node', FunctionNode=FunctionNode)
    ch.cuda = mod('chainer.cuda', get_array_module=lambda *a: np)
    tc = mod('chainer.utils.type_check', expect=lambda *a: None,
             same_types=lambda *a: all(isinstance(x, np.ndarray) for x in a))
    ch.utils = mod('chainer.utils', type_check=tc)

    class Link(object):
        pass

    class Chain(Link):
        pass

    ch.link = mod('chainer.link', Link=Link, Chain=Chain)
    ch.configuration = mod('chainer.configuration')
    ch.Variable = Variable
    ch.Chain = Chain
    sys.modules['chainer'] = ch
    mod('librosa')
    return Variable


def ref_vq(utils, Variable, z, W, gy):
    st = utils.StraightThrough()
    st._inputs = (z, W)
    (e,) = st.forward((z, W))
    idx = st.indexes.copy()
    e = np.ascontiguousarray(e)
    gx, gW = st.backward((0, 1), (Variable(gy),))
    assert gx.data is gy
    return e, idx, gW.data


def stress_inputs(seed_z, seed_w, B, d, T, k):
    """SURVEY 8(d) C4 inputs: half the rows N(0,1), half W[j]+0.5 N(0,1)."""
    rw = np.random.RandomState(seed_w)
    W = (rw.standard_normal((k, d)) / np.sqrt(d)).astype(np.float32)
    rz = np.random.RandomState(seed_z)
    N = B * T
    rows = rz.standard_normal((N, d)).astype(np.float32)
    j = rz.randint(0, k, size=N // 2)
    rows[N // 2:] = W[j] + np.float32(0.5) * rz.standard_normal((N - N // 2, d)).astype(np.float32)
    z = np.ascontiguousarray(rows.reshape(B, T, d).transpose(0, 2, 1))[..., None]
    return z, W


def main():
    Variable = _install_stubs()
    sys.path.insert(0, REF)
    import utils  # the reference's utils.py, executed in place

    # ---- mu-law -------------------------------------------------------- #
    rs = np.random.RandomState(7)
    x = np.concatenate([
        np.array([-1, -.5, -1e-4, 0, 1e-4, .5, 1], np.float32),
        rs.uniform(-1, 1, 4096).astype(np.float32),
        np.linspace(-1, 1, 1025).astype(np.float32)])
    mu = utils.MuLaw(256)
    q = mu.transform(x)
    xr = mu.itransform(np.arange(256))
    np.savez_compressed(os.path.join(HERE, 'mulaw.npz'), x=x, q=q, itransform=xr)

    # ---- VQ, training shape (d=64, k=512, T'=120) ---------------------- #
    rs = np.random.RandomState(11)
    B, d, T, k = 2, 64, 120, 512
    z = rs.standard_normal((B, d, T, 1)).astype(np.float32)
    W = (rs.standard_normal((k, d)) / np.sqrt(d)).astype(np.float32)
    gy = rs.standard_normal((B, d, T, 1)).astype(np.float32)
    e, idx, gW = ref_vq(utils, Variable, z, W, gy)
    np.savez_compressed(os.path.join(HERE, 'vq_train.npz'), z=z, W=W, gy=gy, e=e, idx=idx, gW=gW)

    # ---- VQ, 3-D input branch (utils.py:197-199, 209-210) -------------- #
    z3 = rs.standard_normal((3, 16, 37)).astype(np.float32)
    W3 = rs.standard_normal((40, 16)).astype(np.float32)
    gy3 = rs.standard_normal((3, 16, 37)).astype(np.float32)
    e3, idx3, gW3 = ref_vq(utils, Variable, z3, W3, gy3)
    np.savez_compressed(os.path.join(HERE, 'vq_3d.npz'), z=z3, W=W3, gy=gy3, e=e3, idx=idx3, gW=gW3)

    # ---- VQ, ties: duplicate codebook rows + inputs equal to / midway
    #      between codes + exact fp32 ties (first index must win) --------- #
    rs = np.random.RandomState(13)
    B, d, T, k = 2, 32, 48, 96
    W = rs.standard_normal((k, d)).astype(np.float32)
    W[40] = W[7]                      # duplicates: index 7 must win over 40 and 77
    W[77] = W[7]
    W[90] = W[3]
    z = rs.standard_normal((B, d, T, 1)).astype(np.float32)
    z[0, :, 0, 0] = W[7]              # distance exactly 0 to 7/40/77
    z[0, :, 1, 0] = W[90]             # exactly 3/90
    z[0, :, 2, 0] = (W[10] + W[20]) * np.float32(0.5)   # midpoint (near tie)
    z[0, :, 3, 0] = W[77] + np.float32(1e-3)
    z[1, :, 5, 0] = 0.0
    # small-integer grid rows: many exact ties in fp32
    Wg = W.copy()
    Wg[50:60] = rs.randint(-2, 3, size=(10, d)).astype(np.float32)
    z[1, :, 6:16, 0] = rs.randint(-2, 3, size=(d, 10)).astype(np.float32)
    gy = rs.standard_normal((B, d, T, 1)).astype(np.float32)
    e, idx, gW = ref_vq(utils, Variable, z, Wg, gy)
    np.savez_compressed(os.path.join(HERE, 'vq_ties.npz'), z=z, W=Wg, gy=gy, e=e, idx=idx, gW=gW)

    # ---- VQ stress (k=8192, d=128): inputs regenerated from seeds ------- #
    B, d, T, k = 2, 128, 120, 8192
    z, W = stress_inputs(1, 2, B, d, T, k)
    gy = np.random.RandomState(3).standard_normal((B, d, T, 1)).astype(np.float32)
    es, idxs, gWs = [], [], None
    for b in range(B):               # one batch row at a time: the (1,k,d,T',1) temporaries are 0.5 GB each
        zb = z[b:b + 1]
        st = utils.StraightThrough()
        st._inputs = (zb, W)
        (e,) = st.forward((zb, W))
        idxs.append(st.indexes.copy())
        es.append(np.ascontiguousarray(e))
    idx = np.concatenate(idxs, 0)
    e = np.concatenate(es, 0)
    st = utils.StraightThrough()
    st._inputs = (z, W)
    st._ret = (0, 1)
    st.indexes = idx.copy()
    _, gWv = st.backward((0, 1), (Variable(gy),))
    gW = gWv.data
    # outputs are compact: idx in full; e is W[idx] (checked via idx); gW rows are mostly zero
    nz = np.flatnonzero(np.abs(gW).sum(axis=1))
    np.savez_compressed(os.path.join(HERE, 'vq_stress.npz'),
                        seed_z=1, seed_w=2, seed_gy=3, shape=np.array([B, d, T, k]),
                        idx=idx, gW_rows=nz.astype(np.int32), gW_vals=gW[nz],
                        e_sum=np.float64(e.astype(np.float64).sum()))
    # ---- Preprocess output contract (utils.py:54-110): pad / crop / one-hot / speaker id,
    #      with librosa.load / effects.trim stubbed to hand back a synthetic waveform -------- #
    import random
    import tempfile
    lib = sys.modules['librosa']
    waves = {}
    lib.load = lambda path, sr, res_type=None: (waves[os.path.basename(path)].copy(), sr)
    lib.effects = types.SimpleNamespace(trim=lambda raw, top_db: (raw, None))
    root = tempfile.mkdtemp()
    for spk in ('p225', 'p226', 'p227'):
        os.makedirs(os.path.join(root, 'wav48', spk))
    rs = np.random.RandomState(17)
    waves['short.wav'] = (0.3 * rs.standard_normal(100)).astype(np.float32)      # padding branch
    waves['long.wav'] = (0.3 * rs.standard_normal(700)).astype(np.float32)       # cropping branch
    out = {}
    for use_logistic, input_dim, tag in ((False, 256, 'mulaw'), (True, 1, 'logistic')):
        pre = utils.Preprocess(16000, 'kaiser_fast', 20, input_dim, 256, 255, use_logistic, root, 'VCTK')
        for name, spk in (('short.wav', 'p226'), ('long.wav', 'p227')):
            random.seed(5)
            raw, x_dec, speaker, t = pre(os.path.join(root, 'wav48', spk, name))
            key = '%s_%s_' % (tag, name.split('.')[0])
            out[key + 'raw'], out[key + 'x_dec'] = raw, x_dec
            out[key + 'speaker'], out[key + 't'] = speaker, t
    out['wave_short'], out['wave_long'] = waves['short.wav'], waves['long.wav']
    random.seed(5)
    out['crop_start'] = np.int64(random.randint(0, 700 - 256 - 1))               # utils.py:78
    np.savez_compressed(os.path.join(HERE, 'preprocess.npz'), **out)
    print('golden vectors written to', HERE)


if __name__ == '__main__':
    main()
