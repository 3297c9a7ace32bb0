"""GPU checks at BASELINE configs[1] sizes (B = 16, T = 7680, 256 channels, dilations to 512), where
the NumPy oracle is too slow to restate everything: one residual block against the oracle at full
channel / time extent, and size-independent properties of the path -- linearity and adjointness
of the conv contractions (forward, backward-data, backward-weight), causality, independence of
the batch elements, the ln(256) starting loss and agreement of data-parallel shards."""
import os
import sys

import numpy as np
import pytest

import helpers as H
import vqvae_oracle as O
from helpers import assert_close, assert_close_scaled, to4

pytestmark = pytest.mark.gpu

B, T, C = 16, 7680, 256
FULL = dict(d=64, k=512, n_loop=2, n_layer=10, filter_size=2, input_dim=256, residual=256,
            dilated=256, skip=256, out_dim=256, local_dim=64, global_dim=128, n_speaker=109)


def _dev(gpu, a):
    return gpu.to_device(np.ascontiguousarray(a))


def _dot(a, b):
    return float(np.vdot(a.astype(np.float64), b.astype(np.float64)))


def test_resblock_full_extent_vs_oracle(gpu):
    """One ResidualBlock (modules.py:30-56) at the real channel counts, T = 7680, dilation 512."""
    from test_gpu_kernels import _rb_params
    from vqvae_amd.core import Variable
    from vqvae_amd.wavenet import ResidualBlockFunction
    rs = np.random.RandomState(3)
    p = _rb_params(rs, 256, 256, 256, 192, 2)
    x = rs.standard_normal((1, 256, T)).astype(np.float32)
    c = rs.standard_normal((1, 192, T)).astype(np.float32)
    res_ref, skip_ref, cache = O.resblock_fwd(p, x, c, 512)
    g_res = rs.standard_normal(res_ref.shape).astype(np.float32)
    g_skip = rs.standard_normal(skip_ref.shape).astype(np.float32)
    gx_ref, gc_ref, gr = O.resblock_bwd(p, cache, c, 512, g_res, g_skip)
    order = ['conv', 'condition_proj', 'res', 'skip']
    vs = [Variable(_dev(gpu, to4(x))), Variable(_dev(gpu, to4(c)))]
    for n in order:
        vs += [Variable(_dev(gpu, to4(p[n][0]))), Variable(_dev(gpu, p[n][1]))]
    res, skip = ResidualBlockFunction(512).apply(vs)
    assert_close(res.data.get(), res_ref, 1e-4, 'res')
    assert_close(skip.data.get(), skip_ref, 1e-4, 'skip')
    gouts = res.creator.backward(tuple(range(10)), (Variable(_dev(gpu, to4(g_res))), Variable(_dev(gpu, to4(g_skip)))))
    assert_close_scaled(gouts[0].get(), gx_ref, 1e-4, 'gx')
    assert_close_scaled(gouts[1].get(), gc_ref, 1e-4, 'gcond')
    for i, n in enumerate(order):
        assert_close_scaled(gouts[2 + 2 * i].get(), gr[n][0], 1e-4, 'gW ' + n)
        assert_close_scaled(gouts[3 + 2 * i].get(), gr[n][1], 1e-4, 'gb ' + n)


@pytest.mark.parametrize('dil', [1, 2, 64, 512])
def test_dilated_conv_linearity_and_adjoints_full_size(gpu, dil):
    """The causal dilated conv (modules.py:13-16, 40-41) at B = 16, T = 7680, 256 -> 256:
    linear in x; <gy, conv(x) - b> == <bwd_data(gy), x> == <bwd_weight(x, gy), W>."""
    from vqvae_amd import functions as F
    from vqvae_amd.core import Variable
    rs = np.random.RandomState(10 + dil)
    x1 = rs.standard_normal((B, C, T, 1)).astype(np.float32)
    x2 = rs.standard_normal((B, C, T, 1)).astype(np.float32)
    W = (rs.standard_normal((C, C, 2, 1)) / np.sqrt(2 * C)).astype(np.float32)
    b = rs.standard_normal(C).astype(np.float32)
    gy = rs.standard_normal((B, C, T, 1)).astype(np.float32)

    def conv(x):
        vx, vW, vb = Variable(_dev(gpu, x)), Variable(_dev(gpu, W)), Variable(_dev(gpu, b))
        return F.convolution_1d(vx, vW, vb, pad=dil, dilate=dil, out_len=T), vx, vW, vb
    y1, vx1, vW1, vb1 = conv(x1)
    y2 = conv(x2)[0].data.get()
    y12 = conv(0.5 * x1 + x2)[0].data.get()
    y1h = y1.data.get()
    bb = b[None, :, None, None]
    assert_close_scaled(y12 - bb, 0.5 * (y1h - bb) + (y2 - bb), 1e-4, 'linearity')
    # causality: y[..., t] only sees x[..., <= t]
    x3 = x1.copy()
    x3[:, :, 5000:] = rs.standard_normal((B, C, T - 5000, 1))
    np.testing.assert_array_equal(conv(x3)[0].data.get()[:, :, :5000], y1h[:, :, :5000])
    y1.grad = _dev(gpu, gy)
    y1.backward()
    lhs = _dot(gy, y1h - bb)
    # a dot product of N ~ 3e7 unit-variance terms has standard deviation sqrt(N): 1e-4 of that scale
    tol = 1e-4 * np.linalg.norm(gy.astype(np.float64)) * np.linalg.norm((y1h - bb).astype(np.float64)) / np.sqrt(gy.size)
    assert abs(_dot(vx1.grad.get(), x1) - lhs) <= tol, ('bwd-data adjoint', _dot(vx1.grad.get(), x1), lhs, tol)
    assert abs(_dot(vW1.grad.get(), W) - lhs) <= tol, ('bwd-weight adjoint', _dot(vW1.grad.get(), W), lhs, tol)
    assert_close_scaled(vb1.grad.get(), gy.sum(axis=(0, 2, 3), dtype=np.float64), 1e-4, 'gb')


def _full_model(seed):
    P, model = H.build_model(dict(FULL), seed=seed)
    model.to_gpu()
    return P, model


@pytest.mark.parametrize('mode', ['float32x2', 'float32x3'])
def test_full_size_step_properties(gpu, mode):
    """Whole configs[1] training step (B = 16): starting loss1 ~ ln 256 (loss1.png starts at 5.5),
    loss3 = beta * loss2, batch elements do not interact (sample 5 alone gives the same logits), and the
    decoder is causal end to end.  'float32x3': bit for bit.  'float32x2' scales every tensor by a power of
    two taken from ITS absolute maximum -- over the whole batch, over all times -- so another batch composition or
    a change in the future can move a scale and with it the last bits of everything: there the two properties
    hold to fp32 rounding (1e-5 of the logits' scale; measured 2.4e-6)."""
    from vqvae_amd.core import Variable
    import vqvae_amd as V
    gpu.set_matmul_dtype(mode)
    try:
        _full_size_step_properties(gpu, mode, Variable, V)
    finally:
        gpu.set_matmul_dtype(gpu.default_matmul_dtype())


def _full_size_step_properties(gpu, mode, Variable, V):
    P, model = _full_model(2)

    def same(a, b):
        if mode == 'float32x3':
            np.testing.assert_array_equal(a, b)
        else:
            assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max()
    x_enc, x_dec, spk, t = O.synth_batch(B, length=T, n_speaker=FULL['n_speaker'], seed=71)
    args = [Variable(_dev(gpu, x_enc[..., None])), Variable(_dev(gpu, x_dec[..., None])),
            Variable(_dev(gpu, spk)), Variable(_dev(gpu, t[..., None]))]
    l1, l2, l3 = [float(v.data.get()) for v in model(*args)]
    # random logits of modest variance: cross-entropy a little above ln 256 = 5.545 (loss1.png starts at ~5.5)
    assert abs(l1 - np.log(256)) < 0.6, l1
    assert abs(l3 - 0.25 * l2) <= 1e-6 * max(1.0, abs(l2))

    with V.using_config('train', False), V.core.no_backprop_mode():
        def logits(xe, xd, sp):
            z = model.encoder(Variable(_dev(gpu, xe[..., None])))
            e = model.vq(z)
            cond = model.condition_embed(e, Variable(_dev(gpu, sp)))
            return model.decoder(Variable(_dev(gpu, xd[..., None])), cond).data.get()
        y_all = logits(x_enc, x_dec, spk)
        y_5 = logits(x_enc[5:6], x_dec[5:6], spk[5:6])
        same(y_all[5:6], y_5)
        # end-to-end causality of the decoder input (the condition is left alone)
        xd2 = x_dec[5:6].copy()
        xd2[:, :, 6000:] = np.roll(xd2[:, :, 6000:], 7, axis=1)
        y_5b = logits(x_enc[5:6], xd2, spk[5:6])
        same(y_5b[:, :, :6000], y_5[:, :, :6000])
        assert np.abs(y_5b[:, :, 6000:] - y_5[:, :, 6000:]).max() > 1e-3


def test_full_size_gradients_are_sums_over_shards(gpu):
    """updaters.py:37-38, 71-72 at configs[2] shard size: the gradient of a 16-example minibatch's
    SUMMED per-shard losses equals the sum of the two 8-example shard gradients (what the RCCL
    all-reduce adds up), to fp32 summation tolerance."""
    import vqvae_amd as V
    from vqvae_amd.optimizers import Adam
    from test_gpu_model import _Iter, _grads_by_name
    batch = O.synth_batch(B, length=T, n_speaker=FULL['n_speaker'], seed=72)

    def grads(sub):
        P, model = _full_model(4)
        opt = Adam(2e-4)
        opt.setup(model)
        upd = V.VQVAE_StandardUpdater(_Iter([tuple(a[sub] for a in batch)]), opt, device=0)
        upd.update()
        return _grads_by_name(model, opt, False)
    g_a, g_b = grads(slice(0, B, 2)), grads(slice(1, B, 2))       # batch[0::2], batch[1::2]
    g_all = grads(slice(0, B))
    for name, g in g_all.items():
        # every loss is a MEAN over its minibatch: full-batch grad = (g_a + g_b) / 2
        assert_close_scaled(g, 0.5 * (g_a[name] + g_b[name]), 2e-4, 'shard sum ' + name)


def _vq_indices(gpu, z_dev, W_dev, B, d, T, k, mode):
    """vqvae_vq_nearest_fwd through the C ABI on resident inputs -> (idx host (B, T), rows re-checked)."""
    import ctypes as C
    from vqvae_amd import _lib
    from vqvae_amd.backend import DeviceArray
    idx = DeviceArray((B, T), np.int32)
    nre = DeviceArray((1,), np.int32)
    ws = gpu.workspace(_lib.load().vqvae_vq_workspace_bytes(B, d, T, k))
    _lib.call('vqvae_vq_nearest_fwd', z_dev.ptr, W_dev.ptr, B, d, T, k, mode, idx.ptr, None, nre.ptr, ws.ptr, ws.nbytes,
              gpu.stream())
    return idx.get(), int(nre.get()[0])


def test_vq_configs3_full_size_default_path_equals_exact_path(gpu):
    """BASELINE configs[3] at the size bench.py --workload c4 runs (N = 1 048 560 latent rows = 8 738 x 120, k = 8192,
    d = 128; SURVEY 8d inputs): the default search (MFMA sweep on the matrix pipe, candidate lists of the flagged
    rows, reference-order distances of the candidates) must give the index of the all-exact path (mode 1: every code
    of every row in the reference's operation order, utils.py:189-203) for EVERY row, and the oracle's own argmin on
    a sample of rows.  (Until round 4 the full size was only spot-checked on 64 rows inside the bench.)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import bench
    N, d, k, T = 1048560, 128, 8192, 120
    B = N // T
    rows, W = bench.vq_stress_inputs(N, d, k)
    z = np.ascontiguousarray(rows.reshape(B, T, d).transpose(0, 2, 1))
    zd, Wd = gpu.to_device(z), gpu.to_device(W)
    idx0, nre = _vq_indices(gpu, zd, Wd, B, d, T, k, 0)
    idx1, _ = _vq_indices(gpu, zd, Wd, B, d, T, k, 1)
    np.testing.assert_array_equal(idx0, idx1)
    assert 0 < nre < 0.2 * N
    pick = np.linspace(0, N - 1, 96).astype(np.int64)
    got = idx0.reshape(-1)
    for r in pick:
        dist = np.zeros(k, np.float32)
        for c in range(d):
            dist = dist + (rows[r, c] - W[:, c]) ** 2          # utils.py:189-203 summation order
        assert int(np.argmin(dist)) == int(got[r]), r


def test_vq_candidate_list_overflow_takes_the_all_codes_path(gpu):
    """The candidate re-check lists at most 16 codes per flagged row; a row with more candidates inside the band must
    fall through to the all-codes re-check (vq_exact_cand_kernel -> overflow list).  Crafted input at the size where
    the candidate path is active (N >= 4096 rows, k >= 1024): a quarter of the rows sit exactly on a code that the
    codebook holds 40 times (40 exact ties: first index wins, utils.py:207 numpy.argmin), another quarter on a code held
    17 times with perturbations of 1 ulp in one component (17 near-ties); the rest are ordinary rows.  Every row must
    equal the all-exact path and the duplicated rows must return the FIRST of their copies."""
    rs = np.random.RandomState(77)
    B, T, d, k = 64, 120, 128, 4096
    N = B * T
    W = (rs.standard_normal((k, d)) / np.sqrt(d)).astype(np.float32)
    dup = rs.permutation(k)
    c40, c17 = np.sort(dup[:40]), np.sort(dup[40:57])
    W[c40] = W[c40[0]]
    base = W[c17[0]].copy()
    for j, c in enumerate(c17):
        W[c] = base
        if j:
            W[c, j % d] = np.nextafter(base[j % d], np.float32(np.inf))
    rows = (rs.standard_normal((N, d))).astype(np.float32)
    kind = rs.randint(0, 4, N)
    rows[kind == 0] = W[c40[0]]
    rows[kind == 1] = base
    z = np.ascontiguousarray(rows.reshape(B, T, d).transpose(0, 2, 1))
    zd, Wd = gpu.to_device(z), gpu.to_device(W)
    idx0, nre = _vq_indices(gpu, zd, Wd, B, d, T, k, 0)
    idx1, _ = _vq_indices(gpu, zd, Wd, B, d, T, k, 1)
    np.testing.assert_array_equal(idx0, idx1)
    flat = idx0.reshape(-1)
    assert (flat[kind == 0] == c40[0]).all()
    assert (flat[kind == 1] == c17[0]).all()
    assert nre >= int((kind <= 1).sum())
