"""The kernel INSTANTIATIONS bench.py measures, compared with the oracle directly.

launch_gemm picks its tile shape from the launch size: the 256 x 256-column `conv_gemm_x3_kernel<...,
NB = 2, ...>` (gate kernel, backward-data of the dilated conv) only runs when a launch has >= 256 such
tiles, i.e. B >= 9 at T = 7680, and the streaming `lin128_stream_kernel` (the K = 128 residual
projection) / two-tap kernels are chosen from the channel counts of configs[1].  Every
oracle-compared case of test_gpu_kernels.py has B <= 3, so until round 3 those instantiations were
pinned only transitively (batch independence, shard sums).  Here: ResidualBlock.__call__
(modules.py:30-56) and the ResidualNet chain (modules.py:89-96) at B = 16, T = 7680, 256 channels
against oracle.resblock_fwd / resblock_bwd, and one whole configs[1] step (B = 16) against
oracle.train_step (updaters.py:6-19)."""
import numpy as np
import pytest

import helpers as H
import vqvae_oracle as O
from helpers import assert_close, assert_close_scaled, to4
from test_gpu_kernels import _rb_params

pytestmark = pytest.mark.gpu

B, T = 16, 7680


def _dev(gpu, a):
    return gpu.to_device(np.ascontiguousarray(a))


@pytest.mark.parametrize('dil', [1, 512])
def test_resblock_b16_vs_oracle(gpu, matmul_mode, dil):
    """One ResidualBlock at the benchmarked launch shape (B = 16 -> 480 256 x 256 tiles: the NB = 2,
    TAP2 gate and backward-data kernels), dilation 1 (both taps inside one tile's window) and 512
    (taps two tiles apart), forward and backward, 1e-4."""
    from vqvae_amd.core import Variable
    from vqvae_amd.wavenet import ResidualBlockFunction
    rs = np.random.RandomState(100 + dil)
    p = _rb_params(rs, 256, 256, 256, 192, 2)
    x = rs.standard_normal((B, 256, T)).astype(np.float32)
    c = rs.standard_normal((B, 192, T)).astype(np.float32)
    res_ref, skip_ref, cache = O.resblock_fwd(p, x, c, dil)
    g_res = rs.standard_normal(res_ref.shape).astype(np.float32)
    g_skip = rs.standard_normal(skip_ref.shape).astype(np.float32)
    gx_ref, gc_ref, gr = O.resblock_bwd(p, cache, c, dil, g_res, g_skip)
    order = ['conv', 'condition_proj', 'res', 'skip']
    vs = [Variable(_dev(gpu, to4(x))), Variable(_dev(gpu, to4(c)))]
    for n in order:
        vs += [Variable(_dev(gpu, to4(p[n][0]))), Variable(_dev(gpu, p[n][1]))]
    res, skip = ResidualBlockFunction(dil).apply(vs)
    assert_close(res.data.get(), res_ref, 1e-4, 'res')
    assert_close(skip.data.get(), skip_ref, 1e-4, 'skip')
    gouts = res.creator.backward(tuple(range(10)), (Variable(_dev(gpu, to4(g_res))), Variable(_dev(gpu, to4(g_skip)))))
    assert_close_scaled(gouts[0].get(), gx_ref, 1e-4, 'gx')
    assert_close_scaled(gouts[1].get(), gc_ref, 1e-4, 'gcond')
    for i, n in enumerate(order):
        assert_close_scaled(gouts[2 + 2 * i].get(), gr[n][0], 1e-4, 'gW ' + n)
        assert_close_scaled(gouts[3 + 2 * i].get(), gr[n][1], 1e-4, 'gb ' + n)


def test_resstack_b16_vs_oracle(gpu, matmul_mode):
    """ResidualNet's chain as bench.py runs it: ResidualStackFunction over a LAZY (latent-rate)
    condition -- condition projection at the latent rate + epilogue lerp, the streaming residual
    1x1 (`res` conv alone, K = 128 -> 256 rows, + x), the skip sum as one GEMM, batched weight
    gradients, the latent pull-back -- three blocks (dilations 1, 2, 512) at B = 16, T = 7680 against
    the oracle's per-block restatement fed the materialised (B, 192, T) condition."""
    from vqvae_amd import functions as F
    from vqvae_amd.core import Variable
    from vqvae_amd.wavenet import ResidualStackFunction
    dils = [1, 2, 512]
    Tl, Cl, G, nspk = T // 64, 64, 128, 7
    rs = np.random.RandomState(7)
    blocks = [_rb_params(rs, 256, 256, 256, Cl + G, 2) for _ in dils]
    x = rs.standard_normal((B, 256, T)).astype(np.float32)
    local = rs.standard_normal((B, Cl, Tl)).astype(np.float32)
    E = rs.standard_normal((nspk, G)).astype(np.float32)
    ids = rs.randint(0, nspk, B).astype(np.int32)
    gy = rs.standard_normal((B, 256, T)).astype(np.float32)
    # oracle: full-rate condition (net.py:54-63), block by block (modules.py:89-96)
    cond = np.concatenate([O.upsample_fwd(local, T), np.repeat(E[ids][:, :, None], T, axis=2)], axis=1).astype(np.float32)
    h, caches, skip_ref = x, [], None
    for blk, d in zip(blocks, dils):
        h, sk, cch = O.resblock_fwd(blk, h, cond, d)
        caches.append(cch)
        skip_ref = sk if skip_ref is None else skip_ref + sk
    g_res, gcond_ref, bg = None, np.zeros_like(cond), [None] * len(dils)
    for i in range(len(dils) - 1, -1, -1):
        g_res, gc, bg[i] = O.resblock_bwd(blocks[i], caches[i], cond, dils[i], g_res, gy)
        gcond_ref += gc
    glocal_ref = O.upsample_bwd(gcond_ref[:, :Cl], Tl)
    gE_ref = np.zeros_like(E, dtype=np.float64)
    np.add.at(gE_ref, ids, gcond_ref[:, Cl:].sum(axis=2, dtype=np.float64))
    # device
    vx = Variable(_dev(gpu, to4(x)))
    vlocal, vE = Variable(_dev(gpu, to4(local))), Variable(_dev(gpu, E))
    vcond = F.condition_assemble(vlocal, vE, _dev(gpu, ids), 64)
    assert isinstance(vcond.data, F.LazyUpsampled)
    order = ['conv', 'condition_proj', 'res', 'skip']
    pv = []
    for blk in blocks:
        for n in order:
            pv += [Variable(_dev(gpu, to4(blk[n][0]))), Variable(_dev(gpu, blk[n][1]))]
    skip = ResidualStackFunction(dils).apply([vx, vcond] + pv)[0]
    assert_close(skip.data.get()[..., 0], skip_ref, 1e-4, 'skip_connections')
    skip.grad = _dev(gpu, to4(gy))
    skip.backward()
    assert_close_scaled(vx.grad.get()[..., 0], g_res, 1e-4, 'gx')
    assert_close_scaled(vlocal.grad.get()[..., 0], glocal_ref, 1e-4, 'g local condition')
    assert_close_scaled(vE.grad.get(), gE_ref, 1e-4, 'g speaker embedding')
    last = len(dils) - 1
    for i in range(len(dils)):
        for j, n in enumerate(order):
            gW, gb = pv[8 * i + 2 * j].grad, pv[8 * i + 2 * j + 1].grad
            if i == last and n == 'res':          # unused residual branch of the last block (modules.py:89-96)
                assert gW is None and gb is None
                continue
            assert_close_scaled(gW.get().reshape(bg[i][n][0].shape), bg[i][n][0], 1e-4, 'block %d gW %s' % (i, n))
            assert_close_scaled(gb.get(), bg[i][n][1], 1e-4, 'block %d gb %s' % (i, n))


def test_presplit_storage_changes_only_the_rounding_point(gpu):
    """'float32x2', PRE-SPLIT storage (csrc/gemm_common.h; vqvae_resblock_desc.storage & VQVAE_STORE_*_F16X2): gh_l and
    the residual stream x_l are written by their producers as fp16 hi | lo dwords under an a-priori bound, and their
    readers stage them with two permutes per element pair instead of splitting them again.  Against the same chain with
    every tensor fp32 (backend.set_presplit(0), round 4's form) the only difference is WHERE the split is rounded: every
    output and gradient of a four-block stack (B = 16, T = 7680, dilations 1, 2, 256, 512) agrees to 2e-6 of its scale
    -- fifty times inside the 1e-4 parity bar that test_resstack_b16_vs_oracle holds both forms to -- for gh alone
    (mask 1), the stream alone (mask 2) and both (mask 3).  Mask bit 2 (VQVAE_STORE_GATES_SIG): the gate kernel saves sigmoid and
    z = tanh * sigmoid only and the backward takes tanh = z / sigmoid -- alone (4) and with the rest (7, the default)."""
    from vqvae_amd import functions as F
    from vqvae_amd.core import Variable
    from vqvae_amd.wavenet import ResidualStackFunction
    gpu.set_matmul_dtype('float32x2')
    dils = [1, 2, 256, 512]
    Tl, Cl, G, nspk = T // 64, 64, 128, 7
    rs = np.random.RandomState(11)
    blocks = [_rb_params(rs, 256, 256, 256, Cl + G, 2) for _ in dils]
    x = rs.standard_normal((B, 256, T)).astype(np.float32)
    local = rs.standard_normal((B, Cl, Tl)).astype(np.float32)
    E = rs.standard_normal((nspk, G)).astype(np.float32)
    ids = rs.randint(0, nspk, B).astype(np.int32)
    gy = rs.standard_normal((B, 256, T)).astype(np.float32)
    order = ['conv', 'condition_proj', 'res', 'skip']

    def run(mask, fuse_pullback=True):
        import vqvae_amd.wavenet as wn
        gpu.set_presplit(mask)
        wn.FUSE_PULLBACK = fuse_pullback
        vx = Variable(_dev(gpu, to4(x)))
        vlocal, vE = Variable(_dev(gpu, to4(local))), Variable(_dev(gpu, E))
        vcond = F.condition_assemble(vlocal, vE, _dev(gpu, ids), 64)
        pv = []
        for blk in blocks:
            for n in order:
                pv += [Variable(_dev(gpu, to4(blk[n][0]))), Variable(_dev(gpu, blk[n][1]))]
        fn = ResidualStackFunction(dils)
        skip = fn.apply([vx, vcond] + pv)[0]
        used = [d.storage for d in fn.descs]
        out = {'skip': skip.data.get()}
        skip.grad = _dev(gpu, to4(gy))
        skip.backward()
        used = [u | d.storage for u, d in zip(used, fn.descs)]
        out['gx'] = vx.grad.get()
        out['glocal'] = vlocal.grad.get()
        out['gE'] = vE.grad.get()
        for i, v in enumerate(pv):
            if v.grad is not None:
                out['p%d' % i] = v.grad.get()
        return out, used
    try:
        ref, used0 = run(0)
        assert all(u == 0 for u in used0)
        for mask in (1, 2, 3, 4, 7):
            got, used = run(mask)
            assert all(bool(u & 256) == bool(mask & 4) for u in used), used      # VQVAE_STORE_GATES_SIG on every block
            if mask & 1:
                assert all(u & 32 for u in used), used                      # VQVAE_STORE_GH_F16X2 on every block
            if mask & 2:
                assert used[0] & 128 and not used[0] & 64 and used[-1] & 64 and not used[-1] & 128 and all(u & 192 == 192 for u in used[1:-1]), used
            assert set(got) == set(ref)
            for k in ref:
                assert_close_scaled(got[k], ref[k], 2e-6, 'mask %d: %s' % (mask, k))
        # the latent pull-back inside the gate-derivative launch (vqvae_resblock_amax.pb_part + vqvae_pullback_reduce) against a
        # vqvae_upsample_linear_bwd* launch per block: the same sums in another order
        unfused, _ = run(7, fuse_pullback=False)
        for k in ref:
            assert_close_scaled(got[k], unfused[k], 2e-6, 'fused pull-back: %s' % k)
    finally:
        import vqvae_amd.wavenet as wn
        wn.FUSE_PULLBACK = True
        gpu.set_presplit(7)
        gpu.set_matmul_dtype(gpu.default_matmul_dtype())


@pytest.mark.parametrize('nblocks,ratio', [(3, 1e-4), (20, 1e-3), (20, 1e-4)])
def test_presplit_chain_keeps_a_quiet_sample(gpu, nblocks, ratio):
    """The dynamic-range contract of 'float32x2' (test_gpu_kernels.py::test_float32x2_dynamic_range_contract) through
    ResidualNet's chain with its PRE-SPLIT tensors: x_l and gh_l are split under a-priori BOUNDS (max |x_l| + the res
    conv's row norm; the column norms of Wr / Ws times max |g_res| / max |g_skip|), which sit up to 2^6 above the true
    maxima, so the absolute floor under a quiet sample is that much higher than for a tensor split under its own maximum:
    2 * K * 2^-33 * max|x| * max|W| per contraction.  A B = 4 batch whose last sample is `ratio` of the others in x and in
    the output gradient still gets that sample's forward output and input gradient to 1e-4 of ITS OWN scale against the
    oracle -- through three blocks (dilations 1, 2, 4) and through the configs' full depth of 20 (dilations 1 .. 512,
    twice: the stored stream is re-split twenty times in a row and the gradient stream crosses twenty backward-data
    GEMMs); T = 2048.  The guard (backend.f32x2_contract_violations) must stay silent: every bound within 2^8."""
    from vqvae_amd import functions as F
    from vqvae_amd.core import Variable
    from vqvae_amd.wavenet import ResidualStackFunction
    gpu.set_matmul_dtype('float32x2')
    try:
        Bq, Tq = 4, 2048
        dils = [1, 2, 4] if nblocks == 3 else [2 ** (i % 10) for i in range(nblocks)]
        gpu.f32x2_contract_violations(reset=True)
        Tl, Cl, G, nspk = Tq // 64, 64, 128, 7
        rs = np.random.RandomState(23)
        blocks = [_rb_params(rs, 256, 256, 256, Cl + G, 2) for _ in dils]
        x = rs.standard_normal((Bq, 256, Tq)).astype(np.float32)
        x[-1] *= ratio
        local = rs.standard_normal((Bq, Cl, Tl)).astype(np.float32)
        E = rs.standard_normal((nspk, G)).astype(np.float32)
        ids = rs.randint(0, nspk, Bq).astype(np.int32)
        gy = rs.standard_normal((Bq, 256, Tq)).astype(np.float32)
        gy[-1] *= ratio
        cond = np.concatenate([O.upsample_fwd(local, Tq), np.repeat(E[ids][:, :, None], Tq, axis=2)], axis=1).astype(np.float32)
        h, caches, skip_ref = x, [], None
        for blk, d in zip(blocks, dils):
            h, sk, cch = O.resblock_fwd(blk, h, cond, d)
            caches.append(cch)
            skip_ref = sk if skip_ref is None else skip_ref + sk
        g_res = None
        for i in range(len(dils) - 1, -1, -1):
            g_res, _, _ = O.resblock_bwd(blocks[i], caches[i], cond, dils[i], g_res, gy)
        vx = Variable(_dev(gpu, to4(x)))
        vlocal, vE = Variable(_dev(gpu, to4(local))), Variable(_dev(gpu, E))
        vcond = F.condition_assemble(vlocal, vE, _dev(gpu, ids), 64)
        pv = []
        for blk in blocks:
            for n in ['conv', 'condition_proj', 'res', 'skip']:
                pv += [Variable(_dev(gpu, to4(blk[n][0]))), Variable(_dev(gpu, blk[n][1]))]
        fn = ResidualStackFunction(dils)
        skip = fn.apply([vx, vcond] + pv)[0]
        assert all(d.storage & 192 for d in fn.descs), 'the chain did not take the pre-split stream'
        assert_close(skip.data.get()[..., 0], skip_ref, 1e-4, 'skip_connections')
        skip.grad = _dev(gpu, to4(gy))
        skip.backward()
        assert all(d.storage & 32 for d in fn.descs), 'the chain did not keep gh pre-split'
        gx = vx.grad.get()[..., 0]
        assert_close_scaled(gx, g_res, 1e-4, 'gx')
        rel = np.abs(gx[-1] - g_res[-1]).max() / np.abs(g_res[-1]).max()
        sk = skip.data.get()[..., 0]
        rel_s = np.abs(sk[-1] - skip_ref[-1]).max() / np.abs(skip_ref[-1]).max()
        rep = gpu.f32x2_contract_violations()
        print('quiet sample (%g of the batch) through %d blocks: gx %.2e, skip %.2e of its own scale; loosest bound 2^%.1f over %d pre-split tensors'
              % (ratio, nblocks, rel, rel_s, rep['worst_log2'], rep['checked']))
        assert rel <= 1e-4, 'gx of the quiet sample: %.2e of its own scale' % rel
        assert rel_s <= 1e-4, 'skip of the quiet sample: %.2e of its own scale' % rel_s
        assert rep['violations'] == 0 and rep['checked'] == 2 * nblocks - 1, rep
    finally:
        gpu.set_matmul_dtype(gpu.default_matmul_dtype())


def test_config1_whole_step_matches_oracle(gpu):
    """BASELINE configs[1] as configured (batch 16, length 7680, d=64 k=512, 20 blocks, 256 channels,
    EMA on): one VQVAE_StandardUpdater.update() against oracle.train_step -- 1 920 argmin indices
    bit-exact, losses 1e-4, every gradient 1e-4 of its scale (the assertion's bar; the worst tensor is printed), every
    parameter after Adam 1e-4.  The B = 1 twin of this test is test_gpu_configs.py::test_config0_..."""
    import copy
    import vqvae_amd as V
    from vqvae_amd.optimizers import Adam
    from test_gpu_configs import CFG0, _limit_blas
    from test_gpu_model import _Iter, _grads_by_name
    cfg = dict(CFG0)
    P, model = H.build_model(cfg, seed=1, ema_decay=0.9999)
    P_ema = copy.deepcopy(P['decoder'])
    model.to_gpu()
    opt = Adam(2e-4)
    opt.setup(model)
    batch = O.synth_batch(B, length=T, n_speaker=cfg['n_speaker'], seed=73)
    upd = V.VQVAE_StandardUpdater(_Iter([batch]), opt, device=0)
    sites = H.device_relu_sites(model, batch[0], batch[1], batch[2])
    upd.update()
    ks = {}
    with _limit_blas():
        losses, cache, G, flips = H.oracle_train_step_aligned(P, {}, batch, cfg['n_loop'], cfg['n_layer'], sites,
                                                              ema=P_ema, ema_decay=0.9999, kink_stats=ks)
    # the count is held against the noise model's own prediction (helpers.align_relu_kinks), not a round number
    print('configs[1]: %d ReLU kink elements (of ~90 M) took the other side on the device; the noise model predicts %.1f, allows %d'
          % (flips, ks['expected'], H.kink_flip_ceiling(ks)))
    assert flips <= H.kink_flip_ceiling(ks), ks['per_site']
    idx_dev = model.vq._cache[3][0].get()
    np.testing.assert_array_equal(idx_dev.reshape(cache['idx'].shape), cache['idx'])
    assert cache['idx'].size == 16 * 120
    l_dev = [float(l.data.get()) for l in upd.last_losses]
    for i, (a, b) in enumerate(zip(l_dev, losses)):
        assert_close(a, float(b), 1e-4, 'configs[1] loss%d' % (i + 1))
    g_dev = _grads_by_name(model, opt, True)
    over = []
    for name, arr in G.items():
        dn = H._dev_name(name, True)
        got = g_dev[dn].reshape(arr.shape).astype(np.float64)
        err = float(np.abs(got - arr).max() / max(np.abs(arr).max(), 1e-30))
        over.append((err, dn))
        assert err <= 1e-4, 'configs[1] grad %s: %.3e of scale (north_star: 1e-4)' % (dn, err)
    print('configs[1]: worst of %d gradient tensors, of scale: %s' % (len(G), sorted(over, reverse=True)[:4]))
    named = dict(model.namedparams())
    for name, arr in O.flatten_params(P):
        dn = H._dev_name(name, True)
        assert_close(named[dn].data.get().reshape(arr.shape), arr, 1e-4, 'configs[1] param ' + dn)


def test_pack_once_equals_pack_per_call(gpu):
    """ResidualNet packs the weight slabs of all its blocks once per step (vqvae_resstack_pack + the _packed entry
    points); the stand-alone entry points pack inside every call.  Same slabs, same kernels: the skip output and every
    gradient must agree bit for bit ('float32x3'; in 'float32x2' the two forms run different fp32-accurate kernels: 1e-5)."""
    from vqvae_amd import functions as F, wavenet
    from vqvae_amd.core import Variable
    from vqvae_amd.wavenet import ResidualStackFunction
    dils, Bq, Tq, Cl, G, nspk = [1, 2, 4], 2, 512, 16, 16, 3
    rs = np.random.RandomState(11)
    blocks = [_rb_params(rs, 64, 64, 64, Cl + G, 2) for _ in dils]
    x = rs.standard_normal((Bq, 64, Tq)).astype(np.float32)
    local = rs.standard_normal((Bq, Cl, Tq // 64)).astype(np.float32)
    E = rs.standard_normal((nspk, G)).astype(np.float32)
    ids = rs.randint(0, nspk, Bq).astype(np.int32)
    gy = rs.standard_normal((Bq, 64, Tq)).astype(np.float32)
    order = ['conv', 'condition_proj', 'res', 'skip']

    def run(pack_once):
        old = wavenet.PACK_ONCE
        wavenet.PACK_ONCE = pack_once
        try:
            vx = Variable(_dev(gpu, to4(x)))
            vcond = F.condition_assemble(Variable(_dev(gpu, to4(local))), Variable(_dev(gpu, E)), _dev(gpu, ids), 64)
            pv = []
            for blk in blocks:
                for n in order:
                    pv += [Variable(_dev(gpu, to4(blk[n][0]))), Variable(_dev(gpu, blk[n][1]))]
            skip = ResidualStackFunction(dils).apply([vx, vcond] + pv)[0]
            out = [skip.data.get()]
            skip.grad = _dev(gpu, to4(gy))
            skip.backward()
            out.append(vx.grad.get())
            out += [v.grad.get() for v in pv if v.grad is not None]
            return out
        finally:
            wavenet.PACK_ONCE = old
    try:
        for mode in ('float32x3', 'float32x2'):
            gpu.set_matmul_dtype(mode)
            a, b = run(True), run(False)
            assert len(a) == len(b) and len(a) > 20
            for u, v in zip(a, b):
                if mode == 'float32x3':
                    np.testing.assert_array_equal(u, v)
                else:
                    # 'float32x2': the packed chain runs the three-product kernels (the tensors' maxima travel with
                    # them), the pack-per-call entry points mode 2's: two fp32-accurate evaluations of one chain
                    assert np.abs(u - v).max() <= 1e-5 * max(np.abs(v).max(), 1e-30)
    finally:
        gpu.set_matmul_dtype(gpu.default_matmul_dtype())


_WIN_WORKER = r"""
import hashlib, os, sys
import numpy as np
sys.path[:0] = [os.path.join(sys.argv[1], 'chainer-vq-vae_amd')]
from vqvae_amd import backend as gpu, functions as F
from vqvae_amd.core import Variable
gpu.init(0)
rs = np.random.RandomState(5)
B, C, T, dil = 9, 256, 7680, int(sys.argv[2])
x = Variable(gpu.to_device(rs.standard_normal((B, C, T, 1)).astype(np.float32)))
W = Variable(gpu.to_device((rs.standard_normal((C, C, 2, 1)) / 22).astype(np.float32)))
b = Variable(gpu.to_device(rs.standard_normal(C).astype(np.float32)))
y = F.convolution_1d(x, W, b, pad=dil, dilate=dil, out_len=T)
y.grad = gpu.to_device(rs.standard_normal((B, C, T, 1)).astype(np.float32))
y.backward()
h = hashlib.sha256()
h.update(y.data.get().tobytes()); h.update(x.grad.get().tobytes())
print('HASH', h.hexdigest())
"""


@pytest.mark.parametrize('dil', [1, 64])
def test_two_tap_kernels_agree_bitwise(gpu, tmp_path, dil):
    """The two kernels a two-tap contraction of a 256-row GEMM can run on -- the 256 x 128-tile loop with two workgroups
    per CU (the default wherever an operand has two or more pieces) and the tap-interleaved 256 x 256-tile
    conv_gemm_x3_kernel (VQVAE_X3_LEAN=0, kept as the A/B alternate) -- keep one K order and one product order:
    forward and backward-data of the dilated conv at B = 9, T = 7680 must not differ in a single bit whichever the
    launch picks, in the default matmul mode and in 'float32x3' (the selecting switch is read once per process: one
    subprocess each).  (Round 3's third kernel, one staged window for both taps, measured neutral and was removed.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'win_worker.py'
    script.write_text(_WIN_WORKER)
    for mode in ('float32x2', 'float32x3'):
        out = []
        for lean in ('1', '0'):
            env = dict(os.environ, VQVAE_X3_LEAN=lean, VQVAE_MATMUL=mode)
            r = subprocess.run([sys.executable, str(script), root, str(dil)], env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=600)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            out.append([l for l in r.stdout.decode().splitlines() if l.startswith('HASH')][0])
        assert out[0] == out[1], mode


_LIN_WORKER = r"""
import hashlib, os, sys
import numpy as np
sys.path[:0] = [os.path.join(sys.argv[1], 'chainer-vq-vae_amd')]
from vqvae_amd import backend as gpu, functions as F
from vqvae_amd.core import Variable, no_backprop_mode
gpu.init(0)
rs = np.random.RandomState(6)
x = Variable(gpu.to_device(rs.standard_normal((3, 128, 7680, 1)).astype(np.float32)))
W = Variable(gpu.to_device((rs.standard_normal((256, 128, 1, 1)) / 11).astype(np.float32)))
b = Variable(gpu.to_device(rs.standard_normal(256).astype(np.float32)))
with no_backprop_mode():
    y = F.convolution_1d(x, W, b)
print('HASH', hashlib.sha256(y.data.get().tobytes()).hexdigest())
"""


def test_streaming_1x1_equals_tiled_kernel_bitwise(gpu, tmp_path):
    """lin128_stream_kernel (weights in registers, persistent) multiplies the same bf16 pieces in the same K order and
    adds the bias in the same place as conv_gemm_x3_kernel: identical bits (VQVAE_LIN128=0 selects the tiled kernel)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'lin_worker.py'
    script.write_text(_LIN_WORKER)
    out = []
    for v in ('32', '0'):
        env = dict(os.environ, VQVAE_LIN128=v)
        r = subprocess.run([sys.executable, str(script), root], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        out.append([l for l in r.stdout.decode().splitlines() if l.startswith('HASH')][0])
    assert out[0] == out[1]
