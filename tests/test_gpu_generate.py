"""GPU parity of the incremental generation path: WaveNet.initialize / generate
(modules.py:232-255 and the block queues 58-74, 98-110) and the device-resident sampling loop of
generate.py:101-145, against the NumPy oracle and against the training forward."""
import numpy as np
import pytest

import helpers as H
import vqvae_oracle as O
from helpers import assert_close, assert_close_scaled

pytestmark = pytest.mark.gpu


def _random_biases(seed):
    def tweak(P):
        rs = np.random.RandomState(seed)
        dec = P['decoder']
        for name in ('embed', 'proj1', 'proj2'):
            dec[name] = (dec[name][0], (0.1 * rs.standard_normal(dec[name][1].shape)).astype(np.float32))
        for blk in dec['blocks']:
            for name in blk:
                blk[name] = (blk[name][0], (0.1 * rs.standard_normal(blk[name][1].shape)).astype(np.float32))
    return tweak


def _decoder(cfg, seed, use_logistic=False):
    P, model = H.build_model(cfg, seed=seed, use_logistic=use_logistic, tweak=_random_biases(seed + 1))
    model.to_gpu()
    return P['decoder'], model.decoder


def _cond(cfg, n, T, seed):
    rs = np.random.RandomState(seed)
    return rs.standard_normal((n, cfg['local_dim'] + cfg['global_dim'], T)).astype(np.float32)


def test_generate_step_api_matches_oracle(gpu):
    """decoder.initialize(n); decoder.generate(x, condition[:, :, i:i+1]) as generate.py:100-109
    drives it: the host owns the input vector and the condition column."""
    from vqvae_amd.core import Variable
    cfg = dict(H.SMALL)
    p, dec = _decoder(cfg, 21)
    n, T = 2, 14
    cond = _cond(cfg, n, T, 5)
    rs = np.random.RandomState(9)
    st = O.wavenet_initialize(p, n, cfg['n_loop'], cfg['n_layer'])
    dec.initialize(n)
    x = np.zeros((n, cfg['input_dim'], 1), np.float32)                 # generate.py:52
    for i in range(T):
        want = O.wavenet_generate_step(p, st, x, cond[:, :, i:i + 1], cfg['n_loop'], cfg['n_layer'])
        got = dec.generate(Variable(gpu.to_device(x[..., None])),
                           Variable(gpu.to_device(np.ascontiguousarray(cond[:, :, i:i + 1])[..., None])))
        assert got.shape == (n, cfg['out_dim'], 1, 1)
        assert_close(got.data.get()[..., 0], want, 1e-4, 'step %d logits' % i)
        x = np.zeros((n, cfg['input_dim'], 1), np.float32)
        for b in range(n):
            x[b, rs.randint(cfg['input_dim']), 0] = 1


@pytest.mark.parametrize('persistent', [False, True])
def test_generate_sequence_teacher_forced(gpu, persistent):
    """Device-resident loop with the fed-back value forced: logits of every step against the
    oracle (1e-4), sampled bins bit-exact wherever the uniform is not within 1e-5 of a cdf edge."""
    cfg = dict(H.SMALL)
    p, dec = _decoder(cfg, 22)
    n, T = 2, 41
    cond = _cond(cfg, n, T, 6)
    rs = np.random.RandomState(10)
    forced = rs.randint(0, cfg['input_dim'], (T, n)).astype(np.int32)
    forced[7, 1] = -1
    u = rs.random_sample((T, n))
    want_out, want_logits = O.wavenet_generate(p, cond, u, cfg['n_loop'], cfg['n_layer'], forced=forced)
    out, logits = dec.generate_sequence(gpu.to_device(cond), u, forced=forced, return_logits=True,
                                        graph_steps=5, persistent=persistent)
    assert_close(logits.get(), want_logits, 1e-4, 'teacher-forced logits')
    got = out.get()
    assert got.shape == (n, T) and got.dtype == np.int32 and (got[:, -1] == 0).all()
    for i in range(T - 1):
        for b in range(n):
            cdf = O.softmax_axis1(want_logits[i, b:b + 1])[0].astype(np.float64).cumsum()
            cdf /= cdf[-1]
            if np.abs(cdf - u[i, b]).min() > 1e-5:
                assert got[b, i] == want_out[b, i], (i, b)


def test_generate_sequence_free_running_equals_oracle(gpu):
    """The full autoregressive loop of generate.py:105-145: every sampled bin equals the oracle's
    (same uniforms), for graphs of 1 and 8 steps (identical results)."""
    cfg = dict(H.SMALL)
    p, dec = _decoder(cfg, 23)
    T = 96
    cond = _cond(cfg, 1, T, 7)
    u = np.random.RandomState(12).random_sample((T, 1))
    want_out, _ = O.wavenet_generate(p, cond, u, cfg['n_loop'], cfg['n_layer'])
    got8 = dec.generate_sequence(gpu.to_device(cond), u, graph_steps=8, persistent=False).get()
    got1 = dec.generate_sequence(gpu.to_device(cond), u, graph_steps=1, persistent=False).get()
    got0 = dec.generate_sequence(gpu.to_device(cond), u, graph_steps=0, persistent=False).get()
    gotp = dec.generate_sequence(gpu.to_device(cond), u, persistent=True).get()
    np.testing.assert_array_equal(got8, got1)
    np.testing.assert_array_equal(got8, got0)
    np.testing.assert_array_equal(got8, want_out)
    np.testing.assert_array_equal(gotp, want_out)
    # the persistent launch continued in chunks (queues, mailboxes and feedback survive the seam)
    dec.initialize(1)
    gotc = dec._gen.run(gpu.to_device(cond), u, 1, persistent=True, chunk=7).get()
    np.testing.assert_array_equal(gotc, want_out)
    assert len(np.unique(got8)) > 10                                   # not a degenerate stream


@pytest.mark.parametrize('persistent', [False, True])
def test_generate_partial_steps_and_restart(gpu, persistent):
    cfg = dict(H.SMALL)
    p, dec = _decoder(cfg, 24)
    T = 30
    cond = _cond(cfg, 2, T, 8)
    u = np.random.RandomState(13).random_sample((T, 2))
    full = dec.generate_sequence(gpu.to_device(cond), u, persistent=persistent).get()
    part = dec.generate_sequence(gpu.to_device(cond), u, n_steps=11, persistent=persistent).get()
    np.testing.assert_array_equal(part[:, :11], full[:, :11])
    assert (part[:, 11:] == 0).all()
    st = dec._gen
    with pytest.raises(RuntimeError):                                  # not fresh any more
        st.run(gpu.to_device(cond), u, 1)


@pytest.mark.parametrize('persistent', [False, True])
def test_mol_generation_matches_oracle(gpu, persistent):
    """configs[4] output: generate.py:113-133 (softmax-weighted logistic samples, /127.5, clip),
    scalar feedback into the 1-channel embed."""
    cfg = dict(H.MOL)
    p, dec = _decoder(cfg, 25, use_logistic=True)
    n, T = 2, 200
    cond = _cond(cfg, n, T, 9)
    rs = np.random.RandomState(14)
    u = rs.uniform(0.02, 0.98, (T, n, 10))
    forced = rs.uniform(-1, 1, (T, n)).astype(np.float32)
    want_out, want_logits = O.wavenet_generate(p, cond, u, cfg['n_loop'], cfg['n_layer'], loss_kind='mol',
                                               forced=forced)
    out, logits = dec.generate_sequence(gpu.to_device(cond), u, forced=forced, return_logits=True,
                                        persistent=persistent)
    assert_close(logits.get(), want_logits, 1e-4, 'mol logits')
    got = out.get()
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, want_out, rtol=0, atol=2e-5)
    # free running: the fed-back value is continuous, deviations stay at rounding level
    want_out, _ = O.wavenet_generate(p, cond, u, cfg['n_loop'], cfg['n_layer'], loss_kind='mol', n_steps=48)
    got = dec.generate_sequence(gpu.to_device(cond), u, n_steps=48, persistent=persistent).get()
    np.testing.assert_allclose(got, want_out, rtol=0, atol=1e-3)
    assert np.abs(got).max() <= 1.0 and np.abs(got[:, :48]).max() > 0


@pytest.mark.parametrize('persistent', [False, True])
def test_generation_equals_training_forward_full_size(gpu, persistent):
    """Size-independent property at the BASELINE configs[1] decoder (20 blocks, 256 channels,
    dilations to 512): with the same inputs, step i of the incremental path equals column i of the
    training forward (modules.py:148-160) -- queues of every dilation wrap at least twice."""
    from vqvae_amd.core import Variable
    cfg = dict(d=64, k=512, n_loop=2, n_layer=10, filter_size=2, input_dim=256, residual=256,
               dilated=256, skip=256, out_dim=256, local_dim=64, global_dim=128, n_speaker=5)
    p, dec = _decoder(cfg, 26)
    n, T = 1, 1200
    cond = _cond(cfg, n, T, 10)
    rs = np.random.RandomState(15)
    forced = rs.randint(0, 256, (T, n)).astype(np.int32)
    u = rs.random_sample((T, n))
    out, logits = dec.generate_sequence(gpu.to_device(cond), u, forced=forced, return_logits=True,
                                        persistent=persistent)
    x = np.zeros((n, 256, T), np.float32)
    for i in range(T - 1):
        x[0, forced[i, 0], i + 1] = 1
    y = dec(Variable(gpu.to_device(x[..., None])), Variable(gpu.to_device(cond[..., None]))).data.get()
    y = y.reshape(n, 256, T).transpose(2, 0, 1)[:T - 1]
    assert_close_scaled(logits.get(), y, 1e-4, 'incremental vs training forward')


def test_generation_argument_errors(gpu):
    from vqvae_amd.core import Variable
    cfg = dict(H.SMALL)
    p, dec = _decoder(cfg, 27)
    dec._gen = None
    with pytest.raises(RuntimeError):
        dec.generate(Variable(gpu.zeros((1, 256, 1, 1))), Variable(gpu.zeros((1, 64, 1, 1))))
    with pytest.raises(ValueError):
        dec.initialize(5)                                              # more than 4 lockstep sequences
    dec.initialize(1)
    with pytest.raises(ValueError):
        dec.generate(Variable(gpu.zeros((1, 255, 1, 1))), Variable(gpu.zeros((1, 64, 1, 1))))
    with pytest.raises(ValueError):
        dec.generate(Variable(gpu.zeros((1, 256, 1, 1))), Variable(gpu.zeros((1, 63, 1, 1))))
    with pytest.raises(ValueError):
        dec.generate(np.zeros((1, 256, 1, 1), np.float32), gpu.zeros((1, 64, 1, 1)))   # host array
    with pytest.raises(ValueError):
        dec.generate_sequence(gpu.zeros((1, 64, 8)), np.zeros(3))      # too few uniforms


def test_synthesize_matches_oracle_pipeline(gpu):
    """generate.py:94-149 end to end: encoder -> VQ -> condition embed (other speaker) -> sampling
    loop -> mu-law expansion, against the same pipeline composed from the oracle's parts."""
    import vqvae_amd as V
    cfg = dict(H.SMALL)
    P, model = H.build_model(cfg, seed=31, tweak=_random_biases(32))
    model.to_gpu()
    x_enc, _, spk, _ = O.synth_batch(1, length=128, n_speaker=cfg['n_speaker'], seed=5)
    speaker = np.array([(int(spk[0]) + 1) % cfg['n_speaker']], np.int32)        # convert to another voice
    z, _ = O.encoder_fwd(P['encoder'], x_enc)
    e4, idx = O.vq_forward(O.expand4(z), P['vq'])
    cond, _ = O.cond_embed_fwd(P['condition_embed'], np.ascontiguousarray(O.squeeze4(e4)), speaker)
    T = cond.shape[2]
    u = np.random.RandomState(77).random_sample((T - 1, 1))
    want_out, _ = O.wavenet_generate(P['decoder'], cond, u, cfg['n_loop'], cfg['n_layer'])
    want_wave = O.MuLaw(256).itransform(want_out)
    wave, out = V.synthesize(model.encoder, model.vq, model.decoder, model.condition_embed,
                             x_enc[..., None], speaker, rng=np.random.RandomState(77))
    assert T == 128 and out.shape == (1, T)
    np.testing.assert_array_equal(out, want_out)
    np.testing.assert_allclose(wave, want_wave, rtol=0, atol=1e-7)
    assert wave.dtype == want_wave.dtype


def test_generate_batch_equals_individual_sequences(gpu):
    """Concurrent groups on separate streams: each of the 7 sequences gets exactly what it gets alone."""
    cfg = dict(H.SMALL)
    p, dec = _decoder(cfg, 41)
    N, T = 7, 40
    cond = _cond(cfg, N, T, 17)
    u = np.random.RandomState(18).random_sample((T, N))
    got = dec.generate_batch(gpu.to_device(cond), u, group=3, max_streams=2)
    assert got.shape == (N, T)
    for i in range(N):
        alone = dec.generate_sequence(gpu.to_device(cond[i:i + 1]), u[:, i:i + 1]).get()
        np.testing.assert_array_equal(got[i:i + 1], alone)


def test_wide_model_runs_on_the_per_step_kernels(gpu):
    """More than 256 residual channels exceed the persistent kernel's per-vector limit: the same call
    silently uses the per-step kernels and still matches the oracle."""
    cfg = dict(H.SMALL, residual=320, skip=288)
    p, dec = _decoder(cfg, 51)
    T = 24
    cond = _cond(cfg, 1, T, 19)
    rs = np.random.RandomState(20)
    forced = rs.randint(0, cfg['input_dim'], (T, 1)).astype(np.int32)
    u = rs.random_sample((T, 1))
    want_out, want_logits = O.wavenet_generate(p, cond, u, cfg['n_loop'], cfg['n_layer'], forced=forced)
    out, logits = dec.generate_sequence(gpu.to_device(cond), u, forced=forced, return_logits=True)
    assert_close(logits.get(), want_logits, 1e-4, 'wide model logits')
