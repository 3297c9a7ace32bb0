"""CPU tests (-m "not gpu") of the host side: the C-ABI library loads and exports
every symbol include/vqvae_hip.h declares, argument validation, the Chainer-shaped
Link tree (names, shapes, parameter counts), and that NOTHING computes on the host."""
import ctypes as C
import os
import re
import sys
import time

import numpy as np
import pytest

import vqvae_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'vqvae_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(vqvae_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_header_symbol():
    from vqvae_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), 'libvqvae_hip.so does not export ' + s
        assert s in _lib.PROTOTYPES, 'no ctypes prototype for ' + s
    for s in _lib.PROTOTYPES:
        assert s in syms, 'prototype %s is not declared in include/vqvae_hip.h' % s
    assert lib.vqvae_abi_version() == 5 == _lib.ABI_VERSION


def test_graft_entry_library_check():
    """__graft_entry__.build()'s post-compile check, without the compile."""
    import __graft_entry__ as g
    assert g.check_library() is not None


def test_argument_validation_without_device():
    """Entry points validate before touching the device and return VQVAE_E_INVALID."""
    from vqvae_amd import _lib
    lib = _lib.load()
    d = _lib.Conv1dDesc(1, 4, 16, 4, 99, 2, 1, 0, 1, 0)           # Tout > natural length
    assert lib.vqvae_conv1d_fwd(C.byref(d), 8, 8, None, 8, 8, 1 << 20, None) == -1
    assert b'Tout' in lib.vqvae_last_error_string()
    d = _lib.Conv1dDesc(1, 4, 16, 4, 15, 9, 1, 0, 1, 0)           # K too large
    assert lib.vqvae_conv1d_fwd(C.byref(d), 8, 8, None, 8, 8, 1 << 20, None) == -1
    rb = _lib.ResblockDesc(1, 16, 8, 40, 8, 4, 2, 1)              # Cd/2 not a multiple of 32
    assert lib.vqvae_resblock_workspace_bytes(C.byref(rb)) > 0
    assert lib.vqvae_resblock_fwd(C.byref(rb), None, 8, 8, None, 8, 8, 0, 8, 8, 8, 1 << 30, None) == -1
    assert lib.vqvae_vq_nearest_fwd(None, None, 1, 1, 1, 1, 0, None, None, None, None, 0, None) == -1
    with pytest.raises(_lib.HipError):
        _lib.call('vqvae_sum', None, 4, 1.0, None, None, 0, None)


def test_workspace_queries():
    from vqvae_amd import _lib
    lib = _lib.load()
    rb = _lib.ResblockDesc(16, 7680, 256, 256, 256, 192, 2, 512)
    n = lib.vqvae_resblock_workspace_bytes(C.byref(rb))
    assert 16 * 256 * 7680 * 4 < n < 2 * 2 ** 30
    assert lib.vqvae_resstack_workspace_bytes(C.byref(rb), 20) > 0
    assert lib.vqvae_resstack_workspace_bytes(C.byref(rb), 25) == 0     # > 24 blocks unsupported
    assert lib.vqvae_vq_workspace_bytes(16, 64, 120, 512) >= 512 * 64 * 8


def _c1_model():
    import vqvae_amd as V
    from vqvae_amd import functions as F
    enc = V.Encoder(64)
    wn = V.WaveNet(2, 10, 2, 256, 256, 256, 256, 256, False, 30, -40, 192, 0)
    ce = V.ConditionEmbed(109, 128, 64)
    for i, ci in zip(range(1, 6), [64] * 5):
        getattr(ce, 'local_embed%d' % i)._initialize_params(ci)
    return V.VAE(enc, V.ExponentialMovingAverage(wn, 0.9999), ce, 64, 512, 0.25,
                 F.softmax_cross_entropy), enc, wn, ce


def test_link_tree_names_shapes_counts():
    model, enc, wn, ce = _c1_model()
    assert enc.count_params() == 82560
    assert wn.count_params() == 5198592
    assert ce.count_params() == 75712
    assert model.vq.W.shape == (512, 64)
    names = [n for n, _ in model.namedparams()]
    # generate.py:67-81 key layout: encoder / vq / decoder/{ema,target} / condition_embed
    assert '/encoder/conv1/W' in names and '/vq/W' in names
    assert '/decoder/target/resnet/0/conv/W' in names and '/decoder/ema/resnet/19/skip/b' in names
    assert '/condition_embed/global_embed/W' in names
    assert names == sorted(names, key=lambda s: [int(p) if p.isdigit() else p for p in s.split('/')])
    p = dict(model.namedparams())
    assert p['/encoder/conv1/W'].shape == (64, 1, 4, 1)
    assert p['/decoder/target/resnet/3/conv/W'].shape == (256, 256, 2, 1)
    assert p['/decoder/target/resnet/3/condition_proj/W'].shape == (256, 192, 1, 1)
    assert p['/decoder/target/resnet/3/res/W'].shape == (256, 128, 1, 1)
    assert p['/decoder/target/embed/W'].shape == (256, 256, 2, 1)
    assert [b.dilation for b in wn.resnet.children()] == [2 ** i for i in range(10)] * 2
    # EMA copy starts identical and is flagged as a never-trained shadow
    assert np.array_equal(p['/decoder/ema/proj1/W'].data, p['/decoder/target/proj1/W'].data)
    assert p['/decoder/ema/proj1/W']._shadow and not p['/decoder/target/proj1/W']._shadow


def test_oracle_and_device_model_name_maps_agree():
    import helpers as H
    cfg = dict(H.SMALL)
    P, model = H.build_model(cfg, seed=0, ema_decay=0.99)
    named = dict(model.namedparams())
    for name, arr in O.flatten_params(P):
        dn = H._dev_name(name, True)
        assert dn in named, dn
        assert named[dn].data.reshape(arr.shape).shape == arr.shape
        np.testing.assert_array_equal(named[dn].data.reshape(arr.shape), arr)


def test_no_cpu_compute_path():
    """Host arrays are storage only: any compute on them must raise, not fall back."""
    import vqvae_amd as V
    from vqvae_amd import functions as F
    from vqvae_amd.core import Variable
    from vqvae_amd.optimizers import Adam
    x = Variable(np.zeros((1, 4, 16, 1), np.float32))
    W = Variable(np.zeros((4, 4, 2, 1), np.float32))
    with pytest.raises((ValueError, TypeError)):
        F.convolution_1d(x, W)
    with pytest.raises((ValueError, TypeError)):
        F.relu(x)
    with pytest.raises(ValueError):                      # the mix guard of utils.py:183-186
        V.StraightThrough().apply((Variable(np.zeros((1, 4, 3, 1), np.float32)),
                                   Variable(np.zeros((5, 4), np.float32))))
    with pytest.raises(ValueError):
        Adam(1e-3).setup(V.Encoder(8))                   # parameters still on the host


def test_straight_through_type_check_matches_reference():
    """check_type_forward conditions of utils.py:162-174."""
    import vqvae_amd as V
    from vqvae_amd.core import InvalidType, Variable
    st = V.StraightThrough()
    f32 = np.float32
    with pytest.raises(InvalidType):
        st.check_type_forward((Variable(np.zeros((2, 4), f32)), Variable(np.zeros((3, 4), f32))))
    with pytest.raises(InvalidType):
        st.check_type_forward((Variable(np.zeros((2, 4, 3, 1, 1), f32)), Variable(np.zeros((3, 4), f32))))
    with pytest.raises(InvalidType):
        st.check_type_forward((Variable(np.zeros((2, 4, 3), f32)), Variable(np.zeros((3, 5), f32))))
    with pytest.raises(InvalidType):
        st.check_type_forward((Variable(np.zeros((2, 4, 3), np.int32)), Variable(np.zeros((3, 4), f32))))
    st.check_type_forward((Variable(np.zeros((2, 4, 3), f32)), Variable(np.zeros((3, 4), f32))))


def test_resize_tables_match_oracle_and_invert():
    from vqvae_amd.functions import resize_tables_host
    for H, outH in [(120, 7680), (8, 512), (1, 64), (5, 40)]:
        t = resize_tables_host(H, outH)
        v0, v1, w0, w1 = O.resize_tables(H, outH)
        np.testing.assert_array_equal(t['v0'], v0)
        np.testing.assert_array_equal(t['v1'], v1)
        np.testing.assert_array_equal(t['w0'], w0)
        np.testing.assert_array_equal(t['w1'], w1)
        for p in range(H):
            if H > 1:
                assert (np.flatnonzero(v0 == p) == np.arange(t['lo0'][p], t['hi0'][p])).all()
            assert (np.flatnonzero(v1 == p) == np.arange(t['lo1'][p], t['hi1'][p])).all()


def test_mulaw_product_matches_golden():
    from vqvae_amd.utils import MuLaw
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'mulaw.npz'))
    np.testing.assert_array_equal(MuLaw(256).transform(g['x']), g['q'])


def test_shard_and_alpha():
    from vqvae_amd.comm import scaled_alpha, shard
    batch = list(range(128))
    parts = [shard(batch, r, 8) for r in range(8)]
    assert parts[3] == batch[3::8] and sorted(sum(parts, [])) == batch    # updaters.py:37-38
    assert scaled_alpha(2e-4, 8) == 2e-4 / 8                               # train.py:101


def test_link_signatures_match_reference():
    import inspect
    import vqvae_amd as V
    sig = lambda f: list(inspect.signature(f).parameters)[1:]
    assert sig(V.Encoder.__init__) == ['d']
    assert sig(V.ConditionEmbed.__init__) == ['n_global_cond', 'global_embed_dim', 'local_embed_dim',
                                              'upscale_factor']
    assert sig(V.VAE.__init__) == ['encoder', 'decoder', 'condition_embed', 'd', 'k', 'beta', 'loss_func']
    assert sig(V.VAE.__call__) == ['x_enc', 'x_dec', 'global_condition', 't']
    assert sig(V.VQ.__init__) == ['k', 'd', 'initialW']
    assert sig(V.ResidualBlock.__init__) == ['filter_size', 'dilation', 'residual_channels',
                                             'dilated_channels', 'skip_channels', 'condition_dim',
                                             'dropout_zero_rate']
    assert sig(V.ResidualNet.__init__) == ['n_loop', 'n_layer', 'filter_size', 'residual_channels',
                                           'dilated_channels', 'skip_channels', 'condition_dim',
                                           'dropout_zero_rate']
    assert sig(V.WaveNet.__init__) == ['n_loop', 'n_layer', 'filter_size', 'input_dim',
                                       'residual_channels', 'dilated_channels', 'skip_channels',
                                       'quantize', 'use_logistic', 'n_mixture', 'log_scale_min',
                                       'condition_dim', 'dropout_zero_rate']
    assert sig(V.WaveNet.__call__) == ['x', 'condition', 'generating']
    assert sig(V.ExponentialMovingAverage.__init__) == ['target', 'decay']


def test_snapshot_key_layout_and_roundtrip(tmp_path):
    """the .npz key tree generate.py reads (generate.py:67-81)."""
    import vqvae_amd as V
    from vqvae_amd import serializers
    model, enc, wn, ce = _c1_model()
    path = str(tmp_path / 'snap.npz')
    state = serializers.model_state(model)
    for key in ('updater/model:main/encoder/conv1/W', 'updater/model:main/vq/W',
                'updater/model:main/decoder/target/resnet/0/conv/W',
                'updater/model:main/decoder/ema/proj2/b',
                'updater/model:main/condition_embed/global_embed/W'):
        assert key in state, key
    np.savez(path, **state)
    # generate.py-style partial loads by prefix
    enc2 = V.Encoder(64)
    serializers.load_npz(path, enc2, 'updater/model:main/encoder/')
    for (n, a), (_, b) in zip(enc.namedparams(), enc2.namedparams()):
        np.testing.assert_array_equal(a.data, b.data)
    wn2 = V.WaveNet(2, 10, 2, 256, 256, 256, 256, 256, False, 30, -40, 192, 0)
    serializers.load_npz(path, wn2, 'updater/model:main/decoder/ema/')
    np.testing.assert_array_equal(wn2.proj1.W.data, wn.proj1.W.data)
    with pytest.raises(KeyError):
        serializers.load_npz(path, V.Encoder(64), 'updater/model:main/nonexistent/')
    with pytest.raises(ValueError):
        serializers.load_npz(path, V.Encoder(32), 'updater/model:main/encoder/')


def test_mulaw_thresholds_reproduce_transform():
    """bin(x) = #{j : x >= thr[j-1]} equals utils.py:18-23 for every tested x in [-1, 1], incl.
    both neighbours of every threshold (what the device kernel relies on)."""
    from vqvae_amd.inputs import _f32_key, _key_f32, mulaw_thresholds
    from vqvae_amd.utils import MuLaw
    thr = mulaw_thresholds(256)
    assert thr.shape == (255,) and np.all(np.diff(thr) > 0)
    f = MuLaw(256).transform
    rs = np.random.RandomState(0)
    k = _f32_key(thr)
    x = np.concatenate([rs.uniform(-1, 1, 200000).astype(np.float32), thr,
                        _key_f32(k - 1), _key_f32(k + 1), np.float32([-1, 0, 1, -0.0, 1e-30, -1e-30])])
    got = np.searchsorted(thr, x, side='right').astype(np.int32)
    np.testing.assert_array_equal(got, f(x))
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'mulaw.npz'))
    np.testing.assert_array_equal(np.searchsorted(thr, g['x'], side='right'), g['q'])


def test_crop_or_pad_follows_preprocess_golden():
    """utils.py:57-81 (normalise, pad / trim) against the waveforms the reference produced."""
    import os
    from vqvae_amd.inputs import crop_or_pad
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'preprocess.npz'))
    np.testing.assert_array_equal(crop_or_pad(g['wave_short'], 255), g['mulaw_short_raw'][0, :, 0])
    np.testing.assert_array_equal(crop_or_pad(g['wave_long'], 255, start=int(g['crop_start'])),
                                  g['mulaw_long_raw'][0, :, 0])

    class _Rng(object):
        def randint(self, lo, hi):                     # utils.py:78: random.randint(0, len - length - 1)
            assert (lo, hi) == (0, 700 - 256 - 1)
            return int(g['crop_start'])
    np.testing.assert_array_equal(crop_or_pad(g['wave_long'], 255, rng=_Rng()), g['mulaw_long_raw'][0, :, 0])


def test_log_report_keys_interval_and_file(tmp_path):
    """train.py:135-140: LogReport(trigger=(100,'iteration')) averages what the model reports
    under main/ (net.py:93-95), merges validation/main/* from the Evaluator, adds epoch /
    iteration / elapsed_time and rewrites <out>/log as a JSON list; PrintReport prints the
    reference's eight columns."""
    import io
    import json
    from vqvae_amd import core
    from vqvae_amd.reporting import PRINT_KEYS, LogReport, PrintReport

    class Upd(object):
        iteration = 0
    upd, model = Upd(), object()
    log = LogReport(trigger=(100, 'iteration'), out=str(tmp_path), epoch_of=lambda it: it // 150)
    buf = io.StringIO()
    printer = PrintReport(out=buf)
    for it in range(1, 201):
        core.report({'loss1': float(it), 'loss2': 2.0, 'loss3': 0.5, 'loss': it + 2.5}, model)
        upd.iteration = it
        if it == 150:
            log.report({'validation/main/loss1': 4.0, 'validation/main/loss2': 1.0,
                        'validation/main/loss3': 0.25, 'validation/main/loss': 5.25})
        printer(log(upd))
    with open(str(tmp_path / 'log')) as f:
        entries = json.load(f)
    assert [e['iteration'] for e in entries] == [100, 200]
    assert entries[0]['main/loss1'] == pytest.approx(50.5) and entries[1]['main/loss1'] == pytest.approx(150.5)
    assert entries[0]['main/loss3'] == 0.5 and entries[1]['epoch'] == 1
    assert set(entries[0]) == {'main/loss', 'main/loss1', 'main/loss2', 'main/loss3', 'epoch', 'iteration',
                               'elapsed_time'}
    assert entries[1]['validation/main/loss1'] == 4.0 and 'validation/main/loss' in entries[1]
    assert set(PRINT_KEYS) <= set(entries[1]) | {'epoch', 'iteration'}
    lines = buf.getvalue().splitlines()
    assert len(lines) == 3 and lines[0].split() == PRINT_KEYS


def test_reporter_scope_isolates_validation():
    """Chainer's Evaluator reports inside its own scope: the training observation survives."""
    from vqvae_amd import core
    rep = core.get_current_reporter()
    core.report({'loss1': 1.0}, object())
    keep = dict(rep.observation)
    with core.report_scope({}) as obs:
        core.report({'loss1': 9.0}, object())
        assert obs['main/loss1'] == 9.0
    assert rep.observation == keep and rep.observation['main/loss1'] == 1.0


def test_rendezvous_directory_is_private_and_id_keyed(monkeypatch, tmp_path):
    import stat
    from vqvae_amd import comm
    monkeypatch.setenv('XDG_RUNTIME_DIR', str(tmp_path))
    monkeypatch.setenv('VQVAE_RDZV_ID', 'job/../x y')
    p = comm._rendezvous_path()
    d = os.path.dirname(p)
    assert os.path.dirname(d) == str(tmp_path) and stat.S_IMODE(os.lstat(d).st_mode) == 0o700
    assert os.path.basename(p) == 'uid_job____x_y'            # no path separators survive
    comm._publish(p, b'\x01' * 128)
    assert comm._read_owned(p) == b'\x01' * 128
    os.unlink(p)
    os.symlink('/etc/passwd', p)                               # a planted symlink is refused
    with pytest.raises(OSError):
        comm._read_owned(p)
    os.chmod(d, 0o755)
    with pytest.raises(RuntimeError):
        comm._rendezvous_dir()


def test_bench_traffic_is_null_when_the_profile_is_stale(monkeypatch, tmp_path):
    """bench.py reports roofline.traffic only while profiles/roofline_traffic.json carries the hash
    of the current kernel sources."""
    import json
    import bench
    prof = tmp_path / 'profiles'
    prof.mkdir()
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(bench, 'kernel_source_hash', lambda: 'aaaa')
    (prof / 'roofline_traffic.json').write_text(json.dumps(
        {'c2_B16': {'kernel_source_sha256_16': 'aaaa', 'hbm_traffic_bytes_per_launch': 123, 'source': 's'}}))
    assert bench.measured_traffic('c2_B16') == (123, 's')
    assert bench.measured_traffic('c5_B16')[0] is None
    monkeypatch.setattr(bench, 'kernel_source_hash', lambda: 'bbbb')
    v, why = bench.measured_traffic('c2_B16')
    assert v is None and 'stale' in why


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher environment starts two ranks with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* / a shared rendezvous id and propagates failure (here the
    ranks fail early: there is no GPU in the CPU test environment)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    probe = tmp_path / 'sitecustomize.py'
    probe.write_text(
        "import os\n"
        "if os.environ.get('RANK') is not None:\n"
        "    open(os.path.join(%r, 'rank%%s' %% os.environ['RANK']), 'w').write(' '.join(\n"
        "        os.environ.get(k, '?') for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'VQVAE_RDZV_ID')))\n"
        % str(tmp_path))
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + os.environ.get('PYTHONPATH', ''))
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR', 'VQVAE_RDZV_ID'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1',
                          '--warmup', '0', '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, timeout=300)
    seen = [open(str(tmp_path / ('rank%d' % r))).read().split() for r in range(2)]
    assert [s[0] for s in seen] == ['0', '1'] and [s[1] for s in seen] == ['0', '1']
    assert seen[0][2:] == seen[1][2:] and seen[0][2] == '2' and seen[0][3] == '127.0.0.1'
    assert len(seen[0][5]) == 32
    import vqvae_amd.backend as backend
    if not backend.available():
        assert out.returncode != 0                  # no GPU here: the ranks fail loudly, so does the parent


def test_plot_report_writes_the_three_loss_pngs(tmp_path):
    """train.py:141-149: loss1.png / loss2.png / loss3.png with the train and validation curves."""
    from vqvae_amd.reporting import LogReport, PlotReport, reference_plots
    if not PlotReport.available():
        pytest.skip('matplotlib is not installed (PlotReport then only warns, as Chainer does)')

    class Upd(object):
        iteration = 0
    log = LogReport(trigger=10, out=str(tmp_path))
    upd = Upd()
    for it in range(1, 31):
        upd.iteration = it
        obs = {'main/loss1': 5.5 - 0.1 * it, 'main/loss2': 1.0 / it, 'main/loss3': 0.25 / it}
        if it % 10 == 0:
            obs.update({'validation/main/loss1': 5.6 - 0.1 * it, 'validation/main/loss2': 1.1 / it,
                        'validation/main/loss3': 0.3 / it})
        log(upd, obs)
    paths = [p(log) for p in reference_plots()]
    assert [os.path.basename(p) for p in paths] == ['loss1.png', 'loss2.png', 'loss3.png']
    for p in paths:
        with open(p, 'rb') as f:
            assert f.read(8) == b'\x89PNG\r\n\x1a\n'
        assert os.path.getsize(p) > 2000


def test_hot_kernels_keep_their_staging_in_registers(tmp_path):
    """The GEMM kernels hold two K steps of operands in registers.  hipcc silently moves small private
    arrays to scratch or LDS (seen: a uint4[3] refactor cost 18 % of the step with every test green), so
    the compiled kernels' metadata is part of the contract: the launched instantiations of the conv and
    weight-gradient kernels use no scratch beyond a few spilled registers and exactly their declared LDS."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    text = ''
    procs = []
    for fam in ('conv_gemm_x3', 'conv_gemm_fp32', 'wgrad'):          # the kernel families (csrc/gemm_common.h), compiled side by side
        src = os.path.join(ROOT, 'chainer-vq-vae_amd', 'csrc', fam + '.hip')
        out = str(tmp_path / (fam + '.s'))
        procs.append((out, subprocess.Popen([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off',
                                             '-S', '--cuda-device-only', src, '-o', out], stderr=subprocess.DEVNULL)))
    for out, pr in procs:
        assert pr.wait() == 0, out
        text += open(out).read()
    meta = {}
    for m in re.finditer(r'\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+)'
                         r'.*?\.vgpr_count:\s*(\d+).*?\.vgpr_spill_count:\s*(\d+)', text, re.S):
        meta[m.group(2)] = tuple(int(m.group(i)) for i in (1, 3, 4, 5))
    expect_lds = {  # launched instantiations: (kernel substring) -> LDS bytes of the K loop's operand images
        'conv_gemm_x3_kernelILi0ELi4ELi2ELi3E': 98304, 'conv_gemm_x3_kernelILi1ELi4ELi2ELi3E': 98304,
        'conv_gemm_x3_kernelILi0ELi4ELi1ELi3E': 73728, 'conv_gemm_x3_kernelILi1ELi4ELi1ELi3E': 73728,
        'conv_gemm_x3_kernelILi0ELi2ELi1ELi3E': 49152, 'conv_gemm_x3_kernelILi2ELi2ELi1ELi3E': 49152,
        'conv_gemm_x3_kernelILi0ELi4ELi2ELi1E': 32768, 'conv_gemm_x3_kernelILi1ELi4ELi2ELi1E': 32768,
        # float32x2 (two pieces per operand)
        'conv_gemm_x3_kernelILi0ELi4ELi2ELi2E': 65536, 'conv_gemm_x3_kernelILi0ELi4ELi1ELi2E': 49152,
        'conv_gemm_x3_kernelILi1ELi4ELi1ELi2ELb0': 49152, 'conv_gemm_x3_kernelILi2ELi2ELi1ELi2E': 32768,
        'conv_gemm_x3_kernelILi1ELi4ELi1ELi2ELb1': 73728,      # + the condition step's operand image (round 5); still two workgroups per CU

        'wgrad3_kernelILi4ELi1ELi2E': 50176,
        'wgrad3_kernelILi4ELi1ELi3E': 75264, 'wgrad3_kernelILi2ELi1ELi3E': 50688, 'wgrad3_kernelILi4ELi1ELi1E': 25088,
        'conv_gemm_kernelILi1ELi4ELb0E': 49152, 'wgrad2_kernelILi4E': None,
    }
    for key, lds in expect_lds.items():
        hits = [(n, v) for n, v in meta.items() if key in n]
        assert hits, 'no kernel matching %s in the compiled module' % key
        for name, (got_lds, scratch, vgpr, spills) in hits:
            assert scratch <= 48 and spills <= 12, '%s: %d B of scratch, %d spilled VGPRs' % (name, scratch, spills)
            assert vgpr <= 256
            if lds is not None:       # (+ 64 B: the per-workgroup reduction of the epilogues that publish max |y|)
                if 'x3_kernelILi2ELi2ELi1ELi2E' in name and (name.endswith('Li2EEEvNS_8GemmArgsE') or name.endswith('Li3EEEvNS_8GemmArgsE')):
                    # the gate-derivative kernels with the fused latent pull-back (round 5): + its transposition buffers;
                    # two workgroups per CU still (241-243 VGPRs: two waves per SIMD)
                    assert got_lds == 79936 and vgpr <= 256, '%s: %d B of LDS, %d VGPRs' % (name, got_lds, vgpr)
                    continue
                assert got_lds in (lds, lds + 64), '%s: %d B of LDS (private arrays promoted?), expected %d' % (name, got_lds, lds)
    # the instantiations that run TWO 8-wave workgroups per CU must fit four waves per SIMD: 128 VGPRs
    for key in ('conv_gemm_x3_kernelILi0ELi4ELi1ELi3ELb1', 'conv_gemm_x3_kernelILi1ELi4ELi1ELi3ELb1',
                'conv_gemm_x3_kernelILi0ELi4ELi1ELi1ELb1', 'conv_gemm_x3_kernelILi1ELi4ELi1ELi1ELb1',
                'conv_gemm_x3_kernelILi0ELi4ELi1ELi2ELb1', 'conv_gemm_x3_kernelILi1ELi4ELi1ELi2ELb1',
                'wgrad3_kernelILi4ELi1ELi3E', 'wgrad3_kernelILi4ELi1ELi2E'):
        hits = [(n, v) for n, v in meta.items() if key in n]
        assert hits, 'no kernel matching %s in the compiled module' % key
        for name, (got_lds, scratch, vgpr, spills) in hits:
            assert vgpr <= 128 and 2 * got_lds <= 160 * 1024, '%s: %d VGPRs, %d B of LDS: not two workgroups per CU' % (name, vgpr, got_lds)
    # the LDS-DMA weight-gradient kernel (round 6): ONE 16-wave workgroup per CU = four waves per SIMD, two 64 KB stages, and NO
    # spill -- a scratch reload in its loop would drain the DMAs in flight with the compiler's vmcnt(0)
    hits = [(n, v) for n, v in meta.items() if 'wgrad3_dma_kernel' in n]
    assert len(hits) == 2                                   # pre-split fp16 pairs (mode 3), stored bf16 (mode 1)
    for _, (got_lds, scratch, vgpr, spills) in hits:
        assert (got_lds, scratch, spills) == (131072, 0, 0) and vgpr <= 128, hits


def test_file_rendezvous_with_eight_ranks(tmp_path):
    """The RCCL id hand-off of an 8-rank node (comm.exchange_unique_id: rank 0 publishes 128 bytes through a file in a
    private per-user directory, ranks 1..7 wait for it and read it) with real processes and no GPU: every
    rank must end up with rank 0's bytes, late starters included, and a stale file of an earlier job with the same key
    must not be taken for this job's."""
    import subprocess
    code = (
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "from vqvae_amd import comm\n"
        "rank = int(os.environ['RANK'])\n"
        "time.sleep(0.15 * (rank %% 3))\n"
        "path, raw = comm.exchange_unique_id(rank, lambda: bytes((7 * i + 3) %% 256 for i in range(128)), timeout=60)\n"
        "open(os.path.join(%r, 'got_%%d' %% rank), 'wb').write(raw)\n"
    ) % (os.path.join(ROOT, 'chainer-vq-vae_amd'), str(tmp_path))
    env = dict(os.environ, VQVAE_RDZV_ID='test_rdzv_%d' % os.getpid(), XDG_RUNTIME_DIR=str(tmp_path), WORLD_SIZE='8')
    # a stale file with this job's key, three minutes old: must be ignored until rank 0 replaces it
    from vqvae_amd import comm
    old_env = {k: os.environ.get(k) for k in ('VQVAE_RDZV_ID', 'XDG_RUNTIME_DIR')}
    os.environ.update(VQVAE_RDZV_ID=env['VQVAE_RDZV_ID'], XDG_RUNTIME_DIR=env['XDG_RUNTIME_DIR'])
    try:
        stale = comm._rendezvous_path()
        comm._publish(stale, b'\xff' * 128)
        os.utime(stale, (time.time() - 180, time.time() - 180))
    finally:
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    procs = [subprocess.Popen([sys.executable, '-c', code], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in (3, 5, 1, 7, 2, 6, 4)]          # the readers first ...
    time.sleep(0.5)
    procs.append(subprocess.Popen([sys.executable, '-c', code], env=dict(env, RANK='0', LOCAL_RANK='0')))   # ... rank 0 last
    for p in procs:
        assert p.wait(timeout=120) == 0
    want = bytes((7 * i + 3) % 256 for i in range(128))
    for r in range(8):
        assert open(os.path.join(str(tmp_path), 'got_%d' % r), 'rb').read() == want, r


def test_numa_helpers_never_raise():
    from vqvae_amd import comm
    assert comm.gpu_numa_node('ffff:ff:1f.7') is None
    out = comm.bind_to_numa_node(None)
    assert out['numa_node'] is None and out['mempolicy'] is None


def test_parallel_updater_defaults_follow_the_communicator():
    """VERDICT r4 / ADVICE r4: the overlapped exchange is the library's default whenever there is someone to exchange with
    (overlap_comm=None -> comm.size > 1), and a recorded step (graph=True) with n > 1 is only kept for communicators that
    declare themselves capture-safe -- one that works on the host per exchange would run once, at capture, and the
    replicas would diverge silently."""
    import warnings
    import vqvae_amd as V

    class Comm(object):
        def __init__(self, size, safe=False):
            self.size, self.rank = size, 0
            if safe:
                self.capture_safe = True
    assert V.VQVAE_ParallelUpdater(None, None, comm=Comm(1)).overlap_comm is False
    assert V.VQVAE_ParallelUpdater(None, None, comm=Comm(2)).overlap_comm is True
    assert V.VQVAE_ParallelUpdater(None, None, comm=Comm(8), overlap_comm=False).overlap_comm is False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        u = V.VQVAE_ParallelUpdater(None, None, comm=Comm(2), graph=True)
        assert u.graph is False and any('capture_safe' in str(x.message) for x in w)
    assert V.VQVAE_ParallelUpdater(None, None, comm=Comm(2, safe=True), graph=True).graph is True
    assert V.VQVAE_ParallelUpdater(None, None, comm=Comm(1), graph=True).graph is True
    from vqvae_amd.comm import RcclCommunicator
    assert RcclCommunicator.capture_safe is True
