"""Data-parallel step on CPU (-m "not gpu"): two gloo ranks, each running the PRODUCT's
``VQVAE_ParallelUpdater.update_core`` (tests/dp_worker.py drives it over host shims of the oracle),
must (a) stay bit-identical replicas without any parameter broadcast, (b) hold the SUM of the two
shard gradients in their gradient arenas, and (c) equal a single process that sums the two strided
shards' gradients itself and steps with lr/2 (updaters.py:37-38, 71-77; train.py:101)."""
import os
import socket
import subprocess
import sys

import numpy as np

import vqvae_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_step_matches_single_process(tmp_path):
    out = str(tmp_path / 'rank%d.npy')
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='2')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dp_worker.py'), out], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    a, b = np.load(out % 0), np.load(out % 1)
    np.testing.assert_array_equal(a, b)             # replicas identical without a broadcast
    np.testing.assert_array_equal(np.load((out % 0) + '.grads.npy'), np.load((out % 1) + '.grads.npy'))

    # single-process emulation: per-shard grads summed, alpha = lr/2
    from vqvae_amd.comm import scaled_alpha, shard
    cfg = dict(d=8, k=16, n_loop=1, n_layer=3, residual=16, dilated=32, skip=16, out_dim=256,
               local_dim=8, global_dim=8, n_speaker=3)
    P = O.make_params(np.random.RandomState(0), **cfg)
    full = O.synth_batch(4, length=128, n_speaker=3, seed=5)
    shards = [tuple(x[shard(list(range(4)), r, 2)] for x in full) for r in range(2)]
    state = {}
    import copy
    for _ in range(2):
        other = {}

        def hook(flatG):
            P2 = copy.deepcopy(P)
            _, cache = O.vae_forward(P2, *shards[1], 1, 3)
            G2 = dict(O.flatten_params(O.vae_backward(P2, cache, shards[1][2], shards[1][3], 1, 3)))
            return {n: (flatG[n] + G2[n]).astype(np.float32) for n in flatG}
        O.train_step(P, state, shards[0], 1, 3, alpha=scaled_alpha(2e-4, 2), grad_sum_hook=hook)
    ref = np.concatenate([x.reshape(-1) for _, x in O.flatten_params(P)])
    np.testing.assert_allclose(a, ref, rtol=0, atol=1e-7)
