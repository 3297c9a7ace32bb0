"""Data parallelism on the HIP path with real processes: two processes, each
running the product's VQVAE_ParallelUpdater on its strided shard on the GPU, exchange the flat
gradient arena (host-staged: RCCL cannot put two ranks on the one GPU of the test box) and must
(a) stay bit-identical replicas with no parameter broadcast and (b) equal one process that sums
the two shard gradients itself and steps with lr/2 (updaters.py:34-38, 71-77; train.py:101).
The RCCL communicator itself is exercised as a 1-rank communicator (real ncclCommInitRank +
ncclAllReduce + ncclCommCount) through bench.py's self-spawning launcher."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.parametrize('mode', ['rank', 'rank_overlap'])
def test_two_process_parallel_updater_on_one_gpu(gpu, tmp_path, mode):
    """mode 'rank_overlap': VQVAE_ParallelUpdater(overlap_comm=True) -- the decoder / condition-embed bucket
    of the arena is exchanged on the side stream while loss2 / loss3 back-propagate; same bits."""
    d = str(tmp_path)
    worker = os.path.join(HERE, 'dp_gpu_worker.py')
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    procs = [subprocess.Popen([sys.executable, worker, mode, str(r), '2', d], env=env) for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=900) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert subprocess.call([sys.executable, worker, 'single', '2', d], env=env, timeout=900) == 0
    a = np.load(os.path.join(d, 'params_rank0.npy'))
    b = np.load(os.path.join(d, 'params_rank1.npy'))
    s = np.load(os.path.join(d, 'params_single.npy'))
    assert a.size > 100000 and np.isfinite(a).all()
    np.testing.assert_array_equal(a, b)          # replicas identical, no broadcast
    np.testing.assert_array_equal(a, s)          # == one process summing the two shard gradients
    # the two ranks saw different shards: their local losses differ
    la = np.load(os.path.join(d, 'losses_rank0.npy'))
    lb = np.load(os.path.join(d, 'losses_rank1.npy'))
    assert np.any(la != lb)
    # all-reduce traffic really went through: every step has both ranks' files
    for step in range(3 if mode == 'rank' else 6):
        for r in range(2):
            assert os.path.exists(os.path.join(d, 'ar_%d_rank%d.npy' % (step, r)))


def test_bench_self_spawn_reaches_rccl(gpu):
    """`python bench.py --gpus N` from a bare shell spawns its own ranks and brings RCCL up: with
    one GPU on the test box that is N=1 with --force-comm (ncclCommInitRank, all-reduce of the
    gradient arena every step, ncclCommCount reported in the JSON line)."""
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--force-comm',
                          '--steps', '2', '--warmup', '1', '--batch', '2', '--no-cpu-baseline'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout.decode()[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 1 and rec['ranks_seen_by_rccl'] == [1, 1]
    assert rec['value'] > 0 and rec['roofline']['achieved'] > 0


def test_bench_two_ranks_spawned_from_bare_invocation(gpu):
    """`python bench.py --gpus 2` with no launcher: the parent spawns two ranks that rendezvous
    and call ncclCommInitRank.  On a one-GPU box rank 1 has no device of its own, so both are
    pointed at GPU 0 (VQVAE_LOCAL_DEVICE) -- RCCL then either comes up or refuses the duplicate
    device; either way both ranks got as far as RCCL's initialisation, and the parent returns a
    non-zero exit code on failure instead of hanging."""
    import ctypes as C
    from vqvae_amd import _lib
    n = C.c_int(0)
    _lib.load().vqvae_device_count(C.byref(n))
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR'):
        env.pop(k, None)
    if n.value < 2:
        env['VQVAE_LOCAL_DEVICE'] = '0'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2',
                          '--warmup', '1', '--batch', '2', '--no-cpu-baseline'],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    err = out.stderr.decode()
    if out.returncode == 0:
        rec = json.loads([l for l in out.stdout.decode().splitlines() if l.startswith('{')][0])
        assert rec['n_gpus'] == 2 and rec['ranks_seen_by_rccl'] == [2, 2]
    else:
        assert n.value < 2, err[-2000:]
        assert 'ncclCommInitRank' in err or 'rccl' in err.lower() or 'nccl' in err.lower(), err[-2000:]
