"""Worker for test_gpu_dp.py: the product's data-parallel step on the HIP path with real
processes.  Two modes:

  rank <r> <n> <dir>   one of n processes sharing GPU 0, each running VQVAE_ParallelUpdater with a
                       host-staged communicator (test tooling: D2H -> files -> fixed-order sum ->
                       H2D; RCCL refuses two ranks on one device, so the exchange is staged through
                       the host while everything else is the product path);
  single <n> <dir>     ONE process that computes the n strided shard gradients itself, sums them
                       in rank order and steps with lr/n -- what the n ranks must equal bit for bit
                       (updaters.py:34-38, 71-77; train.py:101).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, 'chainer-vq-vae_amd'), os.path.join(ROOT, 'oracle'), HERE):
    sys.path.insert(0, p)

import helpers as H  # noqa: E402
import vqvae_oracle as O  # noqa: E402

CFG = dict(H.SMALL)
STEPS, GLOBAL_BATCH, LENGTH, LR, DECAY = 3, 4, 512, 2e-3, 0.99


class HostStagedCommunicator(object):
    """vqvae_amd.comm interface; allreduce_grad = D2H, publish, wait for the peers, sum in rank
    order, H2D.  Files are written atomically (rename) and keyed by call number."""

    def __init__(self, rank, size, directory, timeout=300.0):
        self.rank, self.size, self.dir, self.timeout = rank, size, directory, timeout
        self.calls = 0

    def _path(self, call, rank):
        return os.path.join(self.dir, 'ar_%d_rank%d.npy' % (call, rank))

    def allreduce_grad(self, flat, stream=None):
        if stream is not None:                 # side-stream bucket: everything enqueued so far must have run
            from vqvae_amd import backend
            backend.synchronize()
        host = flat.get()
        tmp = self._path(self.calls, self.rank) + '.tmp.npy'
        np.save(tmp, host)
        os.rename(tmp, self._path(self.calls, self.rank))
        total = None
        for r in range(self.size):
            p = self._path(self.calls, r)
            t0 = time.time()
            while not os.path.exists(p):
                if time.time() - t0 > self.timeout:
                    raise RuntimeError('rank %d: timed out waiting for %s' % (self.rank, p))
                time.sleep(0.01)
            part = host if r == self.rank else np.load(p)
            total = part.copy() if total is None else total + part
        flat.set(total)
        self.calls += 1
        return flat

    def barrier(self):
        pass

    def max_scalar(self, v):
        return v


def _examples():
    x_enc, x_dec, spk, t = O.synth_batch(GLOBAL_BATCH, length=LENGTH, n_speaker=CFG['n_speaker'], seed=77)
    return [(x_enc[j][..., None], x_dec[j][..., None], spk[j], t[j][..., None]) for j in range(GLOBAL_BATCH)]


class _Iter(object):
    def __init__(self, ex):
        self.ex = ex

    def next(self):
        return list(self.ex)


def _build(n):
    from vqvae_amd.comm import scaled_alpha
    from vqvae_amd.optimizers import Adam
    _, model = H.build_model(CFG, seed=21, ema_decay=DECAY)
    model.to_gpu(0)
    opt = Adam(scaled_alpha(LR, n))
    opt.setup(model)
    return model, opt


def run_rank(rank, n, directory, overlap=False):
    import vqvae_amd as V
    model, opt = _build(n)
    comm = HostStagedCommunicator(rank, n, directory)
    comm.always_reduce = True
    upd = V.VQVAE_ParallelUpdater(_Iter(_examples()), opt, comm=comm, device=0, overlap_comm=overlap)
    for _ in range(STEPS):
        upd.update()
    from vqvae_amd.updaters import MERGED_BACKWARD
    bk = upd._grad_buckets(opt, model, MERGED_BACKWARD)
    assert comm.calls == STEPS * (len(bk[0]) + len(bk[1]) if overlap else 1)
    upd._check_replicas(opt)        # the parameter-checksum path (runs when lazily shaped parameters are adopted under N > 1)
    np.save(os.path.join(directory, 'params_rank%d.npy' % rank), opt.params.get())
    np.save(os.path.join(directory, 'losses_rank%d.npy' % rank),
            np.array([float(l.data.get()) for l in upd.last_losses]))


def run_single(n, directory):
    """n shard gradients computed one after the other on the SAME parameters, summed in rank
    order; the EMA blend (which runs inside every training forward, utils.py:146-155) is undone
    for all but one of the forwards so that it happens once per step, as on each real rank."""
    import vqvae_amd as V
    from vqvae_amd import core
    from vqvae_amd.updaters import concat_examples, three_loss_backward
    model, opt = _build(n)
    ex = _examples()
    n_shadow = opt.params.size - opt.n_train
    for _ in range(STEPS):
        total = None
        for r in range(n):
            shadow = opt.params.flat_view(opt.n_train, n_shadow)
            keep = shadow.get() if r > 0 else None
            with core.force_backprop_mode():
                losses = model(*concat_examples(ex[r::n], 0))
            three_loss_backward(model, losses)
            g = opt.grads.get()
            total = g if total is None else total + g
            if keep is not None:
                shadow.set(keep)
        opt.grads.set(total)
        opt.update()
    np.save(os.path.join(directory, 'params_single.npy'), opt.params.get())


if __name__ == '__main__':
    if sys.argv[1] in ('rank', 'rank_overlap'):
        run_rank(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], overlap=sys.argv[1] == 'rank_overlap')
    else:
        run_single(int(sys.argv[2]), sys.argv[3])
