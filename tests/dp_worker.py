"""world_size-2 worker for test_dp_gloo.py.  Runs the PRODUCT's data-parallel step --
``vqvae_amd.updaters.VQVAE_ParallelUpdater.update_core`` (strided shard, three-loss backward,
adoption hook, in-place gradient SUM through the communicator, update with alpha = lr/n) -- under
torch.distributed/gloo on CPU.  The package has no CPU arithmetic, so the model and the optimizer
the updater drives are host shims over the NumPy oracle that expose exactly what update_core
touches (model(*arrays) -> three losses with .backward(), model.cleargrads(), model.vq.cleargrads(),
optimizer.target / .grads / .update() / .adopt_new_params())."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, 'chainer-vq-vae_amd'), os.path.join(ROOT, 'oracle'), HERE):
    sys.path.insert(0, p)

CFG = dict(d=8, k=16, n_loop=1, n_layer=3, residual=16, dilated=32, skip=16, out_dim=256,
           local_dim=8, global_dim=8, n_speaker=3)
N_LOOP, N_LAYER, BETA = 1, 3, 0.25


class GlooHostCommunicator(object):
    """The communicator interface of vqvae_amd.comm (rank, size, allreduce_grad, barrier,
    max_scalar) over torch.distributed/gloo on host NumPy buffers -- CPU tests only."""

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        self.calls = 0

    def allreduce_grad(self, flat):
        import torch
        t = torch.from_numpy(flat)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        self.calls += 1
        return flat

    def barrier(self):
        self._dist.barrier()

    def max_scalar(self, v):
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])


class _Loss(object):
    def __init__(self, value, fn):
        self.data = np.asarray(value)
        self._fn = fn

    def backward(self):
        self._fn()


class HostOracleModel(object):
    """VAE-shaped host shim: forward = oracle.vae_forward; each loss's backward() ADDS that loss's
    own gradient contributions (updaters.py:13-19 relies on accumulation and on vq.cleargrads())."""

    class _VQ(object):
        def __init__(self, owner):
            self.owner = owner

        def cleargrads(self):
            self.owner.view['/vq/W'][...] = 0

    def __init__(self, P):
        import vqvae_oracle as O
        self.O, self.P = O, P
        self.names = [n for n, _ in O.flatten_params(P)]
        sizes = [a.size for _, a in O.flatten_params(P)]
        self.grads = np.zeros(sum(sizes), np.float32)          # the flat gradient arena
        self.view, off = {}, 0
        for (n, a), sz in zip(O.flatten_params(P), sizes):
            self.view[n] = self.grads[off:off + sz].reshape(a.shape)
            off += sz
        self.vq = self._VQ(self)
        self.log = []

    def cleargrads(self):
        self.log.append('cleargrads')
        self.grads[...] = 0

    def _add(self, tree, prefix):
        for n, g in self.O.flatten_params(tree, prefix):
            self.view[n] += g

    def __call__(self, x_enc, x_dec, speaker, t):
        O, P = self.O, self.P
        x_enc, x_dec, t = x_enc[..., 0], x_dec[..., 0], t[..., 0]       # Preprocess's (.., 1) axis
        (l1, l2, l3), c = O.vae_forward(P, x_enc, x_dec, speaker, t, N_LOOP, N_LAYER, BETA)
        z, e, n = c['z'], c['e'], c['z'].size

        def bwd1():          # reconstruction loss: decoder, condition embed, encoder and (discarded) codebook
            self.log.append('loss1')
            gy = O.softmax_xent_bwd(c['logp'], t)
            gcond, gdec = O.wavenet_bwd(P['decoder'], c['dcache'], c['cond'], gy, N_LOOP, N_LAYER)
            gce, ge = O.cond_embed_bwd(P['condition_embed'], c['ce_hs'], speaker, gcond, need_ge=True)
            self._add(gdec, '/decoder')
            self._add(gce, '/condition_embed')
            self._add(O.encoder_bwd(P['encoder'], c['enc_hs'], ge), '/encoder')      # straight-through
            self.view['/vq/W'] += O.vq_backward(c['idx'], P['vq'], O.expand4(ge))[1]

        def bwd2():          # codebook loss: vq.W only
            self.log.append('loss2')
            ge_ = z.dtype.type(-2.0 / n) * (z - e)
            self.view['/vq/W'] += O.vq_backward(c['idx'], P['vq'], O.expand4(ge_))[1]

        def bwd3():          # commitment loss: encoder only
            self.log.append('loss3')
            gz = z.dtype.type(BETA) * z.dtype.type(2.0 / n) * (z - e)
            self._add(O.encoder_bwd(P['encoder'], c['enc_hs'], gz), '/encoder')
        return _Loss(l1, bwd1), _Loss(l2, bwd2), _Loss(l3, bwd3)


class HostAdam(object):
    def __init__(self, model, alpha):
        self.target, self.alpha, self.grads = model, alpha, model.grads
        self.state, self.t, self.adopted = {}, 0, 0

    def adopt_new_params(self):
        self.adopted += 1
        return False

    def update(self):
        O, model = model_O(self.target), self.target
        self.t += 1
        skip = '/decoder/blocks/%d/res/' % (N_LOOP * N_LAYER - 1)     # never receives a gradient
        for n, p in O.flatten_params(model.P):
            if n.startswith(skip):
                continue
            m = self.state.setdefault('m' + n, np.zeros_like(p))
            v = self.state.setdefault('v' + n, np.zeros_like(p))
            O.adam_update(p, model.view[n], m, v, self.t, self.alpha)


def model_O(model):
    return model.O


class ListIterator(object):
    def __init__(self, examples):
        self.examples = examples

    def next(self):
        return list(self.examples)


def examples_of(full):
    x_enc, x_dec, spk, t = full
    return [(x_enc[j][..., None], x_dec[j][..., None], spk[j], t[j][..., None]) for j in range(x_enc.shape[0])]


def run_rank(out_path):
    import torch.distributed as dist
    import vqvae_oracle as O
    from vqvae_amd.comm import scaled_alpha
    from vqvae_amd.updaters import VQVAE_ParallelUpdater
    dist.init_process_group('gloo')
    comm = GlooHostCommunicator()
    P = O.make_params(np.random.RandomState(0), **CFG)
    model = HostOracleModel(P)
    opt = HostAdam(model, scaled_alpha(2e-4, comm.size))            # train.py:101
    full = O.synth_batch(4, length=128, n_speaker=3, seed=5)
    upd = VQVAE_ParallelUpdater(ListIterator(examples_of(full)), opt, comm=comm, device=-1)
    for _ in range(2):
        upd.update()
    assert comm.calls == 2 and opt.adopted == 2 and upd.iteration == 2
    assert model.log == ['cleargrads', 'loss1', 'loss2', 'loss3'] * 2, model.log
    flat = np.concatenate([a.reshape(-1) for _, a in O.flatten_params(P)])
    np.save(out_path % comm.rank, flat)
    np.save((out_path % comm.rank) + '.grads.npy', model.grads)
    comm.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    run_rank(sys.argv[1])
