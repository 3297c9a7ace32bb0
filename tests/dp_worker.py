"""world_size-2 worker for test_dp_gloo.py: one data-parallel training step of the
(NumPy oracle) model with the product's shard / alpha / communicator logic over
torch.distributed gloo.  CPU only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, 'chainer-vq-vae_amd'), os.path.join(ROOT, 'oracle'), HERE):
    sys.path.insert(0, p)


class GlooHostCommunicator(object):
    """The communicator interface of vqvae_amd.comm (rank, size, allreduce_grad, barrier,
    max_scalar) over torch.distributed/gloo on host NumPy buffers -- CPU tests only."""

    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()

    def allreduce_grad(self, flat):
        import torch
        t = torch.from_numpy(flat)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return flat

    def barrier(self):
        self._dist.barrier()

    def max_scalar(self, v):
        import torch
        t = torch.tensor([v], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])


def run_rank(out_path):
    import torch.distributed as dist
    import vqvae_oracle as O
    from vqvae_amd.comm import scaled_alpha, shard
    dist.init_process_group('gloo')
    comm = GlooHostCommunicator()
    cfg = dict(d=8, k=16, n_loop=1, n_layer=3, residual=16, dilated=32, skip=16, out_dim=256,
               local_dim=8, global_dim=8, n_speaker=3)
    P = O.make_params(np.random.RandomState(0), **cfg)
    full = O.synth_batch(4, length=128, n_speaker=3, seed=5)
    idx = shard(list(range(4)), comm.rank, comm.size)          # batch[rank::n]
    mine = tuple(a[idx] for a in full)

    def hook(flatG):
        names = sorted(flatG)
        flat = np.concatenate([flatG[n].reshape(-1) for n in names]).astype(np.float32)
        comm.allreduce_grad(flat)                               # SUM over ranks, in place
        out, off = {}, 0
        for n in names:
            sz = flatG[n].size
            out[n] = flat[off:off + sz].reshape(flatG[n].shape)
            off += sz
        return out
    state = {}
    for _ in range(2):
        O.train_step(P, state, mine, 1, 3, alpha=scaled_alpha(2e-4, comm.size), grad_sum_hook=hook)
    flat = np.concatenate([a.reshape(-1) for _, a in O.flatten_params(P)])
    np.save(out_path % comm.rank, flat)
    comm.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    run_rank(sys.argv[1])
