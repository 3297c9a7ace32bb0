"""VQVAE_StandardUpdater / VQVAE_ParallelUpdater -- mirror the reference's
updaters.py (5-19, 22-77) on minimal StandardUpdater / ParallelUpdater bases
that provide the attributes the reference's ``update_core`` bodies use
(``_iterators``, ``converter``, ``_optimizers``, ``loss_func``, ``device``,
``get_optimizer``, ``get_iterator``)."""
import numpy as np

from . import backend, core
from .comm import SingleCommunicator


def concat_examples(batch, device=None):
    """chainer.dataset.concat_examples: list of example tuples -> tuple of stacked
    arrays, moved to the device when ``device >= 0`` (updaters.py:8, 37-38)."""
    first = batch[0]
    n = len(first)
    out = []
    for i in range(n):
        arr = np.stack([np.asarray(ex[i]) for ex in batch])
        if device is not None and device >= 0:
            arr = backend.to_device(arr)
        out.append(arr)
    return tuple(out)


class StandardUpdater(object):
    def __init__(self, iterator, optimizer, converter=concat_examples, device=0, loss_func=None):
        self._iterators = iterator if isinstance(iterator, dict) else {'main': iterator}
        self._optimizers = optimizer if isinstance(optimizer, dict) else {'main': optimizer}
        self.converter = converter
        self.device = device
        self.loss_func = loss_func
        self.iteration = 0

    def get_optimizer(self, name):
        return self._optimizers[name]

    def get_iterator(self, name):
        return self._iterators[name]

    def update(self):
        self.update_core()
        self.iteration += 1


class VQVAE_StandardUpdater(StandardUpdater):
    def update_core(self):
        batch = self._iterators['main'].next()
        in_arrays = self.converter(batch, self.device)

        optimizer = self._optimizers['main']
        loss_func = self.loss_func or optimizer.target

        loss1, loss2, loss3 = loss_func(*in_arrays)
        optimizer.target.cleargrads()
        loss1.backward()
        optimizer.target.vq.cleargrads()
        loss2.backward()
        loss3.backward()
        optimizer.update()
        self.last_losses = (loss1, loss2, loss3)


class VQVAE_ParallelUpdater(StandardUpdater):
    """Data-parallel step.  The reference drives all GPUs from one process and
    reduces by device-to-device adds + a parameter broadcast (updaters.py:23-77).
    MI355X-native form: one process per GPU; this rank takes ``batch[rank::n]``
    (updaters.py:37-38), runs the same three-loss backward, then ONE RCCL
    all-reduce(sum) of the flat gradient arena replaces addgrads (sum, not mean,
    updaters.py:71-72) and -- because every rank applies the identical Adam
    update to identical parameters -- copyparams (updaters.py:76-77).  The
    optimizer's alpha must already be lr/n (train.py:101)."""

    def __init__(self, iterator, optimizer, comm=None, converter=concat_examples, device=0,
                 loss_func=None):
        super(VQVAE_ParallelUpdater, self).__init__(iterator, optimizer, converter, device,
                                                    loss_func)
        self.comm = comm or SingleCommunicator()

    def update_core(self):
        optimizer = self.get_optimizer('main')
        model = optimizer.target

        batch = self.get_iterator('main').next()
        n = self.comm.size
        in_arrays = self.converter(batch[self.comm.rank::n], self.device)

        model.cleargrads()
        loss_func = self.loss_func or model
        with core.force_backprop_mode():
            loss1, loss2, loss3 = loss_func(*in_arrays)

        model.cleargrads()
        loss1.backward()
        model.vq.cleargrads()
        loss2.backward()
        loss3.backward()

        if n > 1 or getattr(self.comm, 'always_reduce', False):
            self.comm.allreduce_grad(optimizer.grads)

        optimizer.update()
        self.last_losses = (loss1, loss2, loss3)
