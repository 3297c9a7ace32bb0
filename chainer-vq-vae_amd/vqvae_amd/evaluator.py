"""Validation pass in the reference's sense: the Chainer ``Evaluator``
extension runs the model with ``config.train = False`` (train.py:131-132), which makes the
ExponentialMovingAverage wrapper route the decoder through its EMA copy (utils.py:156-157),
and reports ``validation/main/loss{1,2,3}`` (train.py:136-140)."""
from . import core
from .updaters import concat_examples


class Evaluator(object):
    def __init__(self, iterator, target, converter=concat_examples, device=0):
        self.iterator = iterator
        self.target = target
        self.converter = converter
        self.device = device

    def evaluate(self, max_batches=None):
        """Mean of the reported losses over the validation iterator; returns a dict with the
        reference's PrintReport keys."""
        sums = {}
        n = 0
        # the model reports 'main/loss*' on every call (net.py:93-95); Chainer's Evaluator runs it
        # inside its own reporter scope so the training observations are not overwritten
        with core.using_config('train', False), core.no_backprop_mode(), core.report_scope({}):
            while max_batches is None or n < max_batches:
                try:
                    batch = self.iterator.next()
                except StopIteration:
                    break
                losses = self.target(*self.converter(batch, self.device))
                for name, v in zip(('loss1', 'loss2', 'loss3'), losses):
                    sums[name] = sums.get(name, 0.0) + float(v.data.get())
                n += 1
        out = {'validation/main/' + k: v / max(n, 1) for k, v in sums.items()}
        if out:
            out['validation/main/loss'] = sum(out.values())
        return out
