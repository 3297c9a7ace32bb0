"""LogReport / PrintReport / PlotReport of the reference's trainer wiring (train.py:135-149), SURVEY.md 8f row 4.

The model reports ``loss1 / loss2 / loss3 / loss`` under the observer prefix ``main/`` on every
call (net.py:93-95); Chainer's ``LogReport(trigger=params.report_interval)`` averages every
reported value over the interval (100 iterations, params.py:8), adds ``epoch``, ``iteration``
and ``elapsed_time`` and rewrites the JSON list ``<out>/log``; ``PrintReport`` prints the chosen
columns of each new entry; ``Evaluator`` results arrive as ``validation/main/loss*``.

Device scalars are only referenced while the interval runs -- they are read back (one tiny D2H
copy each) when the interval closes, so logging adds no per-iteration synchronisation.
"""
import json
import os
import sys
import time

import numpy as np

from . import core
from .backend import DeviceArray
from .core import Variable

PRINT_KEYS = ['epoch', 'iteration', 'main/loss1', 'main/loss2', 'main/loss3',
              'validation/main/loss1', 'validation/main/loss2', 'validation/main/loss3']   # train.py:136-140


def _scalar(v):
    if isinstance(v, Variable):
        v = v.data
    if isinstance(v, DeviceArray):
        v = v.get()
    return float(np.asarray(v).reshape(-1)[0])


class LogReport(object):
    """``log = LogReport(trigger=100, out='result')``; call ``log(updater)`` after every
    ``updater.update()``.  ``log.report(dict)`` merges extra observations (an Evaluator result)
    into the running interval."""

    def __init__(self, trigger=100, out=None, log_name='log', keys=None, epoch_of=None):
        self.trigger = int(trigger[0]) if isinstance(trigger, (tuple, list)) else int(trigger)
        self.out = out
        self.log_name = log_name
        self.keys = keys
        self.epoch_of = epoch_of            # callable(iteration) -> epoch, or None (-> 0)
        self.log = []
        self._pending = {}
        self._t0 = time.time()

    def _add(self, key, value):
        if self.keys is not None and key not in self.keys:
            return
        if isinstance(value, Variable):      # keep the scalar, not the graph behind it
            value = value.data
        self._pending.setdefault(key, []).append(value)

    def report(self, values):
        for k, v in values.items():
            self._add(k, v)

    def __call__(self, updater, observation=None):
        obs = core.get_current_reporter().observation if observation is None else observation
        for k, v in obs.items():
            self._add(k, v)
        it = int(updater.iteration)
        if it % self.trigger == 0 and self._pending:
            return self._close(it)
        return None

    def _close(self, iteration):
        entry = {k: float(np.mean([_scalar(v) for v in vs])) for k, vs in sorted(self._pending.items())}
        entry['epoch'] = int(self.epoch_of(iteration)) if self.epoch_of else 0
        entry['iteration'] = iteration
        entry['elapsed_time'] = time.time() - self._t0
        self._pending = {}
        self.log.append(entry)
        if self.out is not None:
            os.makedirs(self.out, exist_ok=True)
            path = os.path.join(self.out, self.log_name)
            tmp = path + '.tmp'
            with open(tmp, 'w') as f:
                json.dump(self.log, f, indent=4)
            os.replace(tmp, path)
        return entry


class PrintReport(object):
    """Prints the columns of train.py:136-140 for each entry LogReport closes."""

    def __init__(self, entries=None, out=None):
        self.entries = list(entries or PRINT_KEYS)
        self.out = out or sys.stdout
        self._header_done = False

    def __call__(self, entry):
        if entry is None:
            return
        w = [max(10, len(e)) for e in self.entries]
        if not self._header_done:
            self.out.write('  '.join(e.ljust(n) for e, n in zip(self.entries, w)) + '\n')
            self._header_done = True
        cells = []
        for e, n in zip(self.entries, w):
            v = entry.get(e)
            cells.append(('' if v is None else ('%d' % v if isinstance(v, int) else '%.6g' % v)).ljust(n))
        self.out.write('  '.join(cells) + '\n')
        self.out.flush()


class PlotReport(object):
    """extensions.PlotReport(y_keys, x_key, file_name=...) (train.py:141-149: loss1.png, loss2.png,
    loss3.png with the train and validation curves over the iteration).  Call with the LogReport
    after each interval: ``plot(log)`` redraws ``<out>/<file_name>`` from ``log.log``.  Like
    Chainer's, it needs matplotlib (Agg backend) and only warns when that is missing."""

    _warned = False

    def __init__(self, y_keys, x_key='iteration', file_name='plot.png', out=None):
        self.y_keys = [y_keys] if isinstance(y_keys, str) else list(y_keys)
        self.x_key = x_key
        self.file_name = file_name
        self.out = out

    @staticmethod
    def available():
        try:
            import matplotlib  # noqa: F401
            return True
        except ImportError:
            return False

    def __call__(self, log):
        entries = log.log if hasattr(log, 'log') else list(log)
        out = self.out if self.out is not None else getattr(log, 'out', None)
        if out is None or not entries:
            return None
        if not self.available():
            if not PlotReport._warned:
                sys.stderr.write('PlotReport: matplotlib is not installed, no plots are written\n')
                PlotReport._warned = True
            return None
        import matplotlib
        matplotlib.use('Agg')
        import matplotlib.pyplot as plt
        fig = plt.figure()
        ax = fig.add_subplot(1, 1, 1)
        ax.set_xlabel(self.x_key)
        ax.grid(True)
        for key in self.y_keys:
            pts = [(e[self.x_key], e[key]) for e in entries if key in e and self.x_key in e]
            if pts:
                ax.plot([p[0] for p in pts], [p[1] for p in pts], marker='x', label=key)
        if ax.has_data():
            ax.legend(loc='best')
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, self.file_name)
        fig.savefig(path)
        plt.close(fig)
        return path


def reference_plots(out=None):
    """The three PlotReports of train.py:141-149."""
    return [PlotReport(['main/loss%d' % i, 'validation/main/loss%d' % i], 'iteration',
                       file_name='loss%d.png' % i, out=out) for i in (1, 2, 3)]
