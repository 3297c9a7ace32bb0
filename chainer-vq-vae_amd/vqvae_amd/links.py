"""The ``chainer.links`` subset on the hot path, Chainer constructor signatures
kept (positional order matters: net.py:12-17 passes ksize, stride, pad
positionally; modules.py:13-16 / net.py:34-43 use keywords).

Only (K, 1) kernels over a (B, C, T, 1) tensor exist in the reference, so the
H axis is the time axis and the W axis is degenerate.
"""
import numpy as np

from . import functions as F
from .core import Link, Parameter, _get_initializer, Normal, Constant


def _pair(v):
    if isinstance(v, (tuple, list)):
        if len(v) != 2:
            raise ValueError('expected an int or a pair, got %r' % (v,))
        return int(v[0]), int(v[1])
    return int(v), int(v)


class Convolution2D(Link):
    """L.Convolution2D(in_channels, out_channels, ksize, stride=1, pad=0,
    nobias=False, initialW=None, initial_bias=None).  ``in_channels=None`` defers
    the weight shape to the first call."""

    dilate = (1, 1)

    def __init__(self, in_channels, out_channels, ksize=None, stride=1, pad=0, nobias=False,
                 initialW=None, initial_bias=None):
        super(Convolution2D, self).__init__()
        if ksize is None:
            out_channels, ksize, in_channels = in_channels, out_channels, None
        kh, kw = _pair(ksize)
        if kw != 1:
            raise ValueError('only (K, 1) kernels are supported (time axis = H), got %r' % (ksize,))
        self.ksize = (kh, kw)
        self.stride = _pair(stride)
        self.pad = _pair(pad)
        self.out_channels = out_channels
        with self.init_scope():
            self.W = Parameter(_get_initializer(initialW))
            if in_channels is not None:
                self._initialize_params(in_channels)
            if nobias:
                self.b = None
            else:
                init_b = 0 if initial_bias is None else initial_bias
                self.b = Parameter(_get_initializer(init_b), (out_channels,))

    def _initialize_params(self, in_channels):
        self.W.initialize((self.out_channels, in_channels, self.ksize[0], 1))

    def _check_axis(self):
        if self.stride[1] != 1 and self.stride[1] != self.stride[0]:
            raise ValueError('W-axis stride must be 1')
        if self.pad[1] != 0:
            raise ValueError('W-axis padding must be 0 (W == 1)')

    def __call__(self, x, out_len=None, relu=False):
        if self.W.data is None:
            self._initialize_params(x.shape[1])
            if not isinstance(x.data, np.ndarray):
                self.W.to_gpu()
        self._check_axis()
        return F.convolution_1d(x, self.W, self.b, stride=self.stride[0], pad=self.pad[0],
                                dilate=self.dilate[0], out_len=out_len, relu=relu)


class DilatedConvolution2D(Convolution2D):
    """L.DilatedConvolution2D(in_channels, out_channels, ksize, stride=1, pad=0,
    dilate=1, nobias=False, initialW=None, initial_bias=None)."""

    def __init__(self, in_channels, out_channels, ksize=None, stride=1, pad=0, dilate=1,
                 nobias=False, initialW=None, initial_bias=None):
        super(DilatedConvolution2D, self).__init__(in_channels, out_channels, ksize, stride, pad,
                                                   nobias, initialW, initial_bias)
        self.dilate = _pair(dilate)


class EmbedID(Link):
    """L.EmbedID(in_size, out_size): W ~ N(0, 1) [chainer default]."""

    def __init__(self, in_size, out_size, initialW=None):
        super(EmbedID, self).__init__()
        with self.init_scope():
            init = Normal(1.0) if initialW is None else _get_initializer(initialW)
            self.W = Parameter(init, (in_size, out_size))
