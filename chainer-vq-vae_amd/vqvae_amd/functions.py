"""FunctionNodes of the hot path (the ``chainer.functions`` subset the reference
uses), each marshalling device pointers + dims into one C-ABI call.

Reference call sites: F.relu net.py:20-24,49-53 / modules.py:155,158;
F.resize_images + EmbedID + F.concat net.py:54-63; F.mean net.py:90-91;
softmax_cross_entropy train.py:95; Variable arithmetic net.py:90-92.
"""
import ctypes as C
import os

import numpy as np

from . import _lib, backend, prepack
from .backend import DeviceArray
from .core import FunctionNode, Variable, as_variable, type_expect

_S = backend.stream


def _p(a):
    return None if a is None else a.ptr


# --------------------------------------------------------------------------- #
# raw element-wise helpers (no graph)
# --------------------------------------------------------------------------- #
def _ew(op, a, b=None, out=None, alpha=0.0, beta=0.0, like=None):
    ref = a if a is not None else like
    if out is None:
        out = DeviceArray(ref.shape, np.float32)
    _lib.call('vqvae_elementwise', op, out.size, _p(a), _p(b), out.ptr, alpha, beta, _S())
    out.amax = None            # (a maximum published for an earlier content of `out` does not describe this one)
    return out


def raw_add(a, b, out=None):
    if a.size != b.size:
        raise ValueError('shape mismatch in add: %s vs %s' % (a.shape, b.shape))
    return _ew(_lib.EW_ADD, a, b, out)


def full_like(a, value):
    return _ew(_lib.EW_FILL, None, None, None, alpha=float(value), like=a)


def raw_sum(x, scale=1.0):
    out = DeviceArray((), np.float32)
    ws = backend.workspace(4096 * 4)
    _lib.call('vqvae_sum', x.ptr, x.size, float(scale), out.ptr, ws.ptr, ws.nbytes, _S())
    return out


# --------------------------------------------------------------------------- #
# Variable arithmetic
# --------------------------------------------------------------------------- #
class Add(FunctionNode):
    def forward(self, inputs):
        a, b = inputs
        backend.require_device(a, b)
        return raw_add(a, b),

    def backward(self, indexes, gys):
        g = gys[0].data
        return g, g


class Sub(FunctionNode):
    def forward(self, inputs):
        a, b = inputs
        backend.require_device(a, b)
        if a.size != b.size:
            raise ValueError('shape mismatch in sub')
        return _ew(_lib.EW_SUB, a, b),

    def backward(self, indexes, gys):
        g = gys[0].data
        return g, _ew(_lib.EW_SCALE, g, alpha=-1.0)


class Mul(FunctionNode):
    def forward(self, inputs):
        a, b = inputs
        backend.require_device(a, b)
        self.retain_inputs((0, 1))
        if b.size == 1 and a.size != 1:
            return _ew(_lib.EW_MUL_SCALAR_DEV, a, b, alpha=1.0),
        if a.size != b.size:
            raise ValueError('shape mismatch in mul')
        return _ew(_lib.EW_MUL, a, b),

    def backward(self, indexes, gys):
        a, b = (v.data for v in self.get_retained_inputs())
        g = gys[0].data
        if b.size == 1 and a.size != 1:
            ga = _ew(_lib.EW_MUL_SCALAR_DEV, g, b, alpha=1.0)
            gb = raw_sum(_ew(_lib.EW_MUL, g, a))
            return ga, gb
        return _ew(_lib.EW_MUL, g, b), _ew(_lib.EW_MUL, g, a)


class MulScalar(FunctionNode):
    def __init__(self, c):
        self.c = float(c)

    def forward(self, inputs):
        backend.require_device(inputs[0])
        return _ew(_lib.EW_SCALE, inputs[0], alpha=self.c),

    def backward(self, indexes, gys):
        return _ew(_lib.EW_SCALE, gys[0].data, alpha=self.c),


class Square(FunctionNode):
    def forward(self, inputs):
        backend.require_device(inputs[0])
        self.retain_inputs((0,))
        return _ew(_lib.EW_SQUARE, inputs[0]),

    def backward(self, indexes, gys):
        x = self.get_retained_inputs()[0].data
        t = _ew(_lib.EW_MUL, gys[0].data, x)
        return _ew(_lib.EW_SCALE, t, alpha=2.0, out=t),


class Mean(FunctionNode):
    """F.mean over all elements (net.py:90-91)."""

    def forward(self, inputs):
        x = inputs[0]
        backend.require_device(x)
        self._shape = x.shape
        self._n = x.size
        return raw_sum(x, 1.0 / x.size),

    def backward(self, indexes, gys):
        like = DeviceArray(self._shape, np.float32)
        ones = _ew(_lib.EW_FILL, None, out=like, alpha=1.0 / self._n)
        return _ew(_lib.EW_MUL_SCALAR_DEV, ones, gys[0].data, out=ones, alpha=1.0),


class Reshape(FunctionNode):
    def __init__(self, shape):
        self.shape = tuple(shape)

    def forward(self, inputs):
        self._in_shape = inputs[0].shape
        return inputs[0].reshape(self.shape),

    def backward(self, indexes, gys):
        return gys[0].data.reshape(self._in_shape),


class ReLU(FunctionNode):
    def forward(self, inputs):
        backend.require_device(inputs[0])
        y = _ew(_lib.EW_RELU, inputs[0])
        y.amax = getattr(inputs[0], 'amax', None)   # max |relu(x)| <= max |x|: an upper bound is all a scale needs
        y.relu_out = True
        self._y = y
        return y,

    def backward(self, indexes, gys):
        if getattr(gys[0].data, 'relu_masked', False):      # the conv that read y applied the mask where it produced this gradient
            g = gys[0].data.reshape(gys[0].data.shape)
            g.relu_masked = False                           # (the mask was THIS function's: nothing upstream may claim it)
            return g,
        gx = _ew(_lib.EW_RELU_BWD, gys[0].data, self._y)
        gx.amax = getattr(gys[0].data, 'amax', None)
        return gx,


def add(a, b):
    if np.isscalar(b):
        raise NotImplementedError('Variable + scalar is not on the hot path')
    return Add().apply((a, b))[0]


def sub(a, b):
    return Sub().apply((a, b))[0]


def mul(a, b):
    if np.isscalar(b):
        return MulScalar(b).apply((a,))[0]
    if np.isscalar(a):
        return MulScalar(a).apply((b,))[0]
    return Mul().apply((a, b))[0]


class MeanSquaredDifference(FunctionNode):
    """F.mean((a - b) ** 2) (net.py:90-91) as ONE node: two launches forward, one backward, where the Variable arithmetic
    took four and five -- with the same roundings in the same order, so the losses and gradients keep their bits."""

    def forward(self, inputs):
        a, b = inputs
        backend.require_device(a, b)
        if a.size != b.size:
            raise ValueError('shape mismatch in mean_squared_difference: %s vs %s' % (a.shape, b.shape))
        self.retain_inputs((0, 1))
        out = DeviceArray((), np.float32)
        ws = backend.workspace(4096 * 4)
        _lib.call('vqvae_sqdiff_mean', a.ptr, b.ptr, a.size, out.ptr, ws.ptr, ws.nbytes, _S())
        return out,

    def backward(self, indexes, gys):
        a, b = (v.data for v in self.get_retained_inputs())
        ga = DeviceArray(a.shape, np.float32) if 0 in indexes else None
        gb = DeviceArray(b.shape, np.float32) if 1 in indexes else None
        if ga is None and gb is None:
            return None, None
        _lib.call('vqvae_sqdiff_mean_bwd', a.ptr, b.ptr, gys[0].data.ptr, a.size, _p(ga), _p(gb), _S())
        return ga, gb


def mean_squared_difference(a, b):
    return MeanSquaredDifference().apply((a, b))[0]


def square(a):
    return Square().apply((a,))[0]


def mean(a):
    return Mean().apply((a,))[0]


def reshape(a, shape):
    return Reshape(shape).apply((a,))[0]


# A ReLU's backward gx = gy * (y > 0) runs in the epilogue of the conv that reads y and produces gy (FUSE_RELU_BWD=0:
# in a pass of its own, as until round 4)
FUSE_RELU_BWD = True


def relu(a):
    return ReLU().apply((a,))[0]


# --------------------------------------------------------------------------- #
# convolution_2d / dilated_convolution_2d with ksize=(K,1)
# --------------------------------------------------------------------------- #
def _conv_desc(B, Cin, Tin, Cout, Tout, K, stride, pad, dil, relu):
    return _lib.Conv1dDesc(B, Cin, Tin, Cout, Tout, K, stride, pad, dil, 1 if relu else 0)


def _mask_has_a_taker(v):
    """True when the Variable ``v`` (a conv's input that carries ``relu_out``) was produced by a node whose backward CONSUMES
    a gradient flagged ``relu_masked`` -- F.relu, a conv with a fused ReLU, ResidualNet's fused ReLU on the skip sum, through
    any number of reshapes.  Only then may the conv that reads ``v`` apply that ReLU's mask to the gradient it produces: a
    ReLU output re-wrapped as a LEAF (``Variable(h.data)``, gradient-check inputs) must receive the conv's plain dL/dx."""
    c = getattr(v, 'creator', None)
    while isinstance(c, Reshape):
        c = getattr(c.inputs[0], 'creator', None)
    if c is None:
        return False
    if isinstance(c, ReLU):
        return True
    if isinstance(c, Conv1dFunction):
        return bool(c.relu)
    return bool(getattr(c, 'relu_out', False))          # wavenet.ResidualStackFunction(relu_out=True)


class Conv1dFunction(FunctionNode):
    """y = conv(x, W, b)[..., :out_len] (+ fused ReLU).  x:(B,Cin,T,1), W:(Cout,Cin,K,1).
    Replaces chainer.functions.convolution_2d / dilated_convolution_2d behind
    L.Convolution2D / L.DilatedConvolution2D (net.py:12-17,34-43; modules.py:13-22,
    127-141)."""

    def __init__(self, stride=1, pad=0, dilate=1, out_len=None, relu=False):
        self.stride, self.pad, self.dil = int(stride), int(pad), int(dilate)
        self.out_len = out_len
        self.relu = relu

    def check_type_forward(self, in_vars):
        x, W = in_vars[0], in_vars[1]
        type_expect((x.ndim in (3, 4), 'conv: x must be (B,C,T[,1]), got %s' % (x.shape,)),
                    (W.ndim in (3, 4), 'conv: W must be (Cout,Cin,K[,1])'),
                    (x.shape[1] == W.shape[1],
                     'conv: channel mismatch x %s vs W %s' % (x.shape, W.shape)))

    def forward(self, inputs):
        x, W = inputs[0], inputs[1]
        b = inputs[2] if len(inputs) > 2 else None
        backend.require_device(x, W)
        B, Cin, Tin = x.shape[:3]
        Cout, _, K = W.shape[:3]
        nat = (Tin + 2 * self.pad - self.dil * (K - 1) - 1) // self.stride + 1
        Tout = nat if self.out_len is None else min(self.out_len, nat)
        self.desc = _conv_desc(B, Cin, Tin, Cout, Tout, K, self.stride, self.pad, self.dil, self.relu)
        y = DeviceArray((B, Cout, Tout, 1), np.float32)
        wsb = _lib.load().vqvae_conv1d_workspace_bytes(C.byref(self.desc))
        ws = backend.workspace(wsb)
        # 'float32x2', a launch large enough for the three-product kernels: the operand's maximum travels with it (or
        # is found once and remembered on it for the backward), the result's is published by the epilogue
        self._f32x2 = bool(_lib.load().vqvae_conv1d_uses_f32x2(C.byref(self.desc)))
        pre = prepack.lookup(self.inputs[1], W, self.desc, 0)      # the weight slab, if this step packed it ahead
        if self._f32x2:
            y.amax = backend.new_amax()
            am = _lib.Conv1dAmax(backend.absmax(x).ptr, None, y.amax.ptr, pre)
            _lib.call('vqvae_conv1d_fwd_amax', C.byref(self.desc), x.ptr, W.ptr, _p(b), y.ptr, ws.ptr,
                      ws.nbytes, C.byref(am), _S())
        elif pre is not None:
            _lib.call('vqvae_conv1d_fwd_amax', C.byref(self.desc), x.ptr, W.ptr, _p(b), y.ptr, ws.ptr,
                      ws.nbytes, C.byref(_lib.Conv1dAmax(None, None, None, pre)), _S())
        else:
            _lib.call('vqvae_conv1d_fwd', C.byref(self.desc), x.ptr, W.ptr, _p(b), y.ptr, ws.ptr,
                      ws.nbytes, _S())
        self.retain_inputs((0, 1))
        self._y = y if self.relu else None
        y.relu_out = bool(self.relu)
        self._has_b = b is not None
        self._x_shape = x.shape
        return y,

    def backward(self, indexes, gys):
        x, W = (v.data for v in self.get_retained_inputs())
        gy = gys[0].data
        if self.relu and not getattr(gy, 'relu_masked', False):     # (masked: the conv that read y did it in its epilogue)
            gy0 = gy
            gy = _ew(_lib.EW_RELU_BWD, gy, self._y)
            gy.amax = getattr(gy0, 'amax', None)
        wsb = _lib.load().vqvae_conv1d_workspace_bytes(C.byref(self.desc))
        ws = backend.workspace(wsb)
        f32x2 = self._f32x2 and bool(_lib.load().vqvae_conv1d_uses_f32x2(C.byref(self.desc)))
        d = self.desc
        if (FUSE_S2_BWD and 1 in indexes and not f32x2 and _lib.load().vqvae_get_matmul_dtype() != 1
                and _lib.load().vqvae_conv_s2_bwd_supported(d.Cin, d.Cout, d.K, d.stride, d.pad, d.dil, d.Tin, d.Tout)):
            # an encoder stage (net.py:12-17): backward-data, weight and bias gradient in ONE launch + a reduce (csrc/latent.hip)
            gx = DeviceArray(self._x_shape, np.float32) if 0 in indexes else None
            wv = self.inputs[1]
            buf = wv.grad_buffer() if hasattr(wv, 'grad_buffer') else None
            gW = buf.reshape(W.shape) if buf is not None else DeviceArray(W.shape, np.float32)
            gb = None
            if self._has_b:
                bv = self.inputs[2]
                buf = bv.grad_buffer() if hasattr(bv, 'grad_buffer') else None
                gb = buf if buf is not None else DeviceArray((d.Cout,), np.float32)
            mask = bool(gx is not None and getattr(x, 'relu_out', False) and FUSE_RELU_BWD and _mask_has_a_taker(self.inputs[0]))
            w2 = backend.workspace(_lib.load().vqvae_conv_s2_bwd_workspace_bytes(d.B, d.Cin, d.Tin))
            _lib.call('vqvae_conv_s2_bwd', d.B, d.Cin, d.Tin, d.Tout, x.ptr, W.ptr, gy.ptr, 1 if mask else 0,
                      _p(gx), gW.ptr, _p(gb), 0, w2.ptr, w2.nbytes, _S())
            if gx is not None:
                gx.relu_masked = mask
            return (gx, gW, gb) if self._has_b else (gx, gW)
        gx = None
        if 0 in indexes:
            gx = DeviceArray(self._x_shape, np.float32)
            am = None
            pre = prepack.lookup(self.inputs[1], W, self.desc, 1)
            if f32x2:
                gx.amax = backend.new_amax()
                am = _lib.Conv1dAmax(None, backend.absmax(gy).ptr, gx.amax.ptr, pre)
            elif pre is not None:
                am = _lib.Conv1dAmax(None, None, None, pre)
            if getattr(x, 'relu_out', False) and FUSE_RELU_BWD and _mask_has_a_taker(self.inputs[0]):
                # x is the output of a ReLU: that ReLU's backward, gx * (x > 0), in this launch's epilogue
                _lib.call('vqvae_conv1d_bwd_data_relu', C.byref(self.desc), W.ptr, gy.ptr, x.ptr, gx.ptr,
                          ws.ptr, ws.nbytes, C.byref(am) if am is not None else None, _S())
                gx.relu_masked = True
            elif am is not None:
                _lib.call('vqvae_conv1d_bwd_data_amax', C.byref(self.desc), W.ptr, gy.ptr, gx.ptr, 0,
                          ws.ptr, ws.nbytes, C.byref(am), _S())
            else:
                _lib.call('vqvae_conv1d_bwd_data', C.byref(self.desc), W.ptr, gy.ptr, gx.ptr, 0,
                          ws.ptr, ws.nbytes, _S())
        gW = gb = None
        if 1 in indexes:
            wv = self.inputs[1]
            buf = wv.grad_buffer() if hasattr(wv, 'grad_buffer') else None
            gW = buf.reshape(W.shape) if buf is not None else DeviceArray(W.shape, np.float32)
            if self._has_b:
                bv = self.inputs[2]
                buf = bv.grad_buffer() if hasattr(bv, 'grad_buffer') else None
                gb = buf if buf is not None else DeviceArray((self.desc.Cout,), np.float32)
            if f32x2:
                am = _lib.Conv1dAmax(backend.absmax(x).ptr, backend.absmax(gy).ptr, None)
                _lib.call('vqvae_conv1d_bwd_weight_amax', C.byref(self.desc), x.ptr, gy.ptr, gW.ptr,
                          _p(gb), 0, ws.ptr, ws.nbytes, C.byref(am), _S())
            else:
                _lib.call('vqvae_conv1d_bwd_weight', C.byref(self.desc), x.ptr, gy.ptr, gW.ptr,
                          _p(gb), 0, ws.ptr, ws.nbytes, _S())
        return (gx, gW, gb) if self._has_b else (gx, gW)


# --------------------------------------------------------------------------- #
# a stack of "same"-padded dilated 3-tap convs + ReLU at the latent rate in ONE launch per direction
# (ConditionEmbed's five local convs, net.py:34-53; csrc/latent.hip)
# --------------------------------------------------------------------------- #
FUSE_CONV_STACK = True
FUSE_S2_BWD = True        # an encoder stage's whole backward in one launch (Conv1dFunction.backward)


class ConvStackFunction(FunctionNode):
    """inputs (x, W_1, b_1, ..., W_L, b_L) -> h_L, h_l = relu(conv(h_{l-1}; W_l, pad = dilate = dil_l) + b_l).  One workgroup per
    sample keeps the sample's (C, T') state in LDS through all layers; the backward launch walks them back (ReLU masks,
    bias / weight gradients as per-sample shares summed in a fixed order, backward-data) and a second tiny launch reduces the
    shares.  Replaces L conv launches forward and 3 L backward, all of them latency (25-60 us each for ~2 us of arithmetic)."""

    def __init__(self, dilations):
        self.dil = [int(d) for d in dilations]

    @staticmethod
    def supported(x, Ws, dilations):
        if not FUSE_CONV_STACK or not isinstance(x, DeviceArray):
            return False
        if _lib.load().vqvae_get_matmul_dtype() == 1:      # 'bfloat16': every conv rounds its operands (the mode's contract); the stack computes in fp32
            return False
        if len(x.shape) < 3:
            return False
        B, Cc, T = x.shape[:3]
        if any(W is None or not isinstance(W, DeviceArray) or tuple(W.shape[:3]) != (Cc, Cc, 3) for W in Ws):
            return False
        arr = (C.c_int * len(dilations))(*[int(d) for d in dilations])
        return bool(_lib.load().vqvae_convstack_supported(len(dilations), Cc, T, arr))

    def forward(self, inputs):
        backend.require_device(*inputs)
        x = inputs[0]
        L = len(self.dil)
        assert len(inputs) == 1 + 2 * L
        B, Cc, T = x.shape[:3]
        Ws, bs = inputs[1::2], inputs[2::2]
        hs = [DeviceArray(x.shape, np.float32) for _ in range(L)]
        dil = (C.c_int * L)(*self.dil)
        _lib.call('vqvae_convstack_fwd', L, B, Cc, T, dil, x.ptr, _lib.ptr_array(Ws), _lib.ptr_array(bs),
                  _lib.ptr_array(hs), _S())
        self._hs = hs
        self.retain_inputs(tuple(range(len(inputs))))
        hs[-1].relu_out = False           # (this node applies its own ReLU masks: a reader must not fuse one into its backward)
        return hs[-1],

    def backward(self, indexes, gys):
        ins = [v.data for v in self.get_retained_inputs()]
        x = ins[0]
        L = len(self.dil)
        B, Cc, T = x.shape[:3]
        Ws = ins[1::2]
        gy = gys[0].data
        gx = DeviceArray(x.shape, np.float32) if 0 in indexes else None
        gWs, gbs = [], []
        for l in range(L):
            wv, bv = self.inputs[1 + 2 * l], self.inputs[2 + 2 * l]
            gW = gb = None
            if (1 + 2 * l) in indexes:
                buf = wv.grad_buffer() if hasattr(wv, 'grad_buffer') else None
                gW = buf.reshape(Ws[l].shape) if buf is not None else DeviceArray(Ws[l].shape, np.float32)
            if (2 + 2 * l) in indexes:
                buf = bv.grad_buffer() if hasattr(bv, 'grad_buffer') else None
                gb = buf if buf is not None else DeviceArray((Cc,), np.float32)
            gWs.append(gW)
            gbs.append(gb)
        dil = (C.c_int * L)(*self.dil)
        nbytes = _lib.load().vqvae_convstack_workspace_bytes(L, B, Cc)
        ws = DeviceArray((int(nbytes) // 4 + 1,), np.float32)
        _lib.call('vqvae_convstack_bwd', L, B, Cc, T, dil, x.ptr, _lib.ptr_array(Ws), _lib.ptr_array(self._hs), gy.ptr,
                  _p(gx), _lib.ptr_array(gWs), _lib.ptr_array(gbs), 0, ws.ptr, ws.nbytes, _S())
        self._hs = None
        out = [gx]
        for l in range(L):
            out += [gWs[l], gbs[l]]
        return tuple(out)


def conv_stack(x, convs):
    """relu(conv_L(... relu(conv_1(x)))) for a list of DilatedConvolution2D links (3 taps, pad == dilate, stride 1, C -> C): one
    fused node where the library serves the shape, else the links one by one (relu fused in each conv's epilogue)."""
    x = as_variable(x)
    ok = all(getattr(c, 'W', None) is not None and c.W.data is not None and c.b is not None and c.b.data is not None
             and c.stride[0] == 1 and c.pad[0] == c.dilate[0] for c in convs)
    if ok and ConvStackFunction.supported(x.data, [c.W.data for c in convs], [c.dilate[0] for c in convs]):
        args = [x]
        for c in convs:
            args += [c.W, c.b]
        return ConvStackFunction([c.dilate[0] for c in convs]).apply(args)[0]
    h = x
    for c in convs:
        h = c(h, relu=True)
    return h


def convolution_1d(x, W, b=None, stride=1, pad=0, dilate=1, out_len=None, relu=False):
    args = (x, W) if b is None else (x, W, b)
    return Conv1dFunction(stride, pad, dilate, out_len, relu).apply(args)[0]


# --------------------------------------------------------------------------- #
# condition assembly = F.resize_images(local) ++ resize_images(EmbedID(ids)) ++ F.concat
# (net.py:54-63) written straight into the (B, Cl+G, T, 1) tensor
# --------------------------------------------------------------------------- #
_resize_cache = {}


def resize_tables_host(H, outH):
    """Chainer's resize_images sampling along one axis, computed on the host in
    float64 and cast like Chainer does: v = linspace(0, H-1, outH); v0 =
    floor(v).clip(0, H-2); v1 = v0+1; weights (v1-v), (v-v0) as float32.  Also
    the inverse ranges the backward kernel needs: outputs i with v0[i]==p are
    [lo0[p], hi0[p]); with v1[i]==p are [lo1[p], hi1[p])."""
    if H == 1:
        v0 = np.zeros(outH, np.int32)
        v1 = np.zeros(outH, np.int32)
        w0 = np.zeros(outH, np.float32)
        w1 = np.ones(outH, np.float32)
    else:
        v = np.linspace(0, H - 1, num=outH)
        v0 = np.floor(v).astype(np.int32).clip(0, H - 2)
        v1 = v0 + 1
        w0 = (v1 - v).astype(np.float32)
        w1 = (v - v0).astype(np.float32)
    pos = np.arange(H)
    lo0 = np.searchsorted(v0, pos, 'left').astype(np.int32)
    hi0 = np.searchsorted(v0, pos, 'right').astype(np.int32)
    lo1 = np.searchsorted(v1, pos, 'left').astype(np.int32)
    hi1 = np.searchsorted(v1, pos, 'right').astype(np.int32)
    if H == 1:           # weight-0 taps contribute nothing
        lo0[:] = 0
        hi0[:] = 0
    return dict(v0=v0, v1=v1, w0=w0, w1=w1, lo0=lo0, hi0=hi0, lo1=lo1, hi1=hi1)


def resize_tables(H, outH):
    key = (H, outH)
    if key not in _resize_cache:
        _resize_cache[key] = {k: backend.to_device(a) for k, a in resize_tables_host(H, outH).items()}
    return _resize_cache[key]


class LazyUpsampled(DeviceArray):
    """The (B, Cl+G, T, 1) condition tensor of net.py:54-63 in unmaterialised form: it
    carries the latent-rate tensor ``latent`` (B, Cl+G, Tl) = [local | speaker broadcast]
    and is only expanded to full rate when something reads ``.ptr``.  ResidualNet
    consumes ``latent`` directly (condition projection at the latent rate)."""
    __slots__ = ('latent', 'upscale', '_full', '_make')

    def __init__(self, shape, latent, upscale, make):
        self.shape = tuple(shape)
        self.dtype = np.dtype(np.float32)
        self._block = None
        self.latent = latent
        self.upscale = upscale
        self._full = None
        self._make = make

    def materialize(self):
        if self._full is None:
            self._full = self._make()
        return self._full

    @property
    def ptr(self):
        return self.materialize().ptr

    def reshape(self, *shape):
        return self.materialize().reshape(*shape)

    def get(self):
        return self.materialize().get()


class LatentGrad(DeviceArray):
    """Gradient w.r.t. a LazyUpsampled condition, already pulled back to the latent
    rate (B, Cl+G, Tl) by the consumer.  Only ConditionAssemble.backward may take it."""
    __slots__ = ('latent',)

    def __init__(self, shape, latent):
        self.shape = tuple(shape)
        self.dtype = np.dtype(np.float32)
        self._block = None
        self.latent = latent

    @property
    def ptr(self):
        raise RuntimeError('a latent-rate condition gradient cannot be read at full rate')


class ConditionAssemble(FunctionNode):
    def __init__(self, upscale, lazy=True):
        self.upscale = int(upscale)
        self.lazy = lazy

    def _full(self, local, E, ids):
        B, Cl, Tl = local.shape[:3]
        G = E.shape[1]
        T = self.upscale * Tl
        out = DeviceArray((B, Cl + G, T, 1), np.float32)
        bs = (Cl + G) * T
        tb = resize_tables(Tl, T)
        _lib.call('vqvae_upsample_linear_fwd', local.ptr, B, Cl, Tl, T, tb['v0'].ptr, tb['v1'].ptr,
                  tb['w0'].ptr, tb['w1'].ptr, out.ptr, bs, _S())
        _lib.call('vqvae_embed_broadcast_fwd', E.ptr, ids.ptr, B, G, T, out.ptr + Cl * T * 4, bs, _S())
        return out

    def forward(self, inputs):
        local, E, ids = inputs
        backend.require_device(local, E, ids)
        B, Cl, Tl = local.shape[:3]
        n_id, G = E.shape
        T = self.upscale * Tl
        self._dims = (B, Cl, Tl, G, T, n_id)
        self._ids = ids
        self._local_shape = local.shape
        if not self.lazy or Tl < 2:
            return self._full(local, E, ids),
        # latent-rate condition [local | speaker] (B, Cl+G, Tl)
        lat = DeviceArray((B, Cl + G, Tl), np.float32)
        bs = (Cl + G) * Tl
        tb = resize_tables(Tl, Tl)                # identity resize == strided copy into the slice
        _lib.call('vqvae_upsample_linear_fwd', local.ptr, B, Cl, Tl, Tl, tb['v0'].ptr, tb['v1'].ptr,
                  tb['w0'].ptr, tb['w1'].ptr, lat.ptr, bs, _S())
        _lib.call('vqvae_embed_broadcast_fwd', E.ptr, ids.ptr, B, G, Tl, lat.ptr + Cl * Tl * 4, bs, _S())
        return LazyUpsampled((B, Cl + G, T, 1), lat, self.upscale,
                             lambda: self._full(local, E, ids)),

    def backward(self, indexes, gys):
        B, Cl, Tl, G, T, n_id = self._dims
        g = gys[0].data
        gl = DeviceArray(self._local_shape, np.float32)
        gE = DeviceArray((n_id, G), np.float32)
        ws = backend.workspace(B * G * 4)
        if isinstance(g, LatentGrad):
            g = g.latent
            Tg = Tl
        else:
            Tg = T
        bs = (Cl + G) * Tg
        tb = resize_tables(Tl, Tg)
        _lib.call('vqvae_upsample_linear_bwd', g.ptr, bs, B, Cl, Tl, Tg, tb['w0'].ptr, tb['w1'].ptr,
                  tb['lo0'].ptr, tb['hi0'].ptr, tb['lo1'].ptr, tb['hi1'].ptr, gl.ptr, Cl * Tl, _S())
        _lib.call('vqvae_embed_broadcast_bwd', g.ptr + Cl * Tg * 4, bs, self._ids.ptr, B, G, Tg, n_id,
                  gE.ptr, 0, ws.ptr, ws.nbytes, _S())
        return gl, gE, None


LAZY_CONDITION = True     # keep the condition at the latent rate until a consumer needs it


def condition_assemble(local, E, ids, upscale):
    ids = as_variable(ids)
    ids.requires_grad = False
    return ConditionAssemble(upscale, lazy=LAZY_CONDITION).apply((local, E, ids))[0]


# --------------------------------------------------------------------------- #
# softmax cross entropy (train.py:95)
# --------------------------------------------------------------------------- #
class SoftmaxCrossEntropy(FunctionNode):
    def check_type_forward(self, in_vars):
        y, t = in_vars
        type_expect((y.ndim in (3, 4), 'softmax_cross_entropy: y must be (B,q,T[,1])'),
                    (t.dtype == np.int32, 'softmax_cross_entropy: t must be int32'),
                    (y.shape[0] == t.shape[0] and y.shape[2] == t.shape[1],
                     'softmax_cross_entropy: shape mismatch %s vs %s' % (y.shape, t.shape)))

    def forward(self, inputs):
        y, t = inputs
        backend.require_device(y, t)
        B, q, T = y.shape[:3]
        lse = DeviceArray((B, T), np.float32)
        loss = DeviceArray((), np.float32)
        ws = backend.workspace(4096 * 4)
        _lib.call('vqvae_softmax_xent_fwd', y.ptr, t.ptr, B, q, T, lse.ptr, loss.ptr, ws.ptr,
                  ws.nbytes, _S())
        self._saved = (y, t, lse)
        return loss,

    def backward(self, indexes, gys):
        y, t, lse = self._saved
        B, q, T = y.shape[:3]
        gy = DeviceArray(y.shape, np.float32)
        if _lib.load().vqvae_get_matmul_dtype() == 3:
            # 'float32x2': the kernel also leaves an upper bound of max |gy| (|g| / (B T)) where the conv that reads gy
            # looks for its operand's maximum -- no scan of the 126 MB gradient
            am = DeviceArray((_lib.AMAX_SLOTS,), np.uint32)      # (every word is written by the kernel)
            _lib.call('vqvae_softmax_xent_bwd_amax', y.ptr, t.ptr, lse.ptr, gys[0].data.ptr, B, q, T, gy.ptr,
                      am.ptr, _S())
            gy.amax = am
        else:
            _lib.call('vqvae_softmax_xent_bwd', y.ptr, t.ptr, lse.ptr, gys[0].data.ptr, B, q, T, gy.ptr,
                      _S())
        return gy, None


def softmax_cross_entropy(y, t):
    t = as_variable(t)
    t.requires_grad = False
    return SoftmaxCrossEntropy().apply((y, t))[0]


# --------------------------------------------------------------------------- #
# discretised mixture-of-logistics loss (WaveNet/modules.py:169-230)
# --------------------------------------------------------------------------- #
class MixtureOfLogisticsNLL(FunctionNode):
    def __init__(self, quantize, log_scale_min):
        self.quantize = int(quantize)
        self.log_scale_min = float(log_scale_min)

    def check_type_forward(self, in_vars):
        y, t = in_vars
        type_expect((y.ndim in (3, 4) and y.shape[1] % 3 == 0,
                     'logistic loss: y must be (B, 3*nr_mix, T[,1])'),
                    (np.dtype(t.dtype).kind == 'f', 'logistic loss: t must be float'),
                    (t.shape[0] == y.shape[0] and t.shape[1] == 1 and t.shape[2] == y.shape[2],
                     'logistic loss: t must be (B,1,T[,1]), got %s vs y %s' % (t.shape, y.shape)))

    def forward(self, inputs):
        y, t = inputs
        backend.require_device(y, t)
        B, C3, T = y.shape[:3]
        loss = DeviceArray((), np.float32)
        ws = backend.workspace(4096 * 4)
        _lib.call('vqvae_mol_nll_fwd', y.ptr, t.ptr, B, C3 // 3, T, self.quantize,
                  self.log_scale_min, loss.ptr, ws.ptr, ws.nbytes, _S())
        self._saved = (y, t)
        return loss,

    def backward(self, indexes, gys):
        y, t = self._saved
        B, C3, T = y.shape[:3]
        gy = DeviceArray(y.shape, np.float32)
        _lib.call('vqvae_mol_nll_bwd', y.ptr, t.ptr, gys[0].data.ptr, B, C3 // 3, T, self.quantize,
                  self.log_scale_min, gy.ptr, _S())
        return gy, None


def mixture_of_logistics_nll(y, t, quantize=256, log_scale_min=-40.0):
    t = as_variable(t)
    t.requires_grad = False
    return MixtureOfLogisticsNLL(quantize, log_scale_min).apply((y, t))[0]


# --------------------------------------------------------------------------- #
# decoder embed conv on bin indices (device-side input pipeline)
# --------------------------------------------------------------------------- #
class EmbedConvFromIndices(FunctionNode):
    """Causal (K,1) conv of the one-hot of ``idx`` (modules.py:127-128, 151-152) without the
    one-hot: forward is a K-column gather of W (bit-identical to the dense conv); the weight
    gradient is a weighted bincount of the output-gradient rows by class (the same kernel the
    one-hot float input takes once the device has recognised it, so both inputs train alike)."""

    def check_type_forward(self, in_vars):
        idx, W = in_vars[0], in_vars[1]
        type_expect((idx.dtype == np.int32, 'embed_conv_indices: indices must be int32'),
                    (W.ndim in (3, 4), 'embed_conv_indices: W must be (Cout, q, K[,1])'))

    def forward(self, inputs):
        idx, W = inputs[0], inputs[1]
        b = inputs[2] if len(inputs) > 2 else None
        backend.require_device(idx, W)
        B = idx.shape[0]
        T = idx.size // B
        Cout, q, K = W.shape[:3]
        y = DeviceArray((B, Cout, T, 1), np.float32)
        _lib.call('vqvae_embed_gather_fwd', idx.ptr, T, B, T, W.ptr, _p(b), Cout, q, K, y.ptr, _S())
        if _lib.load().vqvae_get_matmul_dtype() == 3:
            # 'float32x2': the first gate GEMM wants max |y|; a bound from the weights (one small launch) instead of a
            # scan of the (B, Cout, T) tensor
            y.amax = DeviceArray((_lib.AMAX_SLOTS,), np.uint32)
            _lib.call('vqvae_embed_gather_bound', W.ptr, _p(b), Cout, q, K, y.amax.ptr, _S())
        self._saved = (idx, B, T, Cout, q, K, b is not None)
        return y,

    def backward(self, indexes, gys):
        idx, B, T, Cout, q, K, has_b = self._saved
        gy = gys[0].data
        gW, gb = _embed_wgrad(self, None, idx, None, gy, B, Cout, q, K, T, has_b)
        return (None, gW, gb) if has_b else (None, gW)


def _embed_wgrad(node, x, idx, flag, gy, B, Cout, q, K, T, has_b):
    """Weight (and bias) gradient of the embed conv as a weighted bincount of the rows of ``gy``
    by input class.  The kernel is bound by the LDS atomic unit (about 3 clocks per lane-add:
    0.31 ms at configs[1], level with the dense MFMA wgrad it replaces, but without reading the
    one-hot tensor).  Running it on the side stream under the rest of the backward sweep was
    measured and costs 0.13 ms/step more than it hides (A/B, 3x interleaved): main stream."""
    wv = node.inputs[1]
    buf = wv.grad_buffer() if hasattr(wv, 'grad_buffer') else None
    gW = buf.reshape(wv.shape) if buf is not None else DeviceArray(wv.shape, np.float32)
    gb = None
    if has_b:
        bv = node.inputs[2]
        buf = bv.grad_buffer() if hasattr(bv, 'grad_buffer') else None
        gb = buf if buf is not None else DeviceArray((Cout,), np.float32)
    ws = backend.workspace(_lib.load().vqvae_embed_onehot_workspace_bytes(B, Cout, q, K, T))
    _lib.call('vqvae_embed_onehot_wgrad', _p(x), idx.ptr, _p(flag), gy.ptr, B, Cout, q, K, T, gW.ptr,
              _p(gb), 0, ws.ptr, ws.nbytes, _S())
    return gW, gb


class EmbedConvOneHot(FunctionNode):
    """The decoder's causal embed conv (modules.py:127-128, 151-152) on the reference's input
    contract -- the one-hot FLOAT tensor (B, q, T, 1) of utils.py:85-87.  A dense conv over it is
    2*B*T*Cout*q*K FLOP of multiplications by zero, forward and again in the weight gradient.
    The device scans the tensor once (class index per column + a flag "exactly one-hot") and then
    runs EITHER the gather / bincount forms OR the dense kernels, selected by that flag on the
    device: no host round trip, and an input that is not one-hot silently takes the dense path.
    Forward is bit-identical to the dense conv on a one-hot input."""

    def check_type_forward(self, in_vars):
        x, W = in_vars[0], in_vars[1]
        type_expect((x.ndim in (3, 4) and W.ndim in (3, 4), 'embed conv: x (B,q,T[,1]), W (Cout,q,K[,1])'),
                    (x.shape[1] == W.shape[1], 'embed conv: in-channels mismatch'))

    def forward(self, inputs):
        x, W = inputs[0], inputs[1]
        b = inputs[2] if len(inputs) > 2 else None
        backend.require_device(x, W)
        B, q, T = x.shape[:3]
        Cout, _, K = W.shape[:3]
        y = DeviceArray((B, Cout, T, 1), np.float32)
        idx = DeviceArray((B, T), np.int32)
        flag = DeviceArray((1,), np.int32)
        ws = backend.workspace(_lib.load().vqvae_embed_onehot_workspace_bytes(B, Cout, q, K, T))
        _lib.call('vqvae_embed_onehot_fwd', x.ptr, W.ptr, _p(b), B, Cout, q, K, T, y.ptr, idx.ptr, flag.ptr,
                  ws.ptr, ws.nbytes, _S())
        self.retain_inputs((0, 1))
        self._saved = (idx, flag, B, T, Cout, q, K, b is not None)
        return y,

    def backward(self, indexes, gys):
        idx, flag, B, T, Cout, q, K, has_b = self._saved
        x, W = [v.data for v in self.get_retained_inputs()]
        gy = gys[0].data
        gW, gb = _embed_wgrad(self, x, idx, flag, gy, B, Cout, q, K, T, has_b)
        gx = None
        if 0 in indexes:                       # the input is data on the training path; kept for completeness
            desc = _conv_desc(B, q, T, Cout, T, K, 1, K - 1, 1, False)
            gx = DeviceArray(x.shape, np.float32)
            wsd = backend.workspace(_lib.load().vqvae_conv1d_workspace_bytes(C.byref(desc)))
            _lib.call('vqvae_conv1d_bwd_data', C.byref(desc), W.ptr, gy.ptr, gx.ptr, 0, wsd.ptr, wsd.nbytes, _S())
        return (gx, gW, gb) if has_b else (gx, gW)


def embed_conv_onehot(x, W, b=None):
    """Causal (K,1) conv, pad K-1, cropped to the input length, of a (possibly) one-hot input."""
    args = (x, W) if b is None else (x, W, b)
    return EmbedConvOneHot().apply(args)[0]


def embed_conv_indices(idx, W, b=None):
    idx = as_variable(idx)
    idx.requires_grad = False
    args = (idx, W) if b is None else (idx, W, b)
    return EmbedConvFromIndices().apply(args)[0]
