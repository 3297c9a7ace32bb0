"""WaveNet decoder -- mirrors the reference's WaveNet/modules.py class for class
(ResidualBlock modules.py:7-74, ResidualNet 77-110, WaveNet 113-160) with the
same constructor and call signatures.  The per-block graph of ~10 Chainer
functions (dilated conv, crop, 1x1 conv, add, split, tanh, sigmoid, mul, two
1x1 convs, add) is one fused FunctionNode backed by vqvae_resblock_fwd/bwd.

The queue-based incremental generation (modules.py:58-74, 98-110, 232-255) is exposed at the
WaveNet level -- initialize(n) / generate(x, condition) / generate_sequence(...) -- with the
per-block push/pop fused into one device-side step (generation.py, csrc/generate.hip).
"""
import ctypes as C
import os

import numpy as np

from . import _lib, backend, core, functions as F, links as L
from .backend import DeviceArray
from .core import Chain, ChainList, FunctionNode, type_expect

_S = backend.stream
DIL_WGRAD_GROUP = 5      # blocks per batched dilated-conv weight-gradient launch
PACK_ONCE = True         # weight slabs of the whole chain re-laid once per step (False: per call; test_pack_once_equals_pack_per_call)


def _p(a):
    return None if a is None else a.ptr


def _rb_desc(x, cond, Wd, Ws, dilation):
    B, Cr, T = x.shape[:3]
    Cd, _, K = Wd.shape[:3]
    return _lib.ResblockDesc(B, T, Cr, Cd, Ws.shape[0], cond.shape[1], K, dilation)


def _rb_workspace(desc):
    return backend.workspace(_lib.load().vqvae_resblock_workspace_bytes(C.byref(desc)))


class ResidualBlockFunction(FunctionNode):
    """(x, condition, Wd, bd, Wc, bc, Wr, br, Ws, bs) -> (residual, skip);
    modules.py:30-56 with dropout_zero_rate == 0."""

    def __init__(self, dilation):
        self.dilation = int(dilation)

    def check_type_forward(self, v):
        x, c = v[0], v[1]
        type_expect((x.ndim == 4 and c.ndim == 4, 'ResidualBlock: x, condition must be (B,C,T,1)'),
                    (x.shape[2] == c.shape[2] and x.shape[0] == c.shape[0],
                     'ResidualBlock: x %s and condition %s disagree' % (x.shape, c.shape)),
                    (v[2].shape[1] == x.shape[1], 'ResidualBlock: conv in-channels mismatch'),
                    (v[4].shape[1] == c.shape[1], 'ResidualBlock: condition_dim mismatch'))

    def forward(self, inputs):
        backend.require_device(*inputs)
        x, cond, Wd, bd, Wc, bc, Wr, br, Ws, bs = inputs
        d = _rb_desc(x, cond, Wd, Ws, self.dilation)
        self.desc = d
        prm = _lib.ResblockParams(Wd.ptr, bd.ptr, Wc.ptr, bc.ptr, Wr.ptr, br.ptr, Ws.ptr, bs.ptr)
        res = DeviceArray((d.B, d.Cr, d.T, 1), np.float32)
        skip = DeviceArray((d.B, d.Cs, d.T, 1), np.float32)
        self.gates = DeviceArray((d.B, d.Cd, d.T), np.float32)
        self.z = DeviceArray((d.B, d.Cd // 2, d.T), np.float32)
        ws = _rb_workspace(d)
        _lib.call('vqvae_resblock_fwd', C.byref(d), C.byref(prm), x.ptr, cond.ptr, None, res.ptr,
                  skip.ptr, 0, self.gates.ptr, self.z.ptr, ws.ptr, ws.nbytes, _S())
        self.retain_inputs(tuple(range(10)))
        return res, skip

    def backward(self, indexes, gys):
        ins = [v.data for v in self.get_retained_inputs()]
        x, cond, Wd, bd, Wc, bc, Wr, br, Ws, bs = ins
        d = self.desc
        g_res = None if gys[0] is None else gys[0].data
        g_skip = None if gys[1] is None else gys[1].data
        if g_skip is None:                      # skip unused: treat as zeros
            g_skip = backend.zeros((d.B, d.Cs, d.T, 1))
        prm = _lib.ResblockParams(Wd.ptr, bd.ptr, Wc.ptr, bc.ptr, Wr.ptr, br.ptr, Ws.ptr, bs.ptr)
        gx = DeviceArray(x.shape, np.float32) if 0 in indexes else None
        gc = DeviceArray(cond.shape, np.float32) if 1 in indexes else None
        gp = [DeviceArray(a.shape, np.float32) for a in ins[2:]]
        if g_res is None:                       # residual output unused (last block)
            gp[4] = gp[5] = None
        grd = _lib.ResblockGrads(*[_p(a) for a in gp])
        ws = _rb_workspace(d)
        _lib.call('vqvae_resblock_bwd', C.byref(d), C.byref(prm), x.ptr, cond.ptr, self.gates.ptr,
                  self.z.ptr, _p(g_res), g_skip.ptr, _p(gx), _p(gc), 0, None, C.byref(grd), 0,
                  ws.ptr, ws.nbytes, _S())
        return tuple([gx, gc] + gp)


def _groups(nb, size=_lib.MAX_STACK_GROUP):
    """[lo, hi) block ranges of at most ``size`` blocks (one resstack call each)."""
    return [(lo, min(nb, lo + size)) for lo in range(0, nb, size)]


def _value_key(params):
    """The parameters' buffers AND the values in them: pointers, the memory's value versions (backend._Block.wver: in-place
    writes through DeviceArray methods -- Link.copyparams, p.data.set(...) -- and recycled addresses), the load / layout /
    init epochs (serializers, arena adoption, lazily shaped parameters).  Optimizer steps are checked by the caller
    (ResidualNet._forward: a slab set packed ahead is only offered to the forward of the step it was packed in)."""
    return (tuple(p.ptr for p in params), tuple(p.wver for p in params),
            core.param_epoch('load'), core.param_epoch('layout'), core.param_epoch('init'))


def _pack_key(params, d0):
    """What slabs packed ahead (ResidualNet.prepack_async) are valid for: _value_key, the block geometry, the matmul mode."""
    return _value_key(params) + (d0.B, d0.T, d0.Cr, d0.Cd, d0.Cs, d0.Cc, d0.K, _lib.load().vqvae_get_matmul_dtype())


def _owner_steps(pvars):
    """The optimizer step count behind each parameter Variable (optimizers.Adam sets `_owner_step`); None without an owner."""
    out = []
    for v in pvars:
        f = getattr(v, '_owner_step', None)
        out.append(f() if f is not None else None)
    return tuple(set(out)) if len(set(out)) <= 1 else tuple(out)


def _build_cproj(params, nb, stream):
    """The latent-rate condition projection of all blocks as ONE conv: its weight (nb * Cd, Cc, 1, 1) and its bias, which
    also carries the dilated convs' biases (the lerp weights of a column sum to one, so they can ride in the projection:
    vqvae_resblock_cproj::P_has_bd).  ``params``: 8 arrays per block (ResidualBlock.param_list order)."""
    Cd, Cc = params[2].shape[0], params[2].shape[1]
    Wc_all = DeviceArray((nb * Cd, Cc, 1, 1), np.float32)
    bc_all = DeviceArray((nb * Cd,), np.float32)
    bd_all = DeviceArray((nb * Cd,), np.float32)
    for lo, hi in _groups(nb):
        _lib.call('vqvae_concat', Wc_all.ptr + lo * Cd * Cc * 4, _lib.ptr_array([params[8 * i + 2] for i in range(lo, hi)]),
                  hi - lo, Cd * Cc, stream)
        _lib.call('vqvae_concat', bc_all.ptr + lo * Cd * 4, _lib.ptr_array([params[8 * i + 3] for i in range(lo, hi)]),
                  hi - lo, Cd, stream)
        _lib.call('vqvae_concat', bd_all.ptr + lo * Cd * 4, _lib.ptr_array([params[8 * i + 1] for i in range(lo, hi)]),
                  hi - lo, Cd, stream)
    _lib.call('vqvae_elementwise', 0, nb * Cd, bc_all.ptr, bd_all.ptr, bc_all.ptr, 1.0, 1.0, stream)
    return Wc_all, bc_all, bd_all        # (bd_all: kept alive until the add has run)


def _pack_stack(params, d0, nb, stream):
    """Every block's weight slabs (forward and backward forms) re-laid for this step (vqvae_resstack_pack)."""
    per = _lib.load().vqvae_resstack_packed_bytes(C.byref(d0))
    packed = DeviceArray((nb * per // 4,), np.float32)
    prms = (_lib.ResblockParams * nb)(*[
        _lib.ResblockParams(*[params[8 * i + j].ptr for j in range(8)]) for i in range(nb)])
    has_res = (C.c_int * nb)(*([1] * (nb - 1) + [0]))
    _lib.call('vqvae_resstack_pack', C.byref(d0), nb, prms, has_res, packed.ptr, packed.nbytes, stream)
    return packed, per


FUSE_PULLBACK = True     # (tests A/B it through this attribute)
DEFER_WGRAD = True
PB_REDUCE_GROUP = 20      # blocks per vqvae_pullback_reduce_into launch (5 while the chain's weight gradients ran 90 us longer per launch; re-measured on the LDS-DMA kernel: 5 / 7 / 10 / 20 -> 14.17 / 14.15 / 14.14 / 14.09 ms per step, two rounds on one box)
DEFER_DIL_BLOCKS = 5      # how many of the blocks nearest the input keep their dilated-conv weight gradients for the side stream (a multiple of DIL_WGRAD_GROUP; 10 and 15 measured 0.05-0.1 ms slower: the tail they would run beside is full)
BATCH_PULLBACK = True
PREPACK_ASYNC = True
# 'bfloat16' mode: the chain's tensors the caller may keep in HBM as bf16 (x_l, gh_l, g_res_l: vqvae_resblock_desc.storage) are
# taken whenever the library offers them; wavenet.BF16_STORAGE = False leaves every one fp32 (operand rounding only: ADVICE r4)
BF16_STORAGE = True


def _grad_out(var, shape):
    """Gradient destination for input ``var``: its slot in the flat gradient arena
    when it is a Parameter that has no gradient yet, else a fresh array."""
    buf = var.grad_buffer() if hasattr(var, 'grad_buffer') else None
    if buf is not None and buf.size == int(np.prod(shape)):
        return buf.reshape(shape)
    return DeviceArray(shape, np.float32)


KEEP_CONTRACT_WORDS = False        # tests: keep a host copy of the last sweep's maxima / scale words in LAST_CONTRACT_WORDS
LAST_CONTRACT_WORDS = None


class ResidualStackFunction(FunctionNode):
    """All blocks of a ResidualNet in one node (modules.py:89-96).
    inputs: (x, condition, then 8 params per block) -> skip_connections.

    MI355X-first restructuring (same math, summation order aside):
      * skip_connections = sum_l skip_l(z_l) is ONE GEMM over K = n_blocks*Cd/2 at the
        end (vqvae_resstack_skip_fwd) instead of n_blocks read-modify-write passes;
      * the last block's unused residual branch is not computed;
      * backward keeps every block's gh in HBM (2.5 GB at B=16 -- 288 GB part) so the
        condition gradient sum_l Wc_l^T gh_l is ONE GEMM over K = n_blocks*Cd, and all
        skip-weight gradients share one launch (g_skip is their common operand)."""

    def __init__(self, dilations, relu_out=False, prepacked=None):
        self.dilations = [int(d) for d in dilations]
        self.prepacked = prepacked       # ResidualNet.prepack_async's (key, packed, stride, event), or None
        # relu_out: the F.relu WaveNet applies to the skip sum (modules.py:158) in the skip GEMM's epilogue; its backward
        # arrives done when the conv that reads the result masked its gradient (functions.FUSE_RELU_BWD)
        self.relu_out = bool(relu_out)

    def forward(self, inputs):
        backend.require_device(*inputs)
        x, cond = inputs[0], inputs[1]
        nb = len(self.dilations)
        assert len(inputs) == 2 + 8 * nb
        self.descs, self.saved = [], []
        if self.prepacked is not None:      # whatever becomes of the slabs, the main stream is behind the side stream's writes of them
            backend.wait_event(_S(), self.prepacked[3])
        h = x
        # condition projection at the latent rate (see vqvae_resblock_cproj in the header)
        self.lat = None
        if isinstance(cond, F.LazyUpsampled):
            self.lat = lat = cond.latent                       # (B, Cc, Tl)
            B, Cc, Tl = lat.shape
            Cd = inputs[2].shape[0]
            self.cproj_shape = (B, Cc, Tl)
            # the projection's weight / bias (the dilated convs' biases ride in the latter: the gate kernels then add no
            # bias, and their condition term is one more step of the contraction -- vqvae_resblock_cproj::P_has_bd /
            # P_amax) and its packed slabs: built on the side stream by ResidualNet.prepack_async when the step started, if
            # the parameters are the ones it saw; else here
            pre = self.prepacked
            cp = pre[4] if (pre is not None and pre[0][:5] == _value_key(inputs[2:])) else None
            self._cslab_b = None
            if cp is not None:
                self.Wc_all, bc_all = cp[0], cp[1]
                slab_f, self._cslab_b = (cp[3], cp[4]) if cp[5] == (B, Cc, Tl) else (None, None)
            else:
                self.Wc_all, bc_all, _bd = _build_cproj(inputs[2:], nb, _S())
                slab_f = None
            self.pdesc = _lib.Conv1dDesc(B, Cc, Tl, nb * Cd, Tl, 1, 1, 0, 1, 0)
            P_all = DeviceArray((B, nb * Cd, Tl), np.float32)
            P_amax = backend.new_amax()
            ws = backend.workspace(_lib.load().vqvae_conv1d_workspace_bytes(C.byref(self.pdesc)))
            _lib.call('vqvae_conv1d_fwd_amax', C.byref(self.pdesc), lat.ptr, self.Wc_all.ptr, bc_all.ptr,
                      P_all.ptr, ws.ptr, ws.nbytes, C.byref(_lib.Conv1dAmax(None, None, P_amax.ptr, _p(slab_f))), _S())
            tb = F.resize_tables(Tl, x.shape[2])
        self.packed = None
        self.amax = None
        if self.lat is not None and PACK_ONCE and _lib.load().vqvae_get_matmul_dtype() == 3:
            # matmul mode 3 (float32x2): the absolute maximum of every tensor of the chain travels with it as a
            # group of _lib.AMAX_SLOTS device uint32 (float bits) -- groups x_l | gh_l | g_res_l | g_skip; the kernels'
            # epilogues raise the words of what they store (atomicMax), only the tensors that arrive from outside
            # are scanned.  Behind them the SCALE words of the tensors kept pre-split (vqvae_resblock_desc.storage &
            # VQVAE_STORE_*_F16X2): x_l | gh_l again -- the bound each was split under, written by its producer
            self.amax = backend.zeros(((5 * nb + 1) * _lib.AMAX_SLOTS,), np.uint32)
            if getattr(x, 'amax', None) is not None:     # it travelled with the tensor
                _lib.call('vqvae_memcpy_d2d', self.amax.ptr, x.amax.ptr, 4 * _lib.AMAX_SLOTS, _S())
            else:
                _lib.call('vqvae_absmax', x.ptr, x.size, self.amax.ptr, _S())
        if self.lat is not None and PACK_ONCE:
            # every block's weight slabs (forward and backward forms), re-laid once for this step
            d0 = _rb_desc(x, cond, inputs[2], inputs[8], self.dilations[0])
            pre = self.prepacked
            self._skipws = None
            if pre is not None and pre[0] == _pack_key(inputs[2:], d0):
                # packed on the side stream while the encoder / quantiser / condition embed ran (ResidualNet.prepack_async)
                self.packed, self.packed_stride = pre[1], pre[2]
                self._skipws = pre[5]            # ... and the skip sum's slabs / bias sums (vqvae_resstack_skip_prepare)
            else:
                self.packed, self.packed_stride = _pack_stack(inputs[2:], d0, nb, _S())
        for i, dil in enumerate(self.dilations):
            Wd, bd, Wc, bc, Wr, br, Ws, bs = inputs[2 + 8 * i: 10 + 8 * i]
            d = _rb_desc(h, cond, Wd, Ws, dil)
            prm = _lib.ResblockParams(Wd.ptr, bd.ptr, Wc.ptr, bc.ptr, Wr.ptr, br.ptr, Ws.ptr, bs.ptr)
            last = (i == nb - 1)
            if self.packed is not None:
                # BASELINE configs[4] (matmul mode 'bfloat16'): the residual stream between the blocks kept as bf16 where
                # the library offers it (vqvae_resblock_desc.storage; the buffers stay fp32-sized, half used): every
                # block but the last stores its residual output that way, every block but the first reads it
                sup = _lib.load().vqvae_resblock_bf16_storage(C.byref(d)) if BF16_STORAGE else 0
                if sup & _lib.STORE_RES_BF16 and sup & _lib.STORE_X_BF16:
                    d.storage = (0 if last else _lib.STORE_RES_BF16) | (_lib.STORE_X_BF16 if i > 0 else 0)
                # matmul mode 'float32x2': the same stream kept PRE-SPLIT (fp16 hi | lo dwords under an a-priori bound:
                # its three readers stage it with two v_perm_b32 per element pair instead of splitting it again)
                if self.amax is not None:
                    sup = _lib.load().vqvae_resblock_f16x2_storage(C.byref(d))
                    if sup & _lib.STORE_RES_F16X2 and sup & _lib.STORE_X_F16X2:
                        d.storage = (0 if last else _lib.STORE_RES_F16X2) | (_lib.STORE_X_F16X2 if i > 0 else 0)
                    # ... and of tanh, sigmoid, z = tanh * sigmoid only the last two are saved (a third of the gate
                    # kernel's stores): the backward takes tanh = z / sigmoid
                    d.storage |= sup & _lib.STORE_GATES_SIG
            res = None if last else DeviceArray((d.B, d.Cr, d.T, 1), np.float32)
            gates = DeviceArray((d.B, d.Cd, d.T), np.float32)
            z = DeviceArray((d.B, d.Cd // 2, d.T), np.float32)
            ws = _rb_workspace(d)
            if self.lat is not None:
                cp = _lib.ResblockCproj(P_all.ptr + i * d.Cd * Tl * 4, nb * d.Cd * Tl, Tl,
                                        tb['v0'].ptr, tb['w0'].ptr, tb['w1'].ptr, 1, P_amax.ptr)
                if self.packed is not None:
                    am = None
                    if self.amax is not None:
                        xpre = bool(d.storage & _lib.STORE_X_F16X2)
                        am = C.byref(_lib.ResblockAmax(
                            self._slot(3 * nb + 1 + i) if xpre else self._slot(i), None if last else self._slot(i + 1),
                            None, None, None, None,
                            self._slot(i), None if last else self._slot(3 * nb + 1 + i + 1), None))
                    _lib.call('vqvae_resblock_fwd_packed', C.byref(d), C.byref(prm), h.ptr, C.byref(cp),
                              _p(res), gates.ptr, z.ptr, ws.ptr, ws.nbytes,
                              self.packed.ptr + i * self.packed_stride, am, _S())
                else:
                    _lib.call('vqvae_resblock_fwd', C.byref(d), C.byref(prm), h.ptr, None, C.byref(cp),
                              _p(res), None, 0, gates.ptr, z.ptr, ws.ptr, ws.nbytes, _S())
            else:
                _lib.call('vqvae_resblock_fwd', C.byref(d), C.byref(prm), h.ptr, cond.ptr, None,
                          _p(res), None, 0, gates.ptr, z.ptr, ws.ptr, ws.nbytes, _S())
            self.descs.append(d)
            self.saved.append((h, gates, z))
            h = res
        d = self.descs[0]
        skip = DeviceArray((d.B, d.Cs, d.T, 1), np.float32)
        if self.amax is not None:
            skip.amax = backend.new_amax()       # published by the skip sum's epilogue: the next conv's operand scale
        skipws = getattr(self, '_skipws', None)
        for lo, hi in _groups(nb):
            n = hi - lo
            zs = _lib.ptr_array([self.saved[i][2] for i in range(lo, hi)])
            if skipws is not None and n == nb:       # slabs and bias sums were prepared ahead (prepack_async): the GEMM only
                _lib.call('vqvae_resstack_skip_fwd_prepared', C.byref(d), n, zs, skip.ptr, 0, 1 if self.relu_out else 0,
                          skipws.ptr, skipws.nbytes, _p(skip.amax), _S())
                continue
            ws = backend.workspace(_lib.load().vqvae_resstack_workspace_bytes(C.byref(d), n))
            Ws = _lib.ptr_array([inputs[2 + 8 * i + 6] for i in range(lo, hi)])
            bs = _lib.ptr_array([inputs[2 + 8 * i + 7] for i in range(lo, hi)])
            _lib.call('vqvae_resstack_skip_fwd', C.byref(d), n, Ws, bs, zs, skip.ptr,
                      0 if lo == 0 else 1, 1 if (self.relu_out and hi == nb) else 0, ws.ptr, ws.nbytes, _p(skip.amax), _S())
        skip.relu_out = self.relu_out
        self._skip = skip if self.relu_out else None
        self.retain_inputs(tuple(range(len(inputs))))
        return skip,

    def _check_contract(self, nb, hpre):
        """'float32x2': every pre-split tensor of this sweep -- x_l (l >= 1) and, with `hpre`, gh_l -- against the bound it was
        split under (backend.f32x2_contract_violations; one launch, no host round trip unless backend.contract_debug())."""
        pairs = [(self._slot(3 * nb + 1 + l), self._slot(l)) for l in range(1, nb)
                 if self.descs[l].storage & _lib.STORE_X_F16X2]
        if hpre:
            pairs += [(self._slot(4 * nb + 1 + l), self._slot(nb + l)) for l in range(nb)]
        if not pairs:
            return
        global LAST_CONTRACT_WORDS
        if KEEP_CONTRACT_WORDS and backend._state.get('arena') is None:           # (tests: the words themselves, per block)
            backend.synchronize()
            LAST_CONTRACT_WORDS = (nb, self.amax.get().reshape(-1, _lib.AMAX_SLOTS).copy())
        sc = (C.c_void_p * len(pairs))(*[p[0] for p in pairs])
        am = (C.c_void_p * len(pairs))(*[p[1] for p in pairs])
        rep = backend.contract_report()
        _lib.call('vqvae_f32x2_contract_check', len(pairs), sc, am, backend.CONTRACT_LOG2_LIMIT, rep.ptr, _S())
        if backend.contract_debug() and backend._state.get('arena') is None:      # (never inside a recording: the read synchronises)
            r = backend.f32x2_contract_violations()
            before, backend._state['contract_seen'] = backend._state.get('contract_seen', 0), r['violations']
            if r['violations'] > before:
                raise FloatingPointError(
                    "float32x2: a pre-split tensor of ResidualNet's chain was split under a bound 2^%.1f above its maximum "
                    "(limit 2^%d, %d violation(s) so far): outside the mode's dynamic-range contract (DESIGN.md 3a) -- use "
                    "set_matmul_dtype('float32x3') or backend.set_presplit(4)" % (r['worst_log2'], r['log2_limit'], r['violations']))

    def _slot(self, i):
        """Device address of group i of self.amax (x_l: l, gh_l: nb + l, g_res_l: 2 nb + l, g_skip: 3 nb; scale words of
        a pre-split x_l: 3 nb + 1 + l, of a pre-split gh_l: 4 nb + 1 + l)."""
        return self.amax.ptr + 4 * _lib.AMAX_SLOTS * i

    def backward(self, indexes, gys):
        in_vars = self.get_retained_inputs()
        ins = [v.data for v in in_vars]
        cond = ins[1]
        lat = self.lat
        g_skip = gys[0].data
        if self.relu_out and not getattr(g_skip, 'relu_masked', False):     # the fused ReLU's backward, unless the reader did it
            g0 = g_skip
            g_skip = F._ew(_lib.EW_RELU_BWD, g0, self._skip)
            g_skip.amax = getattr(g0, 'amax', None)
        nb = len(self.dilations)
        f16 = self.amax is not None
        if f16:
            if getattr(g_skip, 'amax', None) is not None:
                _lib.call('vqvae_memcpy_d2d', self._slot(3 * nb), g_skip.amax.ptr, 4 * _lib.AMAX_SLOTS, _S())
            else:
                _lib.call('vqvae_absmax', g_skip.ptr, g_skip.size, self._slot(3 * nb), _S())
        # BASELINE configs[4] (matmul mode 'bfloat16'): the tensors of the backward chain the library can keep in HBM as
        # bf16 (vqvae_resblock_desc.storage); the buffers below stay fp32-sized, half used
        store = 0
        if self.packed is not None and lat is not None and BF16_STORAGE:
            store = _lib.load().vqvae_resblock_bf16_storage(C.byref(self.descs[0])) & _lib.STORE_GH_BF16
        if f16 and self.packed is not None and lat is not None:    # matmul mode 'float32x2': gh kept pre-split
            store = _lib.load().vqvae_resblock_f16x2_storage(C.byref(self.descs[0])) & _lib.STORE_GH_F16X2
        hpre = bool(store & _lib.STORE_GH_F16X2)
        stream16 = (_lib.STORE_X_BF16 | _lib.STORE_RES_BF16 | _lib.STORE_X_F16X2 | _lib.STORE_RES_F16X2       # the forward's choice for the residual stream stays
                    | _lib.STORE_GATES_SIG)                                                                   # ... and for the saved gate values
        gstream = 0                                              # ... and its counterpart, the gradient stream g_res_l = gx_{l+1}
        if self.packed is not None and lat is not None and BF16_STORAGE:
            sup = _lib.load().vqvae_resblock_bf16_storage(C.byref(self.descs[0]))
            if sup & _lib.STORE_GX_BF16 and sup & _lib.STORE_GRES_BF16:
                gstream = _lib.STORE_GX_BF16 | _lib.STORE_GRES_BF16
        for l, dd in enumerate(self.descs):
            gs = (gstream & _lib.STORE_GX_BF16 if l >= 1 else 0) | (gstream & _lib.STORE_GRES_BF16 if l <= nb - 2 else 0)
            dd.storage = (dd.storage & stream16) | store | gs
        grads = [None] * len(ins)
        g_res = None
        ghs = [None] * nb
        g_ress = [None] * nb
        # Two streams.  The backward-data chain (gz -> gh -> gx, block by block) is serial and
        # each of its kernels leaves a residency tail (1920 workgroups on 1024 slots); the
        # weight gradients of block l only need gh_l, so they -- and the pull-back of gh_l to
        # the latent rate -- run on the side stream and fill those tails.
        overlap = backend.overlap_enabled()
        side = backend.side_stream() if overlap else _S()
        slot = 'side' if overlap else 'main'
        d0 = self.descs[0]
        gP = tb = None
        if lat is not None:
            Bl, Cc, Tl = lat.shape
            tb = F.resize_tables(Tl, d0.T)
            gP = DeviceArray((Bl, nb * d0.Cd, Tl), np.float32)
        # 'float32x2': the pull-back of every gh_l to the latent rate runs inside the launch that produces gh_l
        # (vqvae_resblock_amax.pb_part) and one reduce launch finishes all blocks; FUSE_PULLBACK = False: a launch per block
        pb_part = None
        # (measured in the bf16 mode too: the gate-derivative kernel of that mode loses more than the launch costs -- configs[4]
        # 18.8 -> 20.4 ms -- so it keeps its pull-back launches)
        if (f16 and lat is not None and self.packed is not None and FUSE_PULLBACK and d0.Cd == 256
                and d0.T % 128 == 0 and d0.T >= 64 * Tl):
            pb_part = DeviceArray((nb, d0.B, d0.T // 128, d0.Cd, 4), np.float32)
        # where the pull-back stays a kernel of its own (bf16 mode, float32x3, fp32 MFMA): ONE launch over all blocks' gh when
        # the chain is done (vqvae_upsample_linear_bwd_blocks) instead of a launch per block -- the gh_l then are the
        # slices of one array.  BATCH_PULLBACK = False: a launch per block, as before (same sums in the same order)
        gh_all = None
        if (BATCH_PULLBACK and pb_part is None and lat is not None and self.packed is not None and not overlap
                and not (store & _lib.STORE_GH_F16X2) and Tl >= 3 and d0.T >= 8 * Tl and d0.T % 4 == 0 and d0.T // 4 <= 2048
                and all((dd.B, dd.Cd, dd.T) == (d0.B, d0.Cd, d0.T) for dd in self.descs)):
            gh_all = DeviceArray((nb, d0.B, d0.Cd, d0.T), np.float32)
        # side-stream scratch, sized once for everything it will run (never regrown mid-flight)
        # res-conv weight gradients: one batched launch on the MAIN stream after the chain (it
        # then overlaps with whatever the side stream still has queued); balances the two queues
        grp = nb + 1
        # dilated-conv weight gradients: DIL_GROUP blocks per launch (each block is filter_size
        # segments of one contraction), so the K splits and their partial slabs are shared by
        # DIL_GROUP * 8 output tiles instead of 8 -- 5x less slab traffic for the reduce
        dil_group = max(1, min(DIL_WGRAD_GROUP, _lib.MAX_STACK_GROUP // d0.K))
        need = max(_lib.load().vqvae_resblock_workspace_bytes(C.byref(d0)),
                   _lib.load().vqvae_resstack_workspace_bytes(C.byref(d0), min(nb, _lib.MAX_STACK_GROUP)),
                   _lib.load().vqvae_resstack_dil_wgrad_workspace_bytes(C.byref(d0), min(nb, dil_group)))
        ws_side = backend.workspace(need, slot)
        # The weight gradients that are still due when the chain has reached the first block -- the last group of dilated-conv
        # gradients and every res conv's -- need nothing that comes later, and what comes later is the latent-rate tail of the
        # backward pass (pull-back reduce, condition embed, encoder, VQ: ~70 launches of 5-30 us on a handful of workgroups
        # each, 0.7 ms during which the chip is all but idle).  They go to the side stream and run BESIDE that tail;
        # Variable.backward() joins the streams when the sweep is done (backend.join_side).  DEFER_WGRAD = False: in line.
        defer = DEFER_WGRAD and not overlap and lat is not None and self.packed is not None
        if defer:          # (sized for everything this sweep sends to the side stream before the first of it is enqueued)
            need = max(need, _lib.load().vqvae_conv1d_workspace_bytes(C.byref(self.pdesc)))
        ws_defer = backend.workspace(need, 'side') if defer else None
        dil_pending = []          # blocks whose gh exists but whose dilated-conv wgrad is not issued yet
        gdil = {}                 # block -> (gWd, gbd) destinations

        def flush_dil(stream=None, ws=None):
            stream = side if stream is None else stream
            ws = ws_side if ws is None else ws
            if not dil_pending:
                return
            if len(dil_pending) > dil_group:          # (the deferred blocks: a launch per group, as in line)
                rest = dil_pending[dil_group:]
                del dil_pending[dil_group:]
                flush_dil(stream, ws)
                dil_pending.extend(rest)
                return flush_dil(stream, ws)
            blocks = list(dil_pending)
            del dil_pending[:]
            if overlap:
                backend.wait_event(side, backend.Event().record(_S()))    # their gh are complete
            # one launch per kind of x: the blocks whose input is the bf16 residual stream, and the rest (block 0)
            for x16 in (_lib.STORE_X_BF16, _lib.STORE_X_F16X2, 0):
                sel = [i for i in blocks if (self.descs[i].storage & (_lib.STORE_X_BF16 | _lib.STORE_X_F16X2)) == x16]
                if not sel:
                    continue
                dg = _lib.ResblockDesc.from_buffer_copy(d0)
                dg.storage = store | x16
                dils = (C.c_int * len(sel))(*[self.dilations[i] for i in sel])
                # (a pre-split operand is read under the bound it was split under: its scale words)
                xam = (C.c_void_p * len(sel))(*[self._slot(3 * nb + 1 + i if x16 == _lib.STORE_X_F16X2 else i)
                                                for i in sel]) if f16 else None
                gam = (C.c_void_p * len(sel))(*[self._slot(4 * nb + 1 + i if hpre else nb + i) for i in sel]) if f16 else None
                _lib.call('vqvae_resstack_dil_wgrad', C.byref(dg), len(sel), dils,
                          _lib.ptr_array([self.saved[i][0] for i in sel]),
                          _lib.ptr_array([ghs[i] for i in sel]),
                          _lib.ptr_array([gdil[i][0] for i in sel]),
                          _lib.ptr_array([gdil[i][1] for i in sel]), 0, ws.ptr, ws.nbytes,
                          xam, gam, stream)

        # skip-conv weight gradients need only g_skip and the saved z_l: start them right away
        gWs = [_grad_out(in_vars[2 + 8 * i + 6], ins[2 + 8 * i + 6].shape) for i in range(nb)]
        gbs = [_grad_out(in_vars[2 + 8 * i + 7], ins[2 + 8 * i + 7].shape) for i in range(nb)]
        gWr = [None] * nb
        gbr = [None] * nb
        if overlap:
            backend.wait_event(side, backend.Event().record(_S()))       # g_skip is ready
        for lo, hi in _groups(nb):
            zs = _lib.ptr_array([self.saved[i][2] for i in range(lo, hi)])
            _lib.call('vqvae_resstack_skip_wgrad', C.byref(d0), hi - lo, g_skip.ptr, zs,
                      _lib.ptr_array(gWs[lo:hi]), _lib.ptr_array(gbs[lo:hi]), 0, ws_side.ptr,
                      ws_side.nbytes, self._slot(3 * nb) if f16 else None, side)

        pending = [nb]          # res-conv weight gradients are issued for blocks [lo, pending)

        def flush_res(lo, stream=None, ws=None):
            """gWr_l, gbr_l for blocks lo..pending-1 (their g_res_l exist once the chain has
            passed block l+1)."""
            stream = _S() if stream is None else stream
            hi = pending[0]
            if hi <= lo:
                return
            for i in range(lo, hi):
                if g_ress[i] is not None:
                    gWr[i] = _grad_out(in_vars[2 + 8 * i + 4], ins[2 + 8 * i + 4].shape)
                    gbr[i] = _grad_out(in_vars[2 + 8 * i + 5], ins[2 + 8 * i + 5].shape)
            for glo, ghi in _groups(hi - lo):
                a, b = lo + glo, lo + ghi
                wsm = ws if ws is not None else backend.workspace(_lib.load().vqvae_resstack_workspace_bytes(C.byref(d0), b - a))
                zs = _lib.ptr_array([self.saved[i][2] for i in range(a, b)])
                ram = None
                if f16:      # g_res_l was published by block l + 1's backward-data launch
                    ram = (C.c_void_p * (b - a))(*[None if g_ress[i] is None else self._slot(2 * nb + i)
                                                   for i in range(a, b)])
                _lib.call('vqvae_resstack_res_wgrad', C.byref(d0), b - a, _lib.ptr_array(g_ress[a:b]), zs,
                          _lib.ptr_array(gWr[a:b]), _lib.ptr_array(gbr[a:b]), 0, wsm.ptr, wsm.nbytes, ram, stream)
            pending[0] = lo

        for i in range(nb - 1, -1, -1):
            Wd, bd, Wc, bc, Wr, br, Ws, bs = ins[2 + 8 * i: 10 + 8 * i]
            d = self.descs[i]
            h, gates, z = self.saved[i]
            prm = _lib.ResblockParams(Wd.ptr, bd.ptr, Wc.ptr, bc.ptr, Wr.ptr, br.ptr, Ws.ptr, bs.ptr)
            need_gx = (i > 0) or (0 in indexes)
            gx = DeviceArray(h.shape, np.float32) if need_gx else None
            gp = [_grad_out(in_vars[2 + 8 * i + j], ins[2 + 8 * i + j].shape) for j in range(4)]
            g_ress[i] = g_res                  # None for the last block: residual unused
            gh = (DeviceArray((d.B, d.Cd, d.T), np.float32) if gh_all is None else
                  gh_all.flat_view(i * d.B * d.Cd * d.T, d.B * d.Cd * d.T, (d.B, d.Cd, d.T)))
            ws = _rb_workspace(d)
            if lat is not None:
                # chain on the main stream: gz, gate derivative -> gh, then gx
                if self.packed is not None:
                    am = None
                    if f16:      # in: g_res_i, g_skip; out: gh_i and gx = g_res_{i-1}
                        am = C.byref(_lib.ResblockAmax(
                            None, None, None if g_res is None else self._slot(2 * nb + i), self._slot(3 * nb),
                            self._slot(nb + i), self._slot(2 * nb + i - 1) if (gx is not None and i > 0) else None,
                            None, None, self._slot(4 * nb + 1 + i) if hpre else None,
                            (pb_part.ptr + i * (pb_part.nbytes // nb)) if pb_part is not None else None,
                            tb['v0'].ptr, tb['w0'].ptr, tb['w1'].ptr, Tl))

                    _lib.call('vqvae_resblock_bwd_packed', C.byref(d), C.byref(prm), h.ptr, gates.ptr,
                              z.ptr, _p(g_res), g_skip.ptr, _p(gx), gh.ptr, ws.ptr, ws.nbytes,
                              self.packed.ptr + i * self.packed_stride, am, _S())
                else:
                    none = _lib.ResblockGrads(*([None] * 8))
                    _lib.call('vqvae_resblock_bwd', C.byref(d), C.byref(prm), h.ptr, None, gates.ptr,
                              z.ptr, _p(g_res), g_skip.ptr, _p(gx), None, 0, gh.ptr, C.byref(none), 0,
                              ws.ptr, ws.nbytes, _S())
                ghs[i] = gh
                gdil[i] = (gp[0], gp[1])
                dil_pending.append(i)
                if (len(dil_pending) >= dil_group or i == 0) and not (defer and i < DEFER_DIL_BLOCKS):
                    flush_dil()               # waits (on the side stream) for the chain up to here
                elif overlap:
                    backend.wait_event(side, backend.Event().record(_S()))
                if pb_part is not None or gh_all is not None:
                    pass                       # done in the gate-derivative launch's epilogue and reduced below / one launch below
                elif hpre:
                    _lib.call('vqvae_upsample_linear_bwd_f16x2', gh.ptr, d.Cd * d.T, d.B, d.Cd, Tl, d.T,
                              tb['w0'].ptr, tb['w1'].ptr, tb['lo0'].ptr, tb['hi0'].ptr, tb['lo1'].ptr,
                              tb['hi1'].ptr, gP.ptr + i * d.Cd * Tl * 4, nb * d.Cd * Tl, self._slot(4 * nb + 1 + i), side)
                else:
                    _lib.call('vqvae_upsample_linear_bwd_bf16' if store & _lib.STORE_GH_BF16 else 'vqvae_upsample_linear_bwd',
                              gh.ptr, d.Cd * d.T, d.B, d.Cd, Tl, d.T,
                              tb['w0'].ptr, tb['w1'].ptr, tb['lo0'].ptr, tb['hi0'].ptr, tb['lo1'].ptr,
                              tb['hi1'].ptr, gP.ptr + i * d.Cd * Tl * 4, nb * d.Cd * Tl, side)
                gp[2] = gp[3] = None           # condition_proj grads: one latent-rate conv, below
            else:
                grd = _lib.ResblockGrads(*([_p(a) for a in gp] + [None] * 4))
                _lib.call('vqvae_resblock_bwd', C.byref(d), C.byref(prm), h.ptr, cond.ptr, gates.ptr,
                          z.ptr, _p(g_res), g_skip.ptr, _p(gx), None, 0, gh.ptr, C.byref(grd), 0,
                          ws.ptr, ws.nbytes, _S())
            gp += [None, None, None, None]     # res / skip conv grads: batched launches
            grads[2 + 8 * i: 10 + 8 * i] = gp
            ghs[i] = gh
            g_res = gx
            if pending[0] - i >= grp:          # g_res of blocks i .. pending-1 are all available
                flush_res(i)
            # the fused pull-back's partial sums of the blocks the chain has passed, PB_REDUCE_GROUP at a time (round 5: ~15 us
            # between two chip-filling launches instead of one launch over all blocks in the tail -- 52 us alone, 160 us beside
            # the deferred weight gradients; round 6, with those 90 us shorter each: one launch behind the chain measures best)
            if pb_part is not None and (i % PB_REDUCE_GROUP == 0):
                hi_b = min(nb, i + PB_REDUCE_GROUP)
                _lib.call('vqvae_pullback_reduce_into', pb_part.ptr + i * (pb_part.nbytes // nb), tb['v0'].ptr, hi_b - i,
                          d0.B, d0.T, d0.Cd, Tl, gP.ptr + i * d0.Cd * Tl * 4, nb * d0.Cd * Tl, _S())
        d = self.descs[0]
        # weight gradients of the res convs of the blocks not yet covered (those nearest the input)
        if defer:
            sd = backend.side_stream()
            backend.wait_event(sd, backend.Event().record(_S()))       # every gh, g_res of the chain is complete
            flush_dil(sd, ws_defer)
            flush_res(0, sd, ws_defer)
            backend.defer_to_side([self.saved, ghs, g_ress, gdil, ws_defer, self.packed, self.amax, g_skip],
                                  writes=[a for pair in gdil.values() for a in pair] + gWr + gbr)
        else:
            flush_res(0)
        if overlap:
            # join: everything issued after this point on the main stream (and everything the
            # caller issues after backward returns) is ordered behind the side stream's work
            backend.wait_event(_S(), backend.Event().record(side))
        if pb_part is not None:
            pass                               # reduced group by group inside the chain (above)
        elif gh_all is not None:
            bf = 1 if store & _lib.STORE_GH_BF16 else 0
            n = d.B * d.Cd * d.T
            _lib.call('vqvae_upsample_linear_bwd_blocks', gh_all.ptr, bf, n * (2 if bf else 1), d.Cd * d.T, nb, d.B, d.Cd, Tl, d.T,
                      tb['w0'].ptr, tb['w1'].ptr, tb['lo0'].ptr, tb['hi0'].ptr, tb['lo1'].ptr, tb['hi1'].ptr,
                      gP.ptr, d.Cd * Tl, nb * d.Cd * Tl, _S())
        if lat is not None:
            # every block's gh has been pulled back to the latent rate (adjoint of the epilogue
            # lerp) on the side stream; the (nb*Cd, Cc) 1x1 conv's own backward then gives
            # gWc_l, gbc_l and the condition gradient
            B, Cc, Tl = lat.shape
            wsc = backend.workspace(_lib.load().vqvae_conv1d_workspace_bytes(C.byref(self.pdesc)))
            # the condition gradient first: it is what the rest of the sweep (condition embed, quantiser, encoder) waits
            # for; the projection's own weight gradient is a leaf -- on the side stream behind the other deferred ones
            if 1 in indexes:
                glat = DeviceArray(lat.shape, np.float32)
                _lib.call('vqvae_conv1d_bwd_data_amax', C.byref(self.pdesc), self.Wc_all.ptr, gP.ptr,
                          glat.ptr, 0, wsc.ptr, wsc.nbytes,
                          C.byref(_lib.Conv1dAmax(None, None, None, _p(getattr(self, '_cslab_b', None)))), _S())
                grads[1] = F.LatentGrad(cond.shape, glat)
            sw, wsw = _S(), wsc
            if defer:
                sw, wsw = backend.side_stream(), ws_defer
                backend.wait_event(sw, backend.Event().record(_S()))        # gP is complete
            gWc_all = DeviceArray(self.Wc_all.shape, np.float32)
            gbc_all = DeviceArray((nb * d.Cd,), np.float32)
            _lib.call('vqvae_conv1d_bwd_weight', C.byref(self.pdesc), lat.ptr, gP.ptr, gWc_all.ptr,
                      gbc_all.ptr, 0, wsw.ptr, wsw.nbytes, sw)
            gWc = [_grad_out(in_vars[2 + 8 * i + 2], ins[2 + 8 * i + 2].shape) for i in range(nb)]
            gbc = [_grad_out(in_vars[2 + 8 * i + 3], ins[2 + 8 * i + 3].shape) for i in range(nb)]
            for lo, hi in _groups(nb):
                _lib.call('vqvae_split', gWc_all.ptr + lo * d.Cd * Cc * 4, _lib.ptr_array(gWc[lo:hi]),
                          hi - lo, d.Cd * Cc, 0, sw)
                _lib.call('vqvae_split', gbc_all.ptr + lo * d.Cd * 4, _lib.ptr_array(gbc[lo:hi]),
                          hi - lo, d.Cd, 0, sw)
            if defer:
                backend.defer_to_side([gWc_all, gbc_all, gP, lat, gWc, gbc, wsw], writes=[gWc_all, gbc_all] + gWc + gbc)
            for i in range(nb):
                grads[2 + 8 * i + 2] = gWc[i]
                grads[2 + 8 * i + 3] = gbc[i]
        elif 1 in indexes:
            gcond = DeviceArray(cond.shape, np.float32)
            for lo, hi in _groups(nb):
                ws = backend.workspace(_lib.load().vqvae_resstack_workspace_bytes(C.byref(d), hi - lo))
                Wc = _lib.ptr_array([ins[2 + 8 * i + 2] for i in range(lo, hi)])
                _lib.call('vqvae_resstack_gcond_bwd', C.byref(d), hi - lo, Wc, _lib.ptr_array(ghs[lo:hi]),
                          gcond.ptr, 0 if lo == 0 else 1, ws.ptr, ws.nbytes, _S())
            grads[1] = gcond
        for i in range(nb):
            grads[2 + 8 * i + 4] = gWr[i]
            grads[2 + 8 * i + 5] = gbr[i]
            grads[2 + 8 * i + 6] = gWs[i]
            grads[2 + 8 * i + 7] = gbs[i]
        if f16 and self.packed is not None and lat is not None:
            self._check_contract(nb, hpre)
        self.saved = None                      # release activations
        self.packed = None
        self.amax = None
        grads[0] = g_res if 0 in indexes else None
        return tuple(grads)


class ResidualBlock(Chain):
    """modules.py:7-56."""

    def __init__(self, filter_size, dilation, residual_channels, dilated_channels, skip_channels,
                 condition_dim, dropout_zero_rate):
        super(ResidualBlock, self).__init__()
        with self.init_scope():
            self.conv = L.DilatedConvolution2D(
                residual_channels, dilated_channels, ksize=(filter_size, 1),
                pad=(dilation * (filter_size - 1), 0), dilate=(dilation, 1))
            self.condition_proj = L.Convolution2D(condition_dim, dilated_channels, 1)
            self.res = L.Convolution2D(dilated_channels // 2, residual_channels, 1)
            self.skip = L.Convolution2D(dilated_channels // 2, skip_channels, 1)
        self.filter_size = filter_size
        self.dilation = dilation
        self.residual_channels = residual_channels
        self.condition_dim = condition_dim
        self.dropout_zero_rate = dropout_zero_rate
        if dropout_zero_rate:
            raise NotImplementedError(
                'dropout_zero_rate > 0 (modules.py:34-35) is outside the hot-path scope: '
                'every BASELINE config trains with 0 (params.py:42)')

    def param_list(self):
        return [self.conv.W, self.conv.b, self.condition_proj.W, self.condition_proj.b,
                self.res.W, self.res.b, self.skip.W, self.skip.b]

    def __call__(self, x, condition):
        residual, skip_connection = ResidualBlockFunction(self.dilation).apply(
            [x, condition] + self.param_list())
        return residual, skip_connection


class ResidualNet(ChainList):
    """modules.py:77-96."""

    def __init__(self, n_loop, n_layer, filter_size, residual_channels, dilated_channels,
                 skip_channels, condition_dim, dropout_zero_rate):
        super(ResidualNet, self).__init__()
        dilations = [2 ** i for i in range(n_layer)] * n_loop
        for dilation in dilations:
            self.add_link(ResidualBlock(
                filter_size, dilation, residual_channels, dilated_channels, skip_channels,
                condition_dim, dropout_zero_rate))

    def __call__(self, x, condition):
        return self._forward(x, condition, False)

    def relu_call(self, x, condition):
        """F.relu(self(x, condition)) with the ReLU in the skip sum's epilogue (WaveNet.__call__, modules.py:158)."""
        return self._forward(x, condition, True)

    def _forward(self, x, condition, relu):
        blocks = list(self.children())
        args = [x, condition]
        for b in blocks:
            args += b.param_list()
        pre, self._prepacked = getattr(self, '_prepacked', None), None
        if pre is not None and pre[6] != _owner_steps(args[2:]):       # an optimizer step between the pack and this forward
            pre = None
        fn = ResidualStackFunction([b.dilation for b in blocks], relu_out=relu, prepacked=pre)
        out = fn.apply(args)[0]
        self._cproj_shape = getattr(fn, 'cproj_shape', None)     # (B, Cc, Tl) of the latent-rate projection: the next prepack_async packs for it
        return out

    def prepack_async(self, B, T, after=None):
        """The chain's weight slabs for an upcoming forward over (B, ., T), packed on the SIDE stream now: the 25 packing
        launches (~0.3 ms on a handful of workgroups) run beside the encoder / quantiser / condition-embed chain instead of in
        front of the first gate GEMM.  The forward takes the result if nothing changed (parameters, shapes, matmul mode), else
        packs as before.  ``after``: the main-stream event the side stream has to be behind (the optimizer's last write of the
        parameters); None: the main stream as it is now.  VAE.__call__ records that event when the step starts but calls this
        AFTER it has enqueued the encoder: launches reach the GPU in the order they are issued -- from Python and from a
        replayed hipGraph alike (ROCm enqueues a graph's nodes one by one, in capture order) -- so 25 side-stream launches
        issued in front of the first encoder conv delayed the whole critical path by the time it takes to ISSUE them."""
        if not (PACK_ONCE and PREPACK_ASYNC):
            return
        blocks = list(self.children())
        params = [p.data for b in blocks for p in b.param_list()]
        if any(not isinstance(p, DeviceArray) for p in params):
            return
        Wd, Wc, Ws = params[0], params[2], params[6]
        d0 = _lib.ResblockDesc(B, T, Wd.shape[1], Wd.shape[0], Ws.shape[0], Wc.shape[1], Wd.shape[2], blocks[0].dilation)
        side = backend.side_stream()
        backend.wait_event(side, after if after is not None else backend.Event().record(_S()))       # the optimizer's last write of the parameters
        packed, per = _pack_stack(params, d0, len(blocks), side)
        # ... and, when the last forward projected the condition at the latent rate, that projection's weight / bias
        # (three concats and an add over the blocks' parameters) and its packed slabs for both directions
        cp, shape = None, getattr(self, '_cproj_shape', None)
        if shape is not None:
            nb, Cd = len(blocks), Wc.shape[0]
            Wc_all, bc_all, bd_all = _build_cproj(params, nb, side)
            slab_f = slab_b = None
            if shape[0] == B and shape[1] == Wc.shape[1]:
                lib = _lib.load()
                descs = (_lib.Conv1dDesc * 2)()
                for i in range(2):
                    C.pointer(descs[i])[0] = _lib.Conv1dDesc(B, shape[1], shape[2], nb * Cd, shape[2], 1, 1, 0, 1, 0)
                slab_f = DeviceArray((int(lib.vqvae_conv1d_packed_bytes(C.byref(descs[0]), 0)) // 4,), np.float32)
                slab_b = DeviceArray((int(lib.vqvae_conv1d_packed_bytes(C.byref(descs[1]), 1)) // 4,), np.float32)
                _lib.call('vqvae_conv1d_pack', 2, descs, (C.c_void_p * 2)(Wc_all.ptr, Wc_all.ptr), (C.c_int * 2)(0, 1),
                          (C.c_void_p * 2)(slab_f.ptr, slab_b.ptr), side)
            cp = (Wc_all, bc_all, bd_all, slab_f, slab_b, tuple(shape) if slab_f is not None else None)
        # ... and the skip sum's weight slabs and bias sums (one contraction over all blocks: vqvae_resstack_skip_prepare)
        skipws = None
        if len(blocks) <= _lib.MAX_STACK_GROUP:
            nbytes = _lib.load().vqvae_resstack_workspace_bytes(C.byref(d0), len(blocks))
            skipws = DeviceArray((int(nbytes) // 4 + 1,), np.float32)
            _lib.call('vqvae_resstack_skip_prepare', C.byref(d0), len(blocks),
                      _lib.ptr_array([params[8 * i + 6] for i in range(len(blocks))]),
                      _lib.ptr_array([params[8 * i + 7] for i in range(len(blocks))]), skipws.ptr, skipws.nbytes, side)
        self._prepacked = (_pack_key(params, d0), packed, per, backend.Event().record(side), cp, skipws,
                           _owner_steps([p for b in blocks for p in b.param_list()]))


class WaveNet(Chain):
    """modules.py:113-160."""

    def __init__(self, n_loop, n_layer, filter_size, input_dim, residual_channels,
                 dilated_channels, skip_channels,
                 # arguments for output
                 quantize, use_logistic, n_mixture, log_scale_min,
                 # arguments for conditioning
                 condition_dim,
                 # arguments for dropout
                 dropout_zero_rate):
        super(WaveNet, self).__init__()
        with self.init_scope():
            self.embed = L.Convolution2D(input_dim, residual_channels, (2, 1), pad=(1, 0))
            self.resnet = ResidualNet(
                n_loop, n_layer, filter_size, residual_channels, dilated_channels, skip_channels,
                condition_dim, dropout_zero_rate)
            self.proj1 = L.Convolution2D(skip_channels, skip_channels, 1)
            output_dim = n_mixture if use_logistic else quantize
            self.proj2 = L.Convolution2D(skip_channels, output_dim, 1)
        self.input_dim = input_dim
        self.quantize = quantize
        self.use_logistic = bool(use_logistic)
        self.skip_channels = skip_channels
        self.log_scale_min = log_scale_min

    def calculate_logistic_loss(self, y, t):
        """modules.py:169-230 as one fused kernel pair."""
        return F.mixture_of_logistics_nll(y, t, self.quantize, self.log_scale_min)

    def embed_input(self, x):
        """The causal embed conv (modules.py:151-152) on whichever form of the input arrived."""
        if np.dtype(x.dtype) == np.int32:
            # device-side input pipeline: x holds mu-law bin indices (B, T) instead of the
            # one-hot (B, q, T, 1) tensor -- the embed conv is a gather of its weight columns
            return F.embed_conv_indices(x, self.embed.W, self.embed.b)
        if (self.input_dim >= 8 and self.embed.W.data is not None and x.shape[1] == self.embed.W.shape[1]
                and self.embed.pad[0] == self.embed.ksize[0] - 1 and self.embed.stride[0] == 1
                and _lib.load().vqvae_get_matmul_dtype() != 1):      # bf16 mode rounds W: dense kernels
            # the reference's one-hot float input (utils.py:85-87): same causal conv, but the device
            # checks for one-hot-ness and then gathers / bincounts instead of multiplying by zeros
            return F.embed_conv_onehot(x, self.embed.W, self.embed.b)
        # causal conv: pad 1 then crop to the input length (modules.py:151-152), fused as out_len
        return self.embed(x, out_len=x.shape[2])

    def __call__(self, x, condition, generating=False):
        # `generating` is accepted and ignored, as in modules.py:148-160
        x = self.embed_input(x)
        # residual & skip connections (modules.py:155)
        z = self.resnet.relu_call(x, condition)
        # output (modules.py:158-159); the ReLU after proj1 is fused into its epilogue
        z = self.proj1(z, relu=True)
        y = self.proj2(z)
        return y

    # ---- incremental generation (modules.py:232-255; generate.py:100-145) ----
    def initialize(self, n):
        """modules.py:232-244: fresh all-zero queues for ``n`` sequences (the reference supports
        n = 1, generate.py:42; here 1..4 run in lockstep)."""
        from .generation import GenerationState
        self._gen = GenerationState(self, n)

    def _gen_state(self):
        st = getattr(self, '_gen', None)
        if st is None:
            raise RuntimeError('call WaveNet.initialize(n) before generate (generate.py:100)')
        from . import core
        if st._epochs != (core.param_epoch('layout'), core.param_epoch('load')):
            raise RuntimeError('parameters were moved (optimizer.setup) or re-loaded (load_npz) since '
                               'WaveNet.initialize(n): call initialize(n) again')
        return st

    def generate(self, x, condition):
        """modules.py:246-255: one step.  x (n, input_dim, 1, 1), condition (n, condition_dim, 1, 1)
        -> logits Variable (n, out_dim, 1, 1); the queues advance by one sample."""
        from .core import Variable
        xd = x.data if isinstance(x, Variable) else x
        cd = condition.data if isinstance(condition, Variable) else condition
        backend.require_device(xd, cd)
        return Variable(self._gen_state().step_logits(xd, cd))

    def generate_sequence(self, condition, uniforms, n_steps=None, forced=None, return_logits=False,
                          graph_steps=8, persistent=True):
        """The loop of generate.py:101-145 as one device-resident run from fresh queues:
        condition (n, condition_dim, T[, 1]) on the device; ``uniforms`` the host doubles NumPy's
        RNG would hand generate.py:117 / 136 -- (T, n) for the softmax output, (T, n, nr_mix) for
        the mixture of logistics.  Returns the device array ``output`` (n, T): int32 mu-law bins or
        float32 samples, last column 0 as in generate.py:103-105 (plus the per-step logits
        (steps, n, out_dim) when ``return_logits``).  ``forced`` (T, n) feeds these values back
        instead of the samples (teacher forcing).  ``persistent`` runs the loop as one persistent
        kernel (csrc/generate.hip, channel counts <= 256); otherwise as a replayed hipGraph of
        per-step kernels (``graph_steps`` steps per graph, 0 = eager launches)."""
        from .core import Variable
        cd = condition.data if isinstance(condition, Variable) else condition
        backend.require_device(cd)
        self.initialize(cd.shape[0])
        mode = _lib.GEN_MOL if self.use_logistic else _lib.GEN_SOFTMAX
        return self._gen_state().run(cd, uniforms, mode, n_steps, forced, return_logits, graph_steps,
                                     persistent)

    def generate_batch(self, condition, uniforms, n_steps=None, group=4, max_streams=8):
        """Serving form: N sequences generated as concurrent groups of <= ``group`` lockstep sequences,
        one persistent launch per group on its own stream.  condition (N, condition_dim, T[, 1]) on the
        device, uniforms (T, N[, nr_mix]) host doubles; returns the host array (N, T).  Every sequence
        gets exactly the samples generate_sequence would give it alone with its column of uniforms."""
        from .core import Variable
        from .generation import run_many
        cd = condition.data if isinstance(condition, Variable) else condition
        backend.require_device(cd)
        mode = _lib.GEN_MOL if self.use_logistic else _lib.GEN_SOFTMAX
        return run_many(self, cd, uniforms, mode, n_steps, group, max_streams)
