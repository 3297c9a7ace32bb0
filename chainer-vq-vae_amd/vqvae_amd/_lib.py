"""ctypes binding of libvqvae_hip.so (include/vqvae_hip.h).

There is no CPU fallback: if the shared library is missing, importing anything
that needs it raises, and every compute entry point requires a gfx950 device.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'libvqvae_hip.so')

c_void_p, c_int, c_long, c_size_t = C.c_void_p, C.c_int, C.c_long, C.c_size_t
c_float, c_double, c_char_p = C.c_float, C.c_double, C.c_char_p
P = c_void_p                      # every device pointer crosses as an integer address
PP = C.POINTER(c_void_p)          # host array of device pointers


def ptr_array(arrays):
    """Host array of device pointers (None -> NULL) for the resstack entry points."""
    return (c_void_p * len(arrays))(*[None if a is None else a.ptr for a in arrays])


class Conv1dDesc(C.Structure):
    _fields_ = [(n, c_int) for n in
                ('B', 'Cin', 'Tin', 'Cout', 'Tout', 'K', 'stride', 'pad', 'dil', 'relu')]


class ResblockDesc(C.Structure):
    _fields_ = [(n, c_int) for n in ('B', 'T', 'Cr', 'Cd', 'Cs', 'Cc', 'K', 'dil', 'storage')]


class ResblockParams(C.Structure):
    _fields_ = [(n, P) for n in ('Wd', 'bd', 'Wc', 'bc', 'Wr', 'br', 'Ws', 'bs')]


class ResblockCproj(C.Structure):
    _fields_ = [('P', P), ('P_bstride', c_long), ('Tl', c_int), ('v0', P), ('w0', P), ('w1', P), ('P_has_bd', c_int), ('P_amax', P)]


class ResblockAmax(C.Structure):
    """vqvae_resblock_amax: device uint32 slots (float bit patterns of absolute maxima), matmul mode 3."""
    _fields_ = [(n, P) for n in ('x', 'res', 'g_res', 'g_skip', 'gh', 'gx', 'x_max', 'res_scale', 'gh_scale',
                                 'pb_part', 'pb_v0', 'pb_w0', 'pb_w1')] + [('pb_Tl', c_int)]


class Conv1dAmax(C.Structure):
    """vqvae_conv1d_amax"""
    _fields_ = [(n, P) for n in ('x', 'gy', 'out', 'packed')]


class ResblockGrads(C.Structure):
    _fields_ = [(n, P) for n in ('gWd', 'gbd', 'gWc', 'gbc', 'gWr', 'gbr', 'gWs', 'gbs')]


class GenBlock(C.Structure):
    _fields_ = [(n, P) for n in ('conv_W', 'conv_b', 'cond_W', 'cond_b', 'res_W', 'res_b',
                                 'skip_W', 'skip_b', 'ring')] + [('dilation', c_int)]


class GenDesc(C.Structure):
    _fields_ = ([(n, c_int) for n in ('n', 'n_blocks', 'input_dim', 'residual', 'dilated', 'skip',
                                      'cond_dim', 'out_dim', 'sample_mode')]
                + [('log_scale_min', c_float)]
                + [(n, P) for n in ('embed_W', 'embed_b', 'proj1_W', 'proj1_b', 'proj2_W', 'proj2_b')]
                + [('blocks', C.POINTER(GenBlock))]
                + [(n, P) for n in ('step', 'x_cur', 'x_prev', 'h0', 'h1', 'z', 'skip_acc', 's1',
                                    'logits', 'cond')]
                + [('cond_bstride', c_long), ('cond_cstride', c_long), ('cond_follows_step', c_int),
                   ('uniforms', P), ('n_uniform', c_int), ('forced_next', P), ('out', P),
                   ('out_bstride', c_long), ('logits_out', P), ('max_steps', c_int)])


# name -> (restype, argtypes); this table IS the list of symbols the header declares
PROTOTYPES = {
    'vqvae_last_error_string': (c_char_p, []),
    'vqvae_abi_version': (c_int, []),
    'vqvae_device_count': (c_int, [C.POINTER(c_int)]),
    'vqvae_set_device': (c_int, [c_int]),
    'vqvae_device_info': (c_int, [c_char_p, c_int, C.POINTER(c_int), C.POINTER(c_size_t)]),
    'vqvae_device_pci_bus_id': (c_int, [c_char_p, c_int]),
    'vqvae_malloc': (c_int, [C.POINTER(c_void_p), c_size_t]),
    'vqvae_free': (c_int, [P]),
    'vqvae_memcpy_h2d': (c_int, [P, c_void_p, c_size_t, P]),
    'vqvae_memcpy_d2h': (c_int, [c_void_p, P, c_size_t, P]),
    'vqvae_host_alloc': (c_int, [C.POINTER(c_void_p), c_size_t]),
    'vqvae_host_free': (c_int, [c_void_p]),
    'vqvae_memcpy_h2d_async': (c_int, [P, c_void_p, c_size_t, P]),
    'vqvae_memcpy_d2d': (c_int, [P, P, c_size_t, P]),
    'vqvae_memcpy2d_d2d': (c_int, [P, c_size_t, P, c_size_t, c_size_t, c_size_t, P]),
    'vqvae_memset': (c_int, [P, c_int, c_size_t, P]),
    'vqvae_stream_create': (c_int, [C.POINTER(c_void_p)]),
    'vqvae_stream_destroy': (c_int, [P]),
    'vqvae_stream_synchronize': (c_int, [P]),
    'vqvae_device_synchronize': (c_int, []),
    'vqvae_event_create': (c_int, [C.POINTER(c_void_p)]),
    'vqvae_event_destroy': (c_int, [P]),
    'vqvae_event_record': (c_int, [P, P]),
    'vqvae_event_synchronize': (c_int, [P]),
    'vqvae_stream_wait_event': (c_int, [P, P]),
    'vqvae_event_elapsed_ms': (c_int, [C.POINTER(c_float), P, P]),
    'vqvae_prof_enable': (c_int, [c_int]),
    'vqvae_prof_reset': (c_int, []),
    'vqvae_prof_read': (c_int, [c_int, C.POINTER(c_double), C.POINTER(c_int)]),
    'vqvae_set_matmul_dtype': (c_int, [c_int]),
    'vqvae_get_matmul_dtype': (c_int, []),
    'vqvae_set_wgrad_impl': (c_int, [c_int]),
    'vqvae_absmax': (c_int, [P, c_size_t, P, P]),
    'vqvae_set_f32x2_min_gflop': (c_int, [c_double]),
    'vqvae_conv1d_workspace_bytes': (c_size_t, [C.POINTER(Conv1dDesc)]),
    'vqvae_conv1d_fwd': (c_int, [C.POINTER(Conv1dDesc), P, P, P, P, P, c_size_t, P]),
    'vqvae_conv1d_bwd_data': (c_int, [C.POINTER(Conv1dDesc), P, P, P, c_int, P, c_size_t, P]),
    'vqvae_conv1d_bwd_weight': (c_int, [C.POINTER(Conv1dDesc), P, P, P, P, c_int, P, c_size_t, P]),
    'vqvae_conv1d_uses_f32x2': (c_int, [C.POINTER(Conv1dDesc)]),
    'vqvae_conv1d_packed_bytes': (c_size_t, [C.POINTER(Conv1dDesc), c_int]),
    'vqvae_conv1d_pack': (c_int, [c_int, C.POINTER(Conv1dDesc), C.POINTER(c_void_p), C.POINTER(c_int), C.POINTER(c_void_p), P]),
    'vqvae_conv1d_fwd_amax': (c_int, [C.POINTER(Conv1dDesc), P, P, P, P, P, c_size_t, C.POINTER(Conv1dAmax), P]),
    'vqvae_conv1d_bwd_data_amax': (c_int, [C.POINTER(Conv1dDesc), P, P, P, c_int, P, c_size_t, C.POINTER(Conv1dAmax), P]),
    'vqvae_conv1d_bwd_weight_amax': (c_int, [C.POINTER(Conv1dDesc), P, P, P, P, c_int, P, c_size_t,
                                             C.POINTER(Conv1dAmax), P]),
    'vqvae_resblock_workspace_bytes': (c_size_t, [C.POINTER(ResblockDesc)]),
    'vqvae_resblock_fwd': (c_int, [C.POINTER(ResblockDesc), C.POINTER(ResblockParams), P, P,
                                   C.POINTER(ResblockCproj), P, P, c_int, P, P, P, c_size_t, P]),
    'vqvae_resblock_bwd': (c_int, [C.POINTER(ResblockDesc), C.POINTER(ResblockParams), P, P, P, P,
                                   P, P, P, P, c_int, P, C.POINTER(ResblockGrads), c_int, P,
                                   c_size_t, P]),
    'vqvae_resblock_wgrad': (c_int, [C.POINTER(ResblockDesc), P, P, P, P, c_int, P, c_size_t, P]),
    'vqvae_resstack_packed_bytes': (c_size_t, [C.POINTER(ResblockDesc)]),
    'vqvae_resstack_pack': (c_int, [C.POINTER(ResblockDesc), c_int, C.POINTER(ResblockParams),
                                    C.POINTER(c_int), P, c_size_t, P]),
    'vqvae_resblock_fwd_packed': (c_int, [C.POINTER(ResblockDesc), C.POINTER(ResblockParams), P,
                                          C.POINTER(ResblockCproj), P, P, P, P, c_size_t, P,
                                          C.POINTER(ResblockAmax), P]),
    'vqvae_resblock_bwd_packed': (c_int, [C.POINTER(ResblockDesc), C.POINTER(ResblockParams), P, P, P,
                                          P, P, P, P, P, c_size_t, P, C.POINTER(ResblockAmax), P]),
    'vqvae_resstack_workspace_bytes': (c_size_t, [C.POINTER(ResblockDesc), c_int]),
    'vqvae_resstack_skip_fwd': (c_int, [C.POINTER(ResblockDesc), c_int, PP, PP, PP, P, c_int, c_int, P,
                                        c_size_t, P, P]),
    'vqvae_resstack_skip_prepare': (c_int, [C.POINTER(ResblockDesc), c_int, PP, PP, P, c_size_t, P]),
    'vqvae_resstack_skip_fwd_prepared': (c_int, [C.POINTER(ResblockDesc), c_int, PP, P, c_int, c_int, P, c_size_t, P, P]),
    'vqvae_conv1d_bwd_data_relu': (c_int, [C.POINTER(Conv1dDesc), P, P, P, P, P, c_size_t, C.POINTER(Conv1dAmax), P]),
    'vqvae_resstack_gcond_bwd': (c_int, [C.POINTER(ResblockDesc), c_int, PP, PP, P, c_int, P,
                                         c_size_t, P]),
    'vqvae_resstack_skip_wgrad': (c_int, [C.POINTER(ResblockDesc), c_int, P, PP, PP, PP, c_int, P,
                                          c_size_t, P, P]),
    'vqvae_resstack_res_wgrad': (c_int, [C.POINTER(ResblockDesc), c_int, PP, PP, PP, PP, c_int, P,
                                         c_size_t, PP, P]),
    'vqvae_resstack_dil_wgrad_workspace_bytes': (c_size_t, [C.POINTER(ResblockDesc), c_int]),
    'vqvae_resstack_dil_wgrad': (c_int, [C.POINTER(ResblockDesc), c_int, C.POINTER(c_int), PP, PP, PP, PP,
                                         c_int, P, c_size_t, PP, PP, P]),
    'vqvae_vq_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'vqvae_vq_nearest_fwd': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P,
                                     c_size_t, P]),
    'vqvae_vq_grad_w': (c_int, [P, P, c_int, c_int, c_int, c_int, P, c_int, P, c_size_t, P]),
    'vqvae_upsample_linear_fwd': (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P, P, c_long, P]),
    'vqvae_upsample_linear_bwd': (c_int, [P, c_long, c_int, c_int, c_int, c_int, P, P, P, P, P, P,
                                          P, c_long, P]),
    'vqvae_upsample_linear_bwd_blocks': (c_int, [P, c_int, c_long, c_long, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, c_long, c_long, P]),
    'vqvae_upsample_linear_bwd_bf16': (c_int, [P, c_long, c_int, c_int, c_int, c_int, P, P, P, P, P, P,
                                               P, c_long, P]),
    'vqvae_resblock_bf16_storage': (c_int, [C.POINTER(ResblockDesc)]),
    'vqvae_resblock_f16x2_storage': (c_int, [C.POINTER(ResblockDesc)]),
    'vqvae_set_presplit': (c_int, [c_int]),
    'vqvae_pullback_reduce': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    'vqvae_pullback_reduce_into': (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    'vqvae_f32x2_contract_check': (c_int, [c_int, PP, PP, c_int, P, P]),
    'vqvae_convstack_supported': (c_int, [c_int, c_int, c_int, C.POINTER(c_int)]),
    'vqvae_convstack_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vqvae_convstack_fwd': (c_int, [c_int, c_int, c_int, c_int, C.POINTER(c_int), P, PP, PP, PP, P]),
    'vqvae_convstack_bwd': (c_int, [c_int, c_int, c_int, c_int, C.POINTER(c_int), P, PP, PP, P, P, PP, PP, c_int, P, c_size_t, P]),
    'vqvae_conv_s2_bwd_supported': (c_int, [c_int] * 8),
    'vqvae_conv_s2_bwd_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vqvae_conv_s2_bwd': (c_int, [c_int, c_int, c_int, c_int, P, P, P, c_int, P, P, P, c_int, P, c_size_t, P]),
    'vqvae_upsample_linear_bwd_f16x2': (c_int, [P, c_long, c_int, c_int, c_int, c_int, P, P, P, P, P, P,
                                                P, c_long, P, P]),
    'vqvae_mulaw_bins': (c_int, [P, c_size_t, P, c_int, P, P]),
    'vqvae_onehot': (c_int, [P, c_long, c_int, c_int, c_int, P, P]),
    'vqvae_embed_gather_fwd': (c_int, [P, c_long, c_int, c_int, P, P, c_int, c_int, c_int, P, P]),
    'vqvae_embed_gather_bound': (c_int, [P, P, c_int, c_int, c_int, P, P]),
    'vqvae_embed_onehot_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'vqvae_embed_onehot_fwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, P, P, c_size_t, P]),
    'vqvae_embed_onehot_wgrad': (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P,
                                         c_size_t, P]),
    'vqvae_conv1d_fwd_cond': (c_int, [C.POINTER(Conv1dDesc), P, P, P, P, P, c_size_t, P, P]),
    'vqvae_conv1d_bwd_weight_cond': (c_int, [C.POINTER(Conv1dDesc), P, P, P, P, c_int, P, c_size_t, P, P]),
    'vqvae_concat': (c_int, [P, PP, c_int, c_size_t, P]),
    'vqvae_split': (c_int, [P, PP, c_int, c_size_t, c_int, P]),
    'vqvae_copy_list': (c_int, [c_int, C.POINTER(c_void_p), C.POINTER(c_void_p), C.POINTER(c_size_t), P]),
    'vqvae_embed_broadcast_fwd': (c_int, [P, P, c_int, c_int, c_int, P, c_long, P]),
    'vqvae_embed_broadcast_bwd': (c_int, [P, c_long, P, c_int, c_int, c_int, c_int, P, c_int, P,
                                          c_size_t, P]),
    'vqvae_softmax_xent_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'vqvae_softmax_xent_fwd': (c_int, [P, P, c_int, c_int, c_int, P, P, P, c_size_t, P]),
    'vqvae_softmax_xent_bwd': (c_int, [P, P, P, P, c_int, c_int, c_int, P, P]),
    'vqvae_softmax_xent_bwd_amax': (c_int, [P, P, P, P, c_int, c_int, c_int, P, P, P]),
    'vqvae_mol_nll_fwd': (c_int, [P, P, c_int, c_int, c_int, c_int, c_float, P, P, c_size_t, P]),
    'vqvae_mol_nll_bwd': (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, P, P]),
    'vqvae_elementwise': (c_int, [c_int, c_size_t, P, P, P, c_float, c_float, P]),
    'vqvae_sum': (c_int, [P, c_size_t, c_float, P, P, c_size_t, P]),
    'vqvae_sqdiff_mean': (c_int, [P, P, c_size_t, P, P, c_size_t, P]),
    'vqvae_sqdiff_mean_bwd': (c_int, [P, P, P, c_size_t, P, P, P]),
    'vqvae_adam_step': (c_int, [P, P, P, P, c_size_t, c_double, c_double, c_double, c_double, P]),
    'vqvae_adam_step_dev': (c_int, [P, P, P, P, c_size_t, P, P, c_double, c_double, c_double, P]),
    'vqvae_ema_step': (c_int, [P, P, c_size_t, c_double, P]),
    'vqvae_comm_unique_id': (c_int, [c_char_p]),
    'vqvae_comm_init': (c_int, [C.POINTER(c_void_p), c_int, c_int, c_char_p]),
    'vqvae_comm_allreduce_sum_f32': (c_int, [P, P, c_size_t, P]),
    'vqvae_comm_allreduce_max_f32': (c_int, [P, P, c_size_t, P]),
    'vqvae_comm_count': (c_int, [P, C.POINTER(c_int)]),
    'vqvae_comm_destroy': (c_int, [P]),
    'vqvae_wavenet_gen_step': (c_int, [C.POINTER(GenDesc), P]),
    'vqvae_wavenet_gen_run_workspace_bytes': (c_size_t, [C.POINTER(GenDesc)]),
    'vqvae_wavenet_gen_run': (c_int, [C.POINTER(GenDesc), c_int, c_int, P, c_size_t, P]),
    'vqvae_graph_capture_begin': (c_int, [P]),
    'vqvae_graph_capture_begin_relaxed': (c_int, [P]),
    'vqvae_graph_capture_end': (c_int, [P, C.POINTER(c_void_p)]),
    'vqvae_graph_launch': (c_int, [P, P]),
    'vqvae_graph_destroy': (c_int, [P]),
}

# elementwise op codes / profiler tags (mirror the header)
MAX_STACK_GROUP = 24              # blocks per resstack call (MAXSEG in csrc/gemm_common.h)
STORE_X_BF16, STORE_RES_BF16 = 2, 4
STORE_GX_BF16, STORE_GRES_BF16 = 8, 16
STORE_GH_BF16 = 1                 # vqvae_resblock_desc.storage bits (VQVAE_STORE_*)
STORE_GH_F16X2, STORE_X_F16X2, STORE_RES_F16X2 = 32, 64, 128      # matmul mode 3: kept pre-split (fp16 hi | lo dwords)
STORE_GATES_SIG = 256         # matmul mode 3: the gate kernel saves sigmoid and z only, the backward takes tanh = z / sigmoid
AMAX_SLOTS = 16                   # uint32 words per absolute maximum (vqvae_absmax, vqvae_resblock_amax)
EW_ADD, EW_SUB, EW_MUL, EW_AXPBY, EW_SCALE, EW_SQUARE, EW_RELU, EW_RELU_BWD, EW_FILL, \
    EW_MUL_SCALAR_DEV = range(10)
PROF_RESBLOCK_GATE, PROF_RESBLOCK_OUT, PROF_RESBLOCK_BWD_GZ, PROF_RESBLOCK_BWD_GX, \
    PROF_RESBLOCK_BWD_GC, PROF_RESBLOCK_WGRAD, PROF_CONV_FWD, PROF_CONV_BWD_DATA, \
    PROF_CONV_WGRAD, PROF_VQ_NEAREST, PROF_RESSTACK_SKIP, PROF_WGRAD_DIL, PROF_WGRAD_RES_SKIP = range(1, 14)
GEN_NONE, GEN_SOFTMAX, GEN_MOL = range(3)
GEN_MAX_N = 4


class HipError(RuntimeError):
    pass


_lib = None


ABI_VERSION = 5        # include/vqvae_hip.h: vqvae_abi_version()


def load():
    """Loads libvqvae_hip.so (once).  Raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libvqvae_hip.so not found at %s -- build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` (hipcc, gfx950). '
            'There is no CPU fallback.' % LIB_PATH)
    # the host driver only supports dmabuf IPC: RCCL's peer-memory exchange fails with the legacy mode.
    # Must be in the environment before the HSA runtime initialises (first HIP call).
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)           # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.vqvae_abi_version() != ABI_VERSION:    # the ctypes structs below mirror ONE layout of include/vqvae_hip.h
        raise ImportError('libvqvae_hip.so at %s has ABI %d, this package binds ABI %d: rebuild it (__graft_entry__.build())'
                          % (LIB_PATH, lib.vqvae_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().vqvae_last_error_string()
        raise HipError('%s failed (code %d): %s' % (what or 'libvqvae_hip call', rc,
                                                    msg.decode() if msg else ''))


def call(name, *args):
    """Calls an int-returning entry point and raises HipError on failure."""
    check(getattr(load(), name)(*args), name)
