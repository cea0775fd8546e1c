"""ctypes binding of libconvnet_hip.so (C ABI declared in include/convnet_hip.h).

This is the drop-in boundary: everything above it is Python host code mirroring the reference's
operator / trainer interface, everything below it is hand-written HIP for gfx950.  The library is
built in-tree by ``__graft_entry__.build()`` (or ``csrc/build.sh``).

There is deliberately NO fallback: if the HIP library is missing, loading raises.  The only other
library this module can bind is the TEST-ONLY SIMT emulator build of the *same kernel sources*
(``libconvnet_emul.so``), and only when ``CONVNET_AMD_EMULATE=1`` is set explicitly (the CPU test
suite does that); it is refused whenever a GPU is visible.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CONVNET_AMD_HIP_LIB: alternative build of the same library for A/B measurement (csrc/build.sh CN_LIB_NAME)
HIP_LIB = os.path.join(_HERE, os.environ.get('CONVNET_AMD_HIP_LIB', 'libconvnet_hip.so'))
EMUL_LIB = os.path.join(_HERE, 'libconvnet_emul.so')

F32, BF16, F16 = 0, 1, 2

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float
c_ll = ctypes.c_longlong
c_sz = ctypes.c_size_t
c_ull = ctypes.c_ulonglong

# name -> (restype, argtypes); mirrors include/convnet_hip.h one to one
_SIGNATURES = {
    'cn_last_error': (ctypes.c_char_p, []),
    'cn_build_info': (ctypes.c_char_p, []),
    'cn_last_kernel_name': (ctypes.c_char_p, []),
    'cn_kernel_log': (ctypes.c_char_p, [c_i]),
    'cn_is_emulator': (c_i, []),
    'cn_set_option': (c_i, [ctypes.c_char_p, c_i]),
    'cn_stream_fork': (c_i, [c_p, c_p]),
    'cn_stream_arm': (c_i, []),
    'cn_step_timer_mark': (c_i, [c_p, ctypes.c_longlong]),
    'cn_step_timer_poll': (c_i, [c_p, c_p, c_p]),
    'cn_stream_disarm': (c_i, []),
    'cn_plan_begin': (c_i, [c_p, c_p]),
    'cn_plan_end': (c_i, [c_p]),
    'cn_plan_import_graph': (c_i, [c_p, c_p]),
    'cn_plan_replay': (c_i, [c_p]),
    'cn_plan_info': (c_i, [c_p, c_p]),
    'cn_plan_bind_input': (c_i, [c_p, c_i, c_p, c_sz]),
    'cn_plan_set_input': (c_i, [c_p, c_i, c_p]),
    'cn_plan_describe': (ctypes.c_char_p, [c_p]),
    'cn_plan_destroy': (c_i, [c_p]),
    'cn_stream_wait_mark': (c_i, [c_i, c_p]),
    'cn_conv2d_fwd': (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 11 + [c_i, c_i, c_i, c_p]),
    'cn_conv2d_bnstats_rows': (c_i, [c_ll]),
    'cn_conv2d_fwd_bnstats': (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 11 + [c_i, c_i, c_p, c_i, c_p]),
    'cn_conv2d_fwd_bnstats_centered': (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 11 + [c_i, c_i, c_p, c_i, c_p, c_p]),
    'cn_conv2d_dgrad': (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 11 + [c_i, c_i, c_p]),
    'cn_conv2d_dgrad_junction_ok': (c_i, [c_i, c_i, c_i]),
    'cn_conv2d_dgrad_junction_rows': (c_i, [c_i] * 4),
    'cn_conv2d_dgrad_junction': (c_i, [c_p, c_p, c_p, c_p, c_i] + [c_i] * 6 + [c_p, c_p, c_p, c_p, c_i, c_p]),
    'cn_conv2d_dgrad_bnbwd_rows': (c_i, [c_i] * 6),
    'cn_conv2d_dgrad_bnbwd': (c_i, [c_p, c_p, c_p, c_p] + [c_i] * 11 + [c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    'cn_conv2d_dgrad_sa': (c_i, [c_p, c_p, c_p, c_p, c_i] + [c_i] * 11 + [c_i, c_i, c_p]),
    'cn_conv2d_dgrad_bnbwd_sa': (c_i, [c_p, c_p, c_p, c_p, c_i] + [c_i] * 11 + [c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    'cn_conv2d_wgrad_workspace': (c_sz, [c_i] * 12),
    'cn_conv2d_wgrad': (c_i, [c_p, c_p, c_p, c_i] + [c_i] * 11 + [c_i, c_f, c_f, c_p, c_sz, c_p]),
    'cn_conv2d_fwd_lazyz': (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p] + [c_i] * 6 + [c_p, c_i, c_p, c_p]),
    'cn_conv2d_dgrad_lazy': (c_i, [c_p, c_p, c_p, c_p, c_p] + [c_i] * 11 + [c_i, c_p]),
    'cn_conv2d_wgrad_lazy': (c_i, [c_p, c_p, c_p, c_p, c_p, c_i] + [c_i] * 11 + [c_i, c_f, c_f, c_p, c_sz, c_p]),
    'cn_conv2d_bwd1x1_lazy_ok': (c_i, [c_i, c_i, c_i]),
    'cn_conv2d_bwd1x1_lazy_workspace': (c_sz, [c_i] * 5),
    'cn_conv2d_bwd1x1_lazy': (c_i, [c_p] * 7 + [c_i] * 6 + [c_f, c_f, c_p, c_sz, c_p]),
    'cn_stem_fwd_ok': (c_i, [c_i] * 5),
    'cn_stem_fwd_rows': (c_i, [c_i, c_i]),
    'cn_stem_fwd': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    'cn_stem_wgrad_ok': (c_i, [c_i] * 5),
    'cn_stem_wgrad_workspace': (c_sz, [c_i, c_i]),
    'cn_stem_wgrad': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_p, c_sz, c_p]),
    'cn_conv3x3_c64_ok': (c_i, [c_i] * 5),
    'cn_conv3x3_c64_rows': (c_i, [c_i, c_i]),
    'cn_conv3x3_c64': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p]),
    'cn_conv1x1_stream_fwd_ok': (c_i, [c_i, c_i, c_i]),
    'cn_conv1x1_stream_fwd_rows': (c_i, [c_i] * 4),
    'cn_conv1x1_stream_fwd': (c_i, [c_p, c_p, c_p] + [c_i] * 6 + [c_p, c_i, c_p]),
    'cn_conv2d_dgrad_lazy_stream_ok': (c_i, [c_i, c_i, c_i]),
    'cn_conv2d_dgrad_lazy_stream': (c_i, [c_p] * 5 + [c_i] * 6 + [c_p]),
    'cn_conv1x1_stream_fwd_lazya': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p] + [c_i] * 6 + [c_p, c_i, c_p]),
    'cn_conv3x3_c64_lazya': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p] + [c_i] * 4 + [c_p, c_i, c_p]),
    'cn_conv2d_dgrad_junction_rows_k': (c_i, [c_i] * 5),
    'cn_bn_workspace': (c_sz, [c_i, c_i, c_i]),
    'cn_bn_fwd_train': (c_i, [c_p] * 9 + [c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    'cn_bn_fwd_train_partials': (c_i, [c_p] * 9 + [c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_sz, c_p]),
    'cn_bn_fwd_train_partials_centered': (c_i, [c_p] * 9 + [c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_sz, c_p]),
    'cn_bn_fwd_infer': (c_i, [c_p] * 7 + [c_f, c_p, c_i, c_i, c_i, c_i, c_p]),
    'cn_bn_apply_dual': (c_i, [c_p] * 6 + [c_i, c_i, c_i, c_i, c_p]),
    'cn_bn_bwd': (c_i, [c_p] * 9 + [c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    'cn_bn_bwd_partials': (c_i, [c_p] * 7 + [c_f, c_f, c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_sz, c_p]),
    'cn_bn_local_sums': (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_sz, c_p]),
    'cn_bn_fwd_train_sums': (c_i, [c_p] * 9 + [c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_p, c_ll, c_p]),
    'cn_bn_bwd_local_sums': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_sz, c_p]),
    'cn_bn_bwd_sums': (c_i, [c_p] * 9 + [c_f, c_f, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_ll, c_p]),
    'cn_maxpool_fwd': (c_i, [c_p, c_p, c_p] + [c_i] * 8 + [c_p]),
    'cn_maxpool_bwd': (c_i, [c_p, c_p, c_p] + [c_i] * 8 + [c_p]),
    'cn_maxpool_fwd_bnrelu': (c_i, [c_p, c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p]),
    'cn_bn_bwd_maxpool': (c_i, [c_p] * 8 + [c_f, c_f, c_p] + [c_i] * 8 + [c_p, c_sz, c_p]),
    'cn_maxpool_fwd_bnrelu_xmax': (c_i, [c_p, c_p, c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p]),
    'cn_bn_bwd_maxpool_xmax': (c_i, [c_p] * 9 + [c_f, c_f, c_p] + [c_i] * 8 + [c_p, c_sz, c_p]),
    'cn_avgpool_fwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'cn_avgpool_bwd': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'cn_nchw_to_nhwc': (c_i, [c_p, c_p] + [c_i] * 6 + [c_p]),
    'cn_u8_nhwc_to_nchw_lut': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    'cn_resize_u8_crops': (c_i, [c_p] * 7 + [c_i] * 4 + [c_p]),
    'cn_nchw_to_pairs': (c_i, [c_p, c_p] + [c_i] * 6 + [c_p]),
    'cn_weight_prep_pairs': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'cn_wgrad_unpack_pairs': (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p]),
    'cn_nhwc_to_nchw': (c_i, [c_p, c_p] + [c_i] * 6 + [c_p]),
    'cn_eltwise': (c_i, [c_i, c_p, c_p, c_p, c_ll, c_i, c_p]),
    'cn_softmax_ce': (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_i, c_i, c_f, c_p, c_f, c_p]),
    'cn_sgd_momentum': (c_i, [c_p, c_p, c_p, c_ll, c_f, c_f, c_f, c_f, c_p, c_p, c_p]),
    'cn_grad_norm_workspace': (c_sz, []),
    'cn_grad_norm_clip': (c_i, [c_p, c_ll, c_f, c_f, c_p, c_p, c_f, c_p, c_p]),
    'cn_weight_prep': (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    'cn_weight_prep_multi': (c_i, [c_p, c_p, c_p, c_i, c_ll, c_i, c_p]),
    'cn_weight_prep_tiled': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    'cn_colsum_workspace': (c_sz, [c_i]),
    'cn_colsum': (c_i, [c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p, c_p]),
    'cn_small_linear': (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    'cn_cast_from_f32': (c_i, [c_p, c_p, c_ll, c_i, c_p]),
    'cn_fill_f32': (c_i, [c_p, c_ll, c_f, c_p]),
    'cn_minmax_workspace': (c_sz, [c_i, c_ll]),
    'cn_minmax_rows': (c_i, [c_p, c_i, c_ll, c_i, c_p, c_p, c_sz, c_p]),
    'cn_qparams': (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_f, c_p]),
    'cn_quantize': (c_i, [c_p, c_p, c_ll, c_i, c_p, c_p, c_i, c_p, c_i, c_ull, c_p]),
    'cn_quantize_s': (c_i, [c_p, c_p, c_ll, c_i, c_p, c_p, c_i, c_p, c_i, c_ull, c_p, c_p]),
    'cn_counter_inc': (c_i, [c_p, c_p]),
    'cn_quantize_levels': (c_i, [c_p, c_p, c_ll, c_i, c_p, c_p, c_i, c_p, c_i, c_ull, c_p, c_p]),
    'cn_rangebn_fwd_q8': (c_i, [c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_i,
                                c_p, c_p, c_sz, c_p]),
    'cn_rangebn_bwd_q8': (c_i, [c_p, c_p, c_i, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_i, c_p, c_p,
                                c_p, c_sz, c_p]),
    'cn_quantize_rows': (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    'cn_quantize_rows_multi': (c_i, [c_p, c_p, c_p, c_i, c_p]),
    'cn_rangebn_workspace': (c_sz, [c_i, c_i, c_i]),
    'cn_rangebn_fwd': (c_i, [c_p] * 7 + [c_f, c_f, c_i, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    'cn_rangebn_fwd_q': (c_i, [c_p, c_p, c_i] + [c_p] * 7 + [c_f, c_f, c_i, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_sz, c_p]),
    'cn_rangebn_bwd_mm': (c_i, [c_p] * 8 + [c_i, c_i, c_i, c_f, c_i, c_i, c_p, c_p, c_sz, c_p]),
    'cn_eltwise_mm_workspace': (c_sz, [c_ll, c_i, c_i]),
    'cn_eltwise_mm': (c_i, [c_i, c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_p, c_sz, c_p]),
    'cn_eltwise_mm_qp': (c_i, [c_i, c_p, c_p, c_p, c_ll, c_i, c_i, c_p, c_p, c_p, c_sz, c_p]),
    'cn_rangebn_bwd': (c_i, [c_p] * 8 + [c_i, c_i, c_i, c_f, c_i, c_p, c_sz, c_p]),
    'cn_i8_prepare_activation': (c_i, [c_p] * 5 + [c_i] * 11 + [c_p, c_p, c_p, c_p, c_i, c_p]),
    'cn_i8_prepare_weight': (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    'cn_conv2d_fwd_i8': (c_i, [c_p] * 10 + [c_i, c_p] + [c_i] * 12 + [c_p]),
    'cn_comm_load': (c_i, []),
    'cn_comm_unique_id': (c_i, [c_p]),
    'cn_comm_init': (c_i, [c_p, c_p, c_i, c_i]),
    'cn_comm_info': (c_i, [c_p, c_p, c_p, c_p]),
    'cn_comm_allreduce_bucket': (c_i, [c_p, c_p, c_ll, c_p, c_p, c_i]),
    'cn_comm_join': (c_i, [c_p, c_p]),
    'cn_comm_allreduce': (c_i, [c_p, c_p, c_ll, c_i, c_p]),
    'cn_comm_broadcast': (c_i, [c_p, c_p, c_ll, c_i, c_p]),
    'cn_comm_destroy': (c_i, [c_p]),
    'cn_probe_mfma_bf16': (c_i, [c_p, c_p, c_p, c_p]),
    'cn_probe_mfma_f16': (c_i, [c_p, c_p, c_p, c_p]),
    'cn_probe_mfma_f32': (c_i, [c_p, c_p, c_p, c_p]),
    'cn_probe_tr16': (c_i, [c_p, c_p, c_p]),
    'cn_probe_mfma_i8': (c_i, [c_p, c_p, c_p, c_p]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None
_emulated = False


class ConvNetHipError(RuntimeError):
    pass


def _bind(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


def emulation_requested():
    return os.environ.get('CONVNET_AMD_EMULATE', '0') == '1'


def load():
    """Load (once) and return the bound library."""
    global _lib, _emulated
    if _lib is not None:
        return _lib
    if emulation_requested():
        if torch.cuda.is_available():
            raise ConvNetHipError('CONVNET_AMD_EMULATE=1 is refused when a GPU is visible: '
                                  'the emulator is a CPU-only test harness, not a product path')
        if not os.path.exists(EMUL_LIB):
            raise ConvNetHipError('emulator library missing: run csrc/build.sh emul')
        _lib = _bind(EMUL_LIB)
        _emulated = True
        _apply_env_options(_lib)
        return _lib
    if not os.path.exists(HIP_LIB):
        raise ConvNetHipError(
            'libconvnet_hip.so not found at %s -- build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` (hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback.' % HIP_LIB)
    _lib = _bind(HIP_LIB)
    _emulated = False
    _apply_env_options(_lib)
    return _lib


def _apply_env_options(lib):
    """CONVNET_AMD_OPTIONS="name=value,..." -> cn_set_option: kernel-variant tuning knobs for A/B
    measurement (results never change)."""
    for item in os.environ.get('CONVNET_AMD_OPTIONS', '').split(','):
        if '=' in item:
            k, v = item.split('=', 1)
            lib.cn_set_option(k.strip().encode(), int(v))


def source_hash():
    """Content hash of the library's sources + headers in THIS tree, by the recipe of csrc/build.sh (SRC_HASH): equal to
    the hash inside cn_build_info() exactly when the loaded binary was built from these files."""
    import glob
    import hashlib
    import re
    csrc = os.path.join(_HERE, 'csrc')
    srcs = re.search(r'^SRCS="([^"]+)"', open(os.path.join(csrc, 'build.sh')).read(), re.M).group(1).split()
    files = [os.path.join(csrc, f) for f in srcs] + sorted(glob.glob(os.path.join(csrc, '*.h'))) + \
        [os.path.join(_HERE, '..', 'include', 'convnet_hip.h')]
    h = hashlib.sha1()
    for f in files:
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_hash():
    """The source hash compiled into the loaded library ('unknown' for a build that bypassed csrc/build.sh)."""
    info = load().cn_build_info().decode()
    return info.rsplit('src ', 1)[1] if 'src ' in info else 'unknown'


def is_emulated():
    load()
    return _emulated


def last_error():
    msg = load().cn_last_error()
    return msg.decode() if msg else ''


def check(rc, what=''):
    if rc != 0:
        msg = load().cn_last_error()
        raise ConvNetHipError('%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else ''))


def dtype_code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:
        return F16
    raise ConvNetHipError('unsupported compute dtype %s (float32 / bfloat16 / float16 only)' % dtype)


def chunk_elems(dtype):
    return 4 if dtype == torch.float32 else 8


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream_of(t):
    """Raw hipStream_t of torch's current stream on the tensor's device (NULL for the emulator)."""
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    if not is_emulated():
        raise ConvNetHipError('tensor is on %s but the HIP library needs device memory' % t.device)
    return None


def require_device(t, name='tensor'):
    if not t.is_cuda and not is_emulated():
        raise ConvNetHipError('%s must live on a HIP device (got %s)' % (name, t.device))
    if not t.is_contiguous():
        raise ConvNetHipError('%s must be contiguous' % name)
