"""ctypes binding of libwct_hip.so (include/wct_hip.h).

`cffi` is not installable here, so the "thin C-ABI layer" is ctypes.  The
library is loaded lazily; on a box with a GPU and no built library this raises
-- there is no CPU fallback behind these calls.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libwct_hip.so')

# every symbol include/wct_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int)
_U8 = C.POINTER(C.c_uint8)
_D = C.POINTER(C.c_double)
_PP = C.POINTER(C.c_void_p)
SIGNATURES = [
    ('wct_create', C.c_int, [C.c_int, _PP]),
    ('wct_destroy', None, [_P]),
    ('wct_last_error', C.c_char_p, []),
    ('wct_sync', C.c_int, [_P]),
    ('wct_device_count', C.c_int, [_I]),
    ('wct_get_stream', C.c_int, [_P, _PP]),
    ('wct_set_encoder', C.c_int, [_P, _F, _F, C.POINTER(_F), C.POINTER(_F), C.c_int]),
    ('wct_set_decoder', C.c_int, [_P, C.c_int, C.POINTER(_F), C.POINTER(_F), C.c_int]),
    ('wct_transform', C.c_int, [_P, _F, C.c_int, _F, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, _F, _I]),
    ('wct_adain', C.c_int, [_P, _F, C.c_int, _F, C.c_int, C.c_int, C.c_float, C.c_float, _F]),
    ('wct_style_swap', C.c_int, [_P, _F, C.c_int, C.c_int, _F, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_float, _F]),
    ('wct_set_style_swap', C.c_int, [_P, C.c_float, C.c_int, C.c_int]),
    ('wct_eigh', C.c_int, [_P, _F, C.c_int, C.c_int, _F, _F, _I]),
    ('wct_conv3x3', C.c_int, [_P, _F, C.c_int, C.c_int, C.c_int, _F, _F, C.c_int, C.c_int, C.c_int, _F]),
    ('wct_conv3x3_f16', C.c_int, [_P, _F, C.c_int, C.c_int, C.c_int, C.c_int, _F, _F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F]),
    ('wct_maxpool', C.c_int, [_P, _F, C.c_int, C.c_int, C.c_int, _F]),
    ('wct_encode', C.c_int, [_P, _F, C.c_int, C.c_int, C.c_int, _F]),
    ('wct_decode', C.c_int, [_P, _F, C.c_int, C.c_int, C.c_int, _F]),
    ('wct_coral_stats', C.c_int, [_P, _U8, C.c_int, C.c_int, _D]),
    ('wct_coral_apply', C.c_int, [_P, _U8, C.c_int, C.c_int, _D, _D, _D, _D, _D, _U8, _D]),
    ('wct_output_size', C.c_int, [C.c_int, C.c_int, _I, C.c_int, _I, _I]),
    ('wct_stylize', C.c_int, [_P, _U8, C.c_int, C.c_int, _U8, C.c_int, C.c_int, _I, C.c_int,
                              C.c_float, C.c_uint, _U8]),
    ('wct_stylize_batch_dev', C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, _I,
                                        C.c_int, C.c_float, C.c_uint, _P]),
    ('wct_train_step', C.c_int, [_P, C.c_int, _F, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _F]),
    ('wct_get_decoder_layer', C.c_int, [_P, C.c_int, C.c_int, _F, _F, _F, _F]),
    ('wct_train_grad_buffer', C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ('wct_train_apply', C.c_int, [_P, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int]),
    ('wct_dev_alloc', C.c_int, [_P, C.c_size_t, _PP]),
    ('wct_dev_free', C.c_int, [_P, _P]),
    ('wct_h2d', C.c_int, [_P, _P, _P, C.c_size_t]),
    ('wct_d2h', C.c_int, [_P, _P, _P, C.c_size_t]),
    ('wct_prof_enable', C.c_int, [_P, C.c_int]),
    ('wct_prof_reset', C.c_int, [_P]),
    ('wct_prof_read', C.c_int, [_P, _D, C.POINTER(C.c_longlong), _D, _D]),
    ('wct_eig_stats', C.c_int, [_P, C.POINTER(C.c_longlong)]),
]

WCT_NP, WCT_TF = 0, 1
FLAG_ADAIN, FLAG_MODE_NP, FLAG_SWAP5, FLAG_STYLE_SHARED, FLAG_IMAGES_F32 = 1, 2, 4, 8, 16
PROF_CLASSES = ['conv3x3', 'conv_first', 'conv_last', 'pool', 'wct_cov', 'jacobi', 'wct_apply', 'other', 'conv12', 'conv_wino', 'conv_tail']

_lib = None


class WCTHipError(RuntimeError):
    pass


class WCTNotConverged(WCTHipError):
    """WCT_STATUS_NOCONV: an eigendecomposition behind the call ran out of sweeps or met NaN/Inf (the reference's
    np.linalg.svd raises LinAlgError at the same spot, ops.py:110,123).  Outputs were written but are unreliable."""


STATUS_NOCONV = -5


def load():
    """Load libwct_hip.so and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise WCTHipError('%s not built: run `python -m wct_tf_amd.build` (hipcc, gfx950). '
                              'There is no CPU fallback for the stylize path.' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SIGNATURES:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        cls = WCTNotConverged if rc == STATUS_NOCONV else WCTHipError
        raise cls('libwct_hip error %d: %s' % (rc, load().wct_last_error().decode()))


def fptr(a):
    return a.ctypes.data_as(_F)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def ptr_array(arrays):
    """float** from a list of contiguous float32 arrays (keeps them alive via the return)."""
    arr = (_F * len(arrays))(*[fptr(a) for a in arrays])
    return arr
