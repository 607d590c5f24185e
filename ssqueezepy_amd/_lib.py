# -*- coding: utf-8 -*-
"""ctypes binding of libssq_hip.so (C ABI declared in include/ssq_hip.h).

There is no CPU fallback: if the library cannot be loaded the import of the
compute layer fails loudly. `load()` builds the library with hipcc on first use
when the in-tree .so is missing (hipcc is part of the ROCm image).
"""
import ctypes
import os
from ctypes import (c_int, c_int64, c_double, c_void_p, c_char_p, POINTER,
                    Structure)

HERE = os.path.dirname(os.path.abspath(__file__))
# SSQ_HIP_LIB: load a library built elsewhere (e.g. an A/B build) instead of the in-tree one
LIB_PATH = os.environ.get('SSQ_HIP_LIB') or os.path.join(HERE, 'libssq_hip.so')

F32, F64 = 0, 1
GRID_LOG, GRID_LOG_PIECEWISE, GRID_LIN = 0, 1, 2
PAD = {None: -1, 'zero': 0, 'reflect': 1, 'symmetric': 2, 'replicate': 3,
       'wrap': 4}


class SsqError(RuntimeError):
    pass


class CwtDesc(Structure):
    _fields_ = [('dtype', c_int), ('padtype', c_int), ('n', c_int64),
                ('m', c_int64), ('n1', c_int64), ('na', c_int64),
                ('bank', c_void_p), ('band_off', c_void_p), ('band_lo', c_void_p),
                ('dt', c_double), ('row_scale', c_void_p), ('max_batch', c_int64),
                ('algo', c_int)]


class CwtBlocksDesc(Structure):
    _fields_ = [('n_classes', c_int), ('classes', c_void_p), ('rows', c_void_p),
                ('pbank', c_void_p), ('pxi', c_void_p), ('n_pbank', c_int64), ('ctw', c_void_p),
                ('ctw_off', c_void_p), ('ftw', c_void_p), ('n_ftw', c_int64),
                ('ftw_off', c_int64 * 5), ('items', c_void_p * 5),
                ('n_items', c_int64 * 5), ('generic_rows', c_void_p),
                ('n_generic', c_int64)]


class CwtTilesDesc(Structure):
    _fields_ = [('n_segs', c_int), ('segs', c_void_p), ('n_steps', c_int), ('rows', c_void_p),
                ('wtab', c_void_p), ('n_phases', c_int64),
                ('tbank', c_void_p), ('n_tbank', c_int64), ('n_irows', c_int),
                ('irows', c_void_p), ('n_classes', c_int), ('classes', c_void_p),
                ('u_total', c_int64), ('n_items_tile', c_int64 * 5), ('reserved', c_int)]


class StftDesc(Structure):
    _fields_ = [('dtype', c_int), ('padtype', c_int), ('n', c_int64),
                ('n_fft', c_int64), ('hop_len', c_int64), ('modulated', c_int),
                ('window', c_void_p), ('diff_window', c_void_p),
                ('max_batch', c_int64)]


# every exported symbol of include/ssq_hip.h: (restype, argtypes)
_PROTOS = {
    'ssq_version': (c_int, []),
    'ssq_last_error': (c_char_p, []),
    'ssq_device_count': (c_int, [POINTER(c_int)]),
    'ssq_set_device': (c_int, [c_int]),
    'ssq_device_info': (c_int, [c_int, c_char_p, c_int, POINTER(c_int),
                                POINTER(c_int64)]),
    'ssq_malloc': (c_int, [POINTER(c_void_p), c_int64]),
    'ssq_free': (c_int, [c_void_p]),
    'ssq_memcpy_h2d': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'ssq_memcpy_d2h': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'ssq_memset': (c_int, [c_void_p, c_int, c_int64, c_void_p]),
    'ssq_stream_synchronize': (c_int, [c_void_p]),
    'ssq_phase_cwt': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64,
                              c_int64, c_int64, c_double, c_void_p]),
    'ssq_phase_stft': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_int64, c_int64, c_int64, c_double, c_void_p]),
    'ssq_ssqueeze': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_int, c_int64, c_int64, c_int64, c_double,
                             c_int, POINTER(c_double), c_int, c_void_p, c_void_p]),
    'ssq_indexed_sum': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int, c_int64, c_int64, c_int64, c_int,
                                POINTER(c_double), c_int, c_void_p]),
    'ssq_replace_under_abs': (c_int, [c_int, c_void_p, c_void_p, c_int64,
                                      c_double, c_double, c_void_p]),
    'ssq_buffer': (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                           c_int64, c_int, c_void_p]),
    'ssq_pad_signal': (c_int, [c_int, c_void_p, c_void_p, c_int64, c_int64,
                               c_int64, c_int64, c_int, c_void_p]),
    'ssq_colsum': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                           c_int64, c_void_p]),
    'ssq_band_colsum': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64,
                                c_void_p, c_int64, c_int64, c_void_p]),
    'ssq_icwt2': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'ssq_trigdiff': (c_int, [c_int, c_void_p, c_void_p, c_double, c_void_p, c_int64, c_int64,
                             c_int64, c_int64, c_void_p]),
    'ssq_istft': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                          c_int64, c_int64, c_int64, c_int, c_void_p]),
    'ssq_ridge_energy': (c_int, [c_int, c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    'ssq_ridge_neglog': (c_int, [c_int, c_void_p, c_void_p, c_double, c_int64, c_int64, c_void_p]),
    'ssq_ridge_track': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_double, c_double,
                                c_int64, c_int64, c_void_p, c_void_p]),
    'ssq_ridge_clear': (c_int, [c_int, c_void_p, c_void_p, c_double, c_void_p, c_int64, c_int64,
                                c_void_p]),
    'ssq_ridge_neglog_batch': (c_int, [c_int, c_void_p, c_void_p, c_double, c_int64, c_int64, c_int64,
                                       c_void_p]),
    'ssq_ridge_track_batch': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_double, c_double,
                                      c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    'ssq_ridge_clear_batch': (c_int, [c_int, c_void_p, c_void_p, c_double, c_void_p, c_int64, c_int64,
                                      c_int64, c_void_p]),
    'ssq_cwt_plan_create': (c_int, [POINTER(c_void_p), POINTER(CwtDesc)]),
    'ssq_cwt_plan_destroy': (None, [c_void_p]),
    'ssq_cwt_plan_set_ssq': (c_int, [c_void_p, c_int, POINTER(c_double), c_void_p,
                                     c_int, c_int, c_double]),
    'ssq_cwt_execute': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_int, c_void_p]),
    'ssq_cwt_plan_set_blocks': (c_int, [c_void_p, POINTER(CwtBlocksDesc)]),
    'ssq_cwt_plan_set_tiles': (c_int, [c_void_p, POINTER(CwtTilesDesc)]),
    'ssq_cwt_plan_timing': (c_int, [c_void_p, c_int, POINTER(c_double), POINTER(c_int64)]),
    'ssq_cwt_plan_group': (c_int, [c_void_p]),
    'ssq_cwt_plan_bytes': (c_int64, [c_void_p]),
    'ssq_cwt_plan_algo': (c_char_p, [c_void_p]),
    'ssq_cwt_plan_tiles_done': (c_int64, [c_void_p, c_void_p]),
    'ssq_cwt_plan_tile_cols': (c_int, [c_void_p]),
    'ssq_cwt_plan_tile_kernel': (c_int, [c_void_p]),
    'ssq_cwt_plan_tile_counters': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    'ssq_cwt_plan_set_bin_dump': (c_int, [c_void_p, c_void_p]),
    'ssq_build_sha': (c_char_p, []),
    'ssq_cwt_tile_rows_per_step': (c_int, []),
    'ssq_stft_plan_create': (c_int, [POINTER(c_void_p), POINTER(StftDesc)]),
    'ssq_stft_plan_destroy': (None, [c_void_p]),
    'ssq_stft_plan_set_ssq': (c_int, [c_void_p, c_void_p, c_int, POINTER(c_double),
                                      c_void_p, c_int, c_int, c_double]),
    'ssq_stft_plan_shape': (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64)]),
    'ssq_stft_plan_algo': (c_char_p, [c_void_p]),
    'ssq_stft_execute': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
}

EXPORTS = tuple(_PROTOS)
_lib = None


ABI_VERSION = 105     # include/ssq_hip.h: ssq_version()


def load(build_if_missing=True):
    """Load (building first if necessary) libssq_hip.so and declare prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        if not build_if_missing:
            raise SsqError("libssq_hip.so not found at %s; run "
                           "`python -m ssqueezepy_amd.build`" % LIB_PATH)
        from .build import build
        build(verbose=False)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise SsqError("cannot load %s: %s. The ssqueezepy_amd compute path has no "
                       "CPU fallback; it needs the ROCm runtime (libamdhip64, "
                       "librocfft)." % (LIB_PATH, e))
    # the version first: an older build lacks entry points, and the loop below would stop at the
    # first missing one with a bare AttributeError
    lib.ssq_version.restype, lib.ssq_version.argtypes = c_int, []
    if lib.ssq_version() < ABI_VERSION:
        raise SsqError("%s is an older build (ABI %d, this package needs %d); rebuild with "
                       "`python -m ssqueezepy_amd.build`"
                       % (LIB_PATH, lib.ssq_version(), ABI_VERSION))
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)       # AttributeError here = ABI/header mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().ssq_last_error()
        raise SsqError("libssq_hip: %s (code %d)"
                       % (msg.decode() if msg else 'unknown error', rc))


def params5(p):
    arr = (c_double * 5)()
    for i in range(5):
        arr[i] = float(p[i]) if i < len(p) else 0.0
    return arr
