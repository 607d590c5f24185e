# -*- coding: utf-8 -*-
"""Package defaults (host side).

Mirrors the *values* of the reference's ``ssqueezepy/configs.ini`` (lines 1-40:
gmw gamma=3 beta=60 bandpass order 0 float32; stft float32; downsample=4) and the
``gdefaults`` "fill what is None" contract (``ssqueezepy/configs.py:27-82``), but
as a plain in-memory table: this engine has no ini file and no CPU mode (it is always
the HIP path). `USE_GPU()` / `IS_PARALLEL()` exist so that code written against the
reference's switches (``ssqueezepy/configs.py:142-155``) imports and branches the same way.
"""
import copy
import os


def USE_GPU():
    """The reference reads ``SSQ_GPU`` here to choose between its NumPy and its CuPy / torch
    paths (ssqueezepy/configs.py:142-147). This package has one path, the HIP one: always
    True, whatever the variable says (the compute layer refuses to run without a GPU, see
    `algos._require_gpu`)."""
    return True


def IS_PARALLEL():
    """The reference's ``SSQ_PARALLEL`` picks its multi-threaded numba kernels
    (configs.py:150-155); nothing here runs on host threads but the design step."""
    return os.environ.get('SSQ_PARALLEL', '1') != '0'


def host_threads(limit=32):
    """Threads the host-side design step (filter bank evaluation, margin measurement) may use:
    ``SSQ_HOST_THREADS`` if set (bench.py gives every rank of a multi-GPU job its share of the
    cores), else the core count, capped at `limit`."""
    n = os.environ.get('SSQ_HOST_THREADS')
    n = int(n) if n else (os.cpu_count() or 1)
    return max(1, min(limit, n))

EPS32 = 1.1920928955078125e-07   # np.finfo(np.float32).eps
EPS64 = 2.220446049250313e-16    # np.finfo(np.float64).eps

_DEFAULTS = {
    'gmw':    dict(gamma=3., beta=60., norm='bandpass', order=0.,
                   centered_scale=False, dtype='float32'),
    'morlet': dict(mu=13.4, dtype='float32'),
    'bump':   dict(mu=5., s=1., om=0., dtype='float32'),
    'cmhat':  dict(mu=1., s=1., dtype='float32'),
    'hhhat':  dict(mu=5., dtype='float32'),
    'stft':   dict(dtype='float32'),
    'make_scales': dict(downsample=4.),
}


def defaults(name):
    """Copy of the default-argument table for `name`."""
    return copy.deepcopy(_DEFAULTS[name])


def fill_defaults(name, **kw):
    """Return `kw` with every `None` replaced by the package default and every
    missing key added, keys ordered as in the defaults table followed by extras
    (same contract as the reference's ``gdefaults(get_all=True,
    default_order=True)``)."""
    table = _DEFAULTS[name]
    out = {}
    for key, dflt in table.items():
        val = kw.get(key, None)
        out[key] = dflt if val is None else val
    for key, val in kw.items():
        if key not in out:
            out[key] = val
    return out
