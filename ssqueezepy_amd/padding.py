# -*- coding: utf-8 -*-
"""Signal-extension geometry (reference ``p2up`` / ``padsignal``,
ssqueezepy/utils/common.py:32-158).

The transform itself pads on the device (`pad_signal_*` kernels in
``csrc/ssq_kernels.hip``); this module owns the *geometry* (padded length, left
and right margins) those kernels are launched with, and a host implementation of
the same extension modes for NumPy inputs (used by callers that want the padded
signal itself, and by the tests to pin the device kernels).
"""
import numpy as np

PADTYPES = ('reflect', 'symmetric', 'replicate', 'wrap', 'zero')
PAD_CODE = {None: -1, 'zero': 0, 'reflect': 1, 'symmetric': 2, 'replicate': 3,
            'wrap': 4}

__all__ = ['p2up', 'padsignal', 'pad_geometry']


def p2up(n):
    """Padded length ``2**(1 + round(log2 n))`` and the (left, right) margins
    that centre the original `n` samples (left gets the odd sample)."""
    up = int(2**(1 + np.round(np.log2(n))))
    n2 = int((up - n) // 2)
    n1 = int(up - n - n2)
    return up, n1, n2


def pad_geometry(N, padlength=None):
    """(n_up, n1, n2) for a length-`N` signal; `padlength=None` -> `p2up`."""
    if padlength is None:
        return p2up(N)
    n_up = int(padlength)
    n2 = (n_up - N) // 2
    n1 = n2 if abs(n_up - N) % 2 == 0 else n2 + 1
    return n_up, int(n1), int(n2)


def padsignal(x, padtype='reflect', padlength=None, get_params=False):
    """Extend `x` (1D, or 2D with time last) to `padlength` samples.

    'reflect' mirrors without repeating the edge sample, 'symmetric' repeats it,
    'replicate' holds the edge value, 'wrap' is periodic, 'zero' zero-fills.
    Accepts NumPy arrays (all modes) and torch tensors ('zero', 'reflect'), as
    the reference does.
    """
    if not hasattr(x, 'ndim'):
        raise TypeError("`x` must be a numpy array or torch Tensor "
                        "(got %s)" % type(x))
    elif x.ndim not in (1, 2):
        raise ValueError("`x` must be 1D or 2D (got x.ndim == %s)" % x.ndim)
    is_numpy = isinstance(x, np.ndarray)
    supported = PADTYPES if is_numpy else ('zero', 'reflect')
    if padtype not in supported:
        raise ValueError("`padtype` must be one of: %s (got %s)"
                         % (', '.join(supported), padtype))
    N = x.shape[-1]
    n_up, n1, n2 = pad_geometry(N, padlength)

    if is_numpy:
        width = (n1, n2) if x.ndim == 1 else [(0, 0), (n1, n2)]
        if padtype == 'symmetric':
            rev = x[..., ::-1]
            xp = np.concatenate([rev[..., N - n1:], x, rev[..., :n2]], axis=-1)
        else:
            mode = {'zero': 'constant', 'reflect': 'reflect',
                    'replicate': 'edge', 'wrap': 'wrap'}[padtype]
            xp = np.pad(x, width, mode=mode)
    else:
        import torch
        mode = 'constant' if padtype == 'zero' else 'reflect'
        xp = (torch.nn.functional.pad(x[None], (n1, n2), mode)[0]
              if x.ndim == 1 else
              torch.nn.functional.pad(x, (n1, n2), mode))
    return (xp, n_up, n1, n2) if get_params else xp
