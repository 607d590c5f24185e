# -*- coding: utf-8 -*-
"""Synchrosqueezing: frequency-grid design (host) + dispatch to the HIP kernels.

`ssqueeze` mirrors the reference's ``ssqueezing.ssqueeze``
(ssqueezepy/ssqueezing.py:13-224): it designs the frequency axis `ssq_freqs`
(``_compute_associated_frequencies`` :247-291, ``_ssq_freqrange`` :228-244),
the per-row weights `const` (:122-134) and the closed-form bin map parameters
(``_get_params_find_closest_log``, ssqueezepy/algos.py:356-374), then launches the
fused phase-transform + reassignment kernel (``ssq_ssqueeze`` in the C ABI; the
reference's ``ssqueeze_fast``/``indexed_sum_onfly``, algos.py:126-169).
The grid design is O(na) float64 NumPy and must be value-exact; the O(na*N) work
is on the device.
"""
import logging
from types import FunctionType
import numpy as np

from .configs import EPS64
from .padding import p2up
from .scales import (process_scales, infer_scaletype, logscale_transition_idx,
                     _process_fs_and_t, _to_numpy)
from .wavelets import center_frequency

pi = np.pi
WARN = lambda msg: logging.warning("WARNING: %s" % msg)
NOTE = lambda msg: logging.warning("NOTE: %s" % msg)

GRID_LOG, GRID_LOG_PIECEWISE, GRID_LIN = 0, 1, 2

__all__ = ['ssqueeze', 'ssq_grid_params', 'ssq_const']


# ----------------------------------------------------------- frequency grid
def _get_center_frequency(wavelet, N, maprange, dt, scale, was_padded):
    if was_padded:
        N = p2up(N)[0]
    kw = dict(wavelet=wavelet, N=N, scale=scale, kind=maprange)
    if maprange == 'energy':
        kw['force_int'] = True
    return center_frequency(**kw) / (2 * pi) / dt


def _ssq_freqrange(maprange, dt, N, wavelet, scales, was_padded):
    if isinstance(maprange, tuple):
        return maprange
    if maprange == 'maximal':
        return 1 / (dt * N), 1 / (2 * dt)
    kw = dict(wavelet=wavelet, N=N, maprange=maprange, dt=dt,
              was_padded=was_padded)
    return (_get_center_frequency(**kw, scale=scales[-1]),
            _get_center_frequency(**kw, scale=scales[0]))


def _exp_fm(t, fmin, fmax):
    tmin, tmax = t.min(), t.max()
    a = (fmin**tmax / fmax**tmin) ** (1 / (tmax - tmin))
    b = fmax**(1 / tmax) * (1 / a)**(1 / tmax)
    return a * b**t


def _compute_associated_frequencies(scales, N, wavelet, ssq_scaletype, maprange,
                                    was_padded=True, dt=1, transform='cwt'):
    """Frequencies the rows of `Tx` stand for, low to high: exponential between
    (fm, fM) for 'log', two exponential pieces joined at the scale-rate change
    for 'log-piecewise', linear otherwise."""
    fm, fM = _ssq_freqrange(maprange, dt, N, wavelet, scales, was_padded)
    if not (np.isfinite(fm) and np.isfinite(fM) and fm > 0 and fM > 0):
        # e.g. a compactly supported wavelet sampled at a scale where none of its
        # support lands on the grid: no peak to map (the reference fails here too, with
        # "cannot convert float NaN to integer")
        raise ValueError("could not determine the frequency range of the transform "
                         "(wavelet has no peak on the grid at the extreme scales); "
                         "pass `ssq_freqs` / `maprange` explicitly or adjust `scales`")
    na = len(scales)
    if ssq_scaletype == 'log':
        return fm * np.power(fM / fm, np.arange(na) / (na - 1))
    if ssq_scaletype == 'log-piecewise':
        idx = logscale_transition_idx(scales)
        if idx is None:
            return fm * np.power(fM / fm, np.arange(na) / (na - 1))
        f1 = _get_center_frequency(wavelet, N, maprange, dt, scales[idx],
                                   was_padded)
        t1 = np.arange(0, na - idx - 1) / (na - 1)
        t2 = np.arange(na - idx - 1, na) / (na - 1)
        t1 = np.hstack([t1, t2[0]])
        ssq_freqs = np.hstack([_exp_fm(t1, fm, f1)[:-1], _exp_fm(t2, f1, fM)])
        ssq_idx = logscale_transition_idx(ssq_freqs)
        if ssq_idx is None:
            raise Exception("couldn't find logscale transition index of "
                            "generated `ssq_freqs`; something went wrong")
        assert (na - ssq_idx) == idx, "{} != {}".format(na - ssq_idx, idx)
        return ssq_freqs
    if transform == 'cwt':
        return np.linspace(fm, fM, na)
    return np.linspace(0, .5, na) / dt


# -------------------------------------------------------- bin-map parameters
def _floor_at_eps(name, x, silent=False):
    if x < EPS64:
        if not silent:
            WARN("computed `%s` (%.2e) is below EPS64; will set to " % (name, x)
                 + "EPS64. Advised to check `ssq_freqs`.")
        x = EPS64
    return x


def ssq_grid_params(ssq_freqs, logscale):
    """(grid_kind, params[5]) of the closed-form "which bin is `w` closest to"
    map: log -> (log2 v0, dlog2 v); log-piecewise -> (log2 v0, log2 v[idx-1],
    dlog2 of each piece, idx-1); linear -> (v0, dv)."""
    v = np.asarray(_to_numpy(ssq_freqs))
    p = np.zeros(5, dtype=np.float64)
    if not logscale:
        p[0] = float(v[0])
        p[1] = _floor_at_eps('dv', float(v[1] - v[0]))
        return GRID_LIN, p
    idx = logscale_transition_idx(v)
    p[0] = float(np.log2(v[0]))
    if idx is None:
        p[1] = _floor_at_eps('dvl', float(np.log2(v[1]) - np.log2(v[0])))
        return GRID_LOG, p
    p[1] = float(np.log2(v[idx - 1]))
    p[2] = _floor_at_eps('dvl0', float(np.log2(v[1]) - np.log2(v[0])),
                         silent=True)
    p[3] = _floor_at_eps('dvl1', float(np.log2(v[idx]) - np.log2(v[idx - 1])))
    p[4] = float(np.asarray(idx - 1, dtype=np.int32))
    return GRID_LOG_PIECEWISE, p


def ssq_const(transform, cwt_scaletype, nv, scales, ssq_freqs):
    """Reassignment weights: ln2/nv on exponential scales (a float64 array for
    'log-piecewise', where `nv` is per-scale), ds/s on linear scales, the
    frequency step for the STFT."""
    if transform == 'cwt':
        if cwt_scaletype.startswith('log'):
            return np.log(2) / nv
        return ((scales[1] - scales[0]) / scales).squeeze()
    return ssq_freqs[1] - ssq_freqs[0]


# ------------------------------------------------------------------- checks
def _check_ssqueezing_args(squeezing, maprange=None, wavelet=None, difftype=None,
                           difforder=None, get_w=None, transform='cwt'):
    if transform not in ('cwt', 'stft'):
        raise ValueError("`transform` must be one of: cwt, stft "
                         "(got %s)" % squeezing)
    if not isinstance(squeezing, (str, FunctionType)):
        raise TypeError("`squeezing` must be string or function "
                        "(got %s)" % type(squeezing))
    elif isinstance(squeezing, str) and squeezing not in ('sum', 'lebesgue',
                                                           'abs'):
        raise ValueError("`squeezing` must be one of: sum, lebesgue, abs "
                         "(got %s)" % squeezing)
    if maprange is not None:
        if isinstance(maprange, (tuple, list)):
            if not all(isinstance(m, (float, int)) for m in maprange):
                raise ValueError("all elements of `maprange` must be "
                                 "float or int")
        elif isinstance(maprange, str):
            if maprange not in ('maximal', 'peak', 'energy'):
                raise ValueError("`maprange` must be one of: maximal, peak, "
                                 "energy (got %s)" % maprange)
        else:
            raise TypeError("`maprange` must be str, tuple, or list "
                            "(got %s)" % type(maprange))
        if isinstance(maprange, str) and maprange != 'maximal':
            if transform != 'cwt':
                NOTE("string `maprange` currently only functional with "
                     "`transform='cwt'`")
            elif wavelet is None:
                raise ValueError(f"maprange='{maprange}' requires `wavelet`")
    if difftype is not None:
        if difftype not in ('trig', 'phase', 'numeric'):
            raise ValueError("`difftype` must be one of: direct, phase, numeric"
                             " (got %s)" % difftype)
        elif difftype != 'trig':
            raise ValueError("GPU computation only supports "
                             "`difftype = 'trig'`")
    if difforder is not None:
        if difftype != 'numeric':
            WARN("`difforder` is ignored if `difftype != 'numeric'")
        elif difforder not in (1, 2, 4):
            raise ValueError("`difforder` must be one of: 1, 2, 4 "
                             "(got %s)" % difforder)
    elif difftype == 'numeric':
        difforder = 4
    return difforder


# ------------------------------------------------------------------ ssqueeze
def ssqueeze(Wx, w=None, ssq_freqs=None, scales=None, Sfs=None, fs=None, t=None,
             squeezing='sum', maprange='maximal', wavelet=None, gamma=None,
             was_padded=True, flipud=False, dWx=None, transform='cwt'):
    """Synchrosqueeze a CWT or STFT that already lives on the device.

    Same arguments and return values as the reference's ``ssqueeze``
    (ssqueezepy/ssqueezing.py:13-224): `Wx` (and `dWx` or `w`) are
    ``(na, N)`` or batched ``(B, na, N)`` arrays (NumPy arrays are uploaded,
    torch tensors on the GPU are used in place); returns ``(Tx, ssq_freqs)``
    with `Tx` a device tensor of `Wx`'s shape and dtype.
    """
    from . import algos          # device wrappers (needs the HIP library)

    if w is None and (dWx is None or gamma is None):
        raise ValueError("if `w` is None, `dWx` and `gamma` must not be.")
    _check_ssqueezing_args(squeezing, maprange, transform=transform,
                           wavelet=wavelet)
    if scales is None and transform == 'cwt':
        raise ValueError("`scales` can't be None if `transform == 'cwt'`")
    Wx = algos.to_device(Wx)
    if w is not None:
        w = algos.to_device(w)
        if float(w.min()) < 0:
            raise ValueError("found negatives in `w`")
    if dWx is not None:
        dWx = algos.to_device(dWx)
    N = Wx.shape[-1]
    dt, *_ = _process_fs_and_t(fs, t, N)

    if transform == 'cwt':
        scales, cwt_scaletype, _, nv = process_scales(scales, N, get_params=True)
    else:
        cwt_scaletype, nv = None, None

    if not (isinstance(ssq_freqs, np.ndarray) or hasattr(ssq_freqs, 'detach')):
        ssq_scaletype = ssq_freqs if isinstance(ssq_freqs, str) else cwt_scaletype
        if ((maprange == 'maximal' or isinstance(maprange, tuple)) and
                ssq_scaletype == 'log-piecewise'):
            raise ValueError("can't have `ssq_scaletype = log-piecewise` or "
                             "tuple with `maprange = 'maximal'` "
                             "(got %s)" % str(maprange))
        ssq_freqs = _compute_associated_frequencies(
            scales, N, wavelet, ssq_scaletype, maprange, was_padded, dt,
            transform)
    elif transform == 'stft':
        ssq_freqs = _to_numpy(ssq_freqs)
        ssq_scaletype = 'linear'
    else:
        ssq_freqs = _to_numpy(ssq_freqs)
        ssq_scaletype, _ = infer_scaletype(ssq_freqs)

    if isinstance(squeezing, FunctionType):
        Wx = squeezing(Wx)
    elif squeezing == 'lebesgue':
        Wx = algos.ones_like(Wx) / len(Wx)
    elif squeezing == 'abs':
        Wx = algos.cabs(Wx)

    const = ssq_const(transform, cwt_scaletype, nv, scales, ssq_freqs)
    logscale = ssq_scaletype.startswith('log')
    if w is None:
        Tx = algos.ssqueeze_fast(Wx, dWx, ssq_freqs, const, logscale, flipud,
                                 gamma, Sfs=Sfs)
    else:
        Tx = algos.indexed_sum_onfly(Wx, w, ssq_freqs, const, logscale, flipud)

    if (transform == 'cwt' and not flipud) or flipud:
        ssq_freqs = ssq_freqs[::-1]
    return Tx, ssq_freqs
