# -*- coding: utf-8 -*-
"""CWT scale design (host side, once per configuration).

Restates, value for value, the scale-vector logic the reference keeps in
``ssqueezepy/utils/cwt_utils.py``: ``process_scales`` (196-261),
``infer_scaletype`` (264-298), ``make_scales`` (301-372),
``logscale_transition_idx`` (375-394), ``nv_from_scales`` (397-409),
``cwt_scalebounds`` (66-188), ``find_min_scale`` (412-432), ``find_max_scale``
(435-458), ``find_downsampling_scale`` (461-579), ``find_max_scale_alt``
(630-695) and ``_process_fs_and_t`` (698-718). The scale vector decides the rows
of the transform and (through `nv`) the synchrosqueezing weights, so it must be
value-exact; it is O(na) work and stays in NumPy float64.
"""
import logging
import numpy as np

from .configs import defaults
from .padding import p2up
from .wavelets import (Wavelet, center_frequency, find_maximum,
                       find_first_occurrence)

pi = np.pi
WARN = lambda msg: logging.warning("WARNING: %s" % msg)

__all__ = ['adm_ssq', 'adm_cwt', 'integrate_analytic', 'process_scales', 'infer_scaletype', 'make_scales',
           'logscale_transition_idx', 'nv_from_scales', 'cwt_scalebounds',
           'find_min_scale', 'find_max_scale', 'find_max_scale_alt',
           'find_downsampling_scale']


def _to_numpy(x):
    if hasattr(x, 'detach'):
        return x.detach().cpu().numpy()
    return x


def _process_fs_and_t(fs, t, N):
    """``(dt, fs, t)`` from a sampling rate or a time vector (which wins when both are given and
    must be uniform and as long as the signal). Reference: utils/cwt_utils.py:698-716."""
    if t is None:
        if fs is None:
            return 1.0, 1, None
        if fs <= 0:
            raise ValueError("`fs` must be > 0")
        return 1 / fs, fs, None
    if fs is not None:
        WARN("`t` will override `fs` (both were passed)")
    if len(t) != N:
        raise Exception("`t` must be of same length as `x` (%s != %s)" % (len(t), N))
    second_difference = np.abs(np.diff(t, 2, axis=0))
    if not np.mean(second_difference) < 1e-7:
        raise Exception("Time vector `t` must be uniformly sampled.")
    fs = 1 / (t[1] - t[0])
    return 1 / fs, fs, t


# ---------------------------------------------------------- scale-vector kinds
def logscale_transition_idx(scales):
    """Index splitting a two-rate exponential vector into its two pieces, or
    None when there is no single, clean change of rate."""
    scales = _to_numpy(scales)
    d2 = np.abs(np.diff(np.log(scales), 2, axis=0))
    idx = np.argmax(d2) + 2
    d2_max = d2.max()
    d2[idx - 2] = 0                      # all *other* 2nd differences must vanish
    th = 1e-14 if scales.dtype == np.float64 else 1e-6
    if not np.any(d2_max > 100 * np.abs(d2).mean()):
        return None
    if not np.all(np.abs(d2) < th):
        return None
    return idx


def nv_from_scales(scales):
    """Voices per octave at every scale (array; two values for 'log-piecewise')."""
    scales = _to_numpy(scales)
    rate = 1 / np.diff(np.log2(scales), axis=0)
    nv = np.vstack([rate[:1], rate])
    idx = logscale_transition_idx(scales)
    if idx is not None:
        at = np.argmax(np.abs(np.diff(nv, axis=0))) + 1
        assert at == idx, "%s != %s" % (at, idx)
    return nv


def infer_scaletype(scales):
    """'log' / 'linear' / 'log-piecewise' and `nv`, from the values alone."""
    scales = _to_numpy(scales)
    if not isinstance(scales, np.ndarray):
        raise TypeError("`scales` must be a numpy array (got %s)" % type(scales))
    scales = scales.reshape(-1, 1)
    if scales.dtype not in (np.float32, np.float64):
        raise TypeError("`scales.dtype` must be np.float32 or np.float64 "
                        "(got %s)" % scales.dtype)
    th_log = 4e-15 if scales.dtype == np.float64 else 8e-7
    th_lin = th_log * 1e3

    if np.mean(np.abs(np.diff(np.log(scales), 2, axis=0))) < th_log:
        nv = int(np.round(1 / np.diff(np.log2(scales), axis=0)[0].squeeze()))
        return 'log', nv
    if np.mean(np.abs(np.diff(scales, 2, axis=0))) < th_lin:
        return 'linear', None
    if logscale_transition_idx(scales) is None:
        raise ValueError("could not infer `scaletype` from `scales`; "
                         "`scales` array must be linear or exponential. "
                         "(got diff(scales)=%s..." % np.diff(scales, axis=0)[:4])
    return 'log-piecewise', nv_from_scales(scales)


# ------------------------------------------------------------- scale bounds
def find_min_scale(wavelet, cutoff=1):
    """Scale at which the wavelet, sampled at Nyquist, equals ``|cutoff|`` times
    its peak (right of the peak for cutoff > 0, left otherwise)."""
    wavelet = Wavelet._init_if_not_isinstance(wavelet)
    w_peak, peak = find_maximum(wavelet.fn)
    lo, hi = (w_peak, 10 * w_peak) if cutoff > 0 else (0, w_peak)
    w_cut, _ = find_first_occurrence(wavelet.fn, value=abs(cutoff) * peak,
                                     step_start=lo, step_limit=hi)
    return w_cut / pi


def find_max_scale(wavelet, N, bin_loc=1, bin_amp=1):
    """Scale putting `bin_amp` of the wavelet's peak amplitude on DFT bin
    `bin_loc` of an N-point grid."""
    wavelet = Wavelet._init_if_not_isinstance(wavelet)
    scale_c = (4 / pi) * center_frequency(wavelet, kind='peak-ct', N=N)
    psih = np.asarray(wavelet(scale=scale_c, N=N)[:N // 2 + 1])
    xi = np.asarray(wavelet.xifn(scale_c, N))
    peak_at = np.argmax(psih)
    below = np.where(psih[:peak_at] < psih.max() * bin_amp)[0][-1]
    return scale_c * (xi[below] / xi[bin_loc])


def find_max_scale_alt(wavelet, N, min_cutoff=.1, max_cutoff=.8):
    """'minimal'-preset upper scale: the bin spacing that straddles the peak
    symmetrically using the fewest bins (see the reference's docstring,
    ssqueezepy/utils/cwt_utils.py:630-658, for the construction)."""
    if max_cutoff <= 0 or min_cutoff <= 0:
        raise ValueError("`max_cutoff` and `min_cutoff` must be positive "
                         "(got %s, %s)" % (max_cutoff, min_cutoff))
    elif max_cutoff <= min_cutoff:
        raise ValueError("must have `max_cutoff > min_cutoff` "
                         "(got %s, %s)" % (max_cutoff, min_cutoff))
    wavelet = Wavelet._init_if_not_isinstance(wavelet)
    w_peak, peak = find_maximum(wavelet.fn)
    w_cut, _ = find_first_occurrence(wavelet.fn, value=min_cutoff * peak,
                                     step_start=0, step_limit=w_peak)
    left = np.arange(w_cut, w_peak, step=1 / N)[:-1]
    spacing = (w_peak - left) * 2
    n_div = left / spacing
    try:
        at = np.where(np.diff(n_div % 1) < -.8)[0][0]
    except Exception:
        raise Exception("Failed to find suffciently-integer xi divisions; try "
                        "widening (min_cutoff, max_cutoff)")
    return spacing[at + 1] / (pi / (N / 2))


def cwt_scalebounds(wavelet, N, preset=None, min_cutoff=None, max_cutoff=None,
                    cutoff=None, bin_loc=None, bin_amp=None, use_padded_N=True):
    """(min_scale, max_scale) over which `wavelet` is well-behaved on the
    (padded) length-N grid; `preset` in ('maximal', 'minimal', 'naive', None)."""
    dflt = dict(min_cutoff=.6, max_cutoff=.8, cutoff=-.5)
    if preset is not None:
        if any((min_cutoff, max_cutoff, cutoff)):
            WARN("`preset` will override `min_cutoff, max_cutoff, cutoff`")
        elif preset == 'minimal' and any((bin_amp, bin_loc)):
            WARN("`preset='minimal'` ignores `bin_amp` & `bin_loc`")
        if preset not in ('maximal', 'minimal', 'naive'):
            raise ValueError("`preset` must be one of: maximal, minimal, naive "
                             "(got %s)" % preset)
        if preset in ('naive', 'maximal'):
            min_cutoff, max_cutoff = None, None
            if preset == 'maximal':
                cutoff = -.5
        else:
            min_cutoff, max_cutoff, cutoff = dflt.values()
    else:
        if min_cutoff is None:
            min_cutoff = dflt['min_cutoff']
        elif min_cutoff <= 0:
            raise ValueError("`min_cutoff` must be >0 (got %s)" % min_cutoff)
        if max_cutoff is None:
            max_cutoff = dflt['max_cutoff']
        elif max_cutoff < min_cutoff:
            raise ValueError("must have `max_cutoff > min_cutoff` "
                             "(got %s, %s)" % (max_cutoff, min_cutoff))
    bin_loc = bin_loc or (2 if preset == 'maximal' else None)
    bin_amp = bin_amp or (1 if preset == 'maximal' else None)
    cutoff = cutoff if (cutoff is not None) else dflt['cutoff']

    if preset == 'naive':
        return 1, N
    M = p2up(N)[0] if use_padded_N else N
    min_scale = find_min_scale(wavelet, cutoff=cutoff)
    if preset in ('minimal', None):
        max_scale = find_max_scale_alt(wavelet, M, min_cutoff=min_cutoff,
                                       max_cutoff=max_cutoff)
    else:
        max_scale = find_max_scale(wavelet, M, bin_loc=bin_loc, bin_amp=bin_amp)
    return min_scale, max_scale


def find_downsampling_scale(wavelet, scales, span=5, tol=3, method='sum',
                            nonzero_th=.02, nonzero_tol=4., N=None):
    """First scale index past which `span` consecutive wavelets (on a 2048-point
    grid) crowd the same few bins — where 'log-piecewise' starts skipping
    scales. None when no such index exists."""
    if method not in ('any', 'all', 'sum'):
        raise ValueError("`method` must be one of: any, all, sum (got %s)"
                         % method)
    N = N or 2048
    if isinstance(wavelet, np.ndarray):
        bank = wavelet
    else:
        wavelet = Wavelet._init_if_not_isinstance(wavelet)
        bank = np.asarray(wavelet(scale=scales, N=N))
    if len(bank) != len(scales):
        raise ValueError("len(Psih) != len(scales) "
                         "(%s != %s)" % (len(bank), len(scales)))
    bank = bank[:, :bank.shape[1] // 2]          # analytic: right half is zero
    n_groups = len(bank) - span - 1

    i = None
    for i in range(n_groups):
        grp = bank[i:i + span]
        tops = grp.max(axis=1)[:, None]
        if (grp > nonzero_th * tops).sum() / span > nonzero_tol:
            continue
        peaks = np.where(grp == tops)[1]
        joint = np.argmax(np.prod(grp, 0))
        dist = np.abs(peaks - joint)
        if method == 'any':
            crowded = dist.max() < tol
        elif method == 'all':
            crowded = not np.all(dist > tol)
        else:
            crowded = dist.sum() < tol
        if crowded:
            break
    return i if (i is not None and i < n_groups - 1) else None


# ----------------------------------------------------------------- assembly
def make_scales(N, min_scale=None, max_scale=None, nv=32, scaletype='log',
                wavelet=None, downsample=None):
    """Scale column vector ``(na, 1)`` float64: powers ``2**(p/nv)`` for 'log',
    the same with the redundant tail kept every `downsample`-th for
    'log-piecewise', an even spacing for 'linear'."""
    if scaletype == 'log-piecewise' and wavelet is None:
        raise ValueError("must pass `wavelet` for `scaletype == 'log-piecewise'`")
    if min_scale is None and max_scale is None and wavelet is not None:
        min_scale, max_scale = cwt_scalebounds(wavelet, N, use_padded_N=True)
    else:
        min_scale = min_scale or 1
        max_scale = max_scale or N
    if downsample is None:
        downsample = defaults('make_scales')['downsample']
    downsample = int(downsample)

    na = int(np.ceil(nv * np.log2(max_scale / min_scale)))
    p_lo = int(np.floor(nv * np.log2(min_scale)))
    p_hi = p_lo + na

    if scaletype in ('log', 'log-piecewise'):
        scales = 2 ** (np.arange(p_lo, p_hi) / nv)
        if scaletype == 'log-piecewise':
            cut = find_downsampling_scale(wavelet, scales)
            if cut is not None:
                scales = np.hstack([scales[:cut],
                                    scales[cut + downsample - 1::downsample]])
    elif scaletype == 'linear':
        min_scale, max_scale = 2**(p_lo / nv), 2**(p_hi / nv)
        na = int(np.ceil(max_scale / min_scale))
        scales = np.linspace(min_scale, max_scale, na)
    else:
        raise ValueError("`scaletype` must be 'log' or 'linear'; "
                         "got: %s" % scaletype)
    return scales.reshape(-1, 1)


def process_scales(scales, N, wavelet=None, nv=None, get_params=False,
                   use_padded_N=True):
    """Build the scale vector from a scheme name ('log', 'log-piecewise',
    'linear', optionally 'name:preset'), or validate a given array; with
    `get_params` also return ``(scaletype, na, nv)``."""
    preset = None
    if isinstance(scales, str):
        if ':' in scales:
            scales, preset = scales.split(':')
        elif scales == 'log-piecewise':
            preset = 'maximal'
        if scales not in ('log', 'log-piecewise', 'linear'):
            raise ValueError("`scales` must be one of: log, log-piecewise, "
                             "linear (got %s)" % scales)
        if nv is None:
            nv = 32
        if wavelet is None:
            raise ValueError("must set `wavelet` if `scales` isn't array")
        scaletype = scales
    elif isinstance(scales, np.ndarray) or hasattr(scales, 'detach'):
        scales = _to_numpy(scales)
        if scales.squeeze().ndim != 1:
            raise ValueError("`scales`, if array, must be 1D "
                             "(got shape %s)" % str(scales.shape))
        scaletype, nv_seen = infer_scaletype(scales)
        if scaletype == 'log':
            if nv is not None and nv_seen != nv:
                raise Exception("`nv` used in `scales` differs from "
                                "`nv` passed (%s != %s)" % (nv_seen, nv))
            nv = nv_seen
        elif scaletype == 'log-piecewise':
            nv = nv_seen
        scales = scales.reshape(-1, 1)
    else:
        raise TypeError("`scales` must be a string or Numpy array "
                        "(got %s)" % type(scales))

    if nv is not None and not isinstance(nv, np.ndarray):
        if not (nv > 0 and float(nv).is_integer()):
            raise ValueError(f"'nv' must be a positive integer (got {nv})")
        nv = int(nv)

    if isinstance(scales, np.ndarray):
        return (scales if not get_params else
                (scales, scaletype, len(scales), nv))

    lo, hi = cwt_scalebounds(wavelet, N=N, preset=preset,
                             use_padded_N=use_padded_N)
    scales = make_scales(N, lo, hi, nv=nv, scaletype=scaletype, wavelet=wavelet)
    return (scales if not get_params else
            (scales, scaletype, len(scales), nv))


# ------------------------------------------------------- admissibility constants
def integrate_analytic(int_fn, nowarn=False):
    """Integral over (0, inf) of a function that vanishes for negative arguments, has one
    hump and decays to the right (an analytic wavelet divided by `w`): trapezoid rule on
    a logarithmic grid for (1e-15, 0.1) plus a uniform grid from 0.1 up to the first
    candidate limit (1, 20, 80, 160; 1e4, 1e4, 4e4, 8e4 points) that contains the hump
    and at least 1000*m points of it below 1e-15 -- the quadrature the reference defines
    (utils/cwt_utils.py:583-627), so the constants agree to the last digits."""
    from scipy import integrate
    t0 = np.logspace(-15, -1, 1000)
    near_zero = integrate.trapezoid(int_fn(t0), t0)
    arr = t = None
    cut = 0
    for m, lim in zip((1, 1, 4, 8), (1, 20, 80, 160)):
        n = 10000 * m
        t = np.linspace(lim, .1, n, endpoint=False)[::-1].copy()
        arr = int_fn(t)
        peak = int(np.argmax(arr))
        small = np.nonzero(np.abs(arr[peak:]) < 1e-15)[0]
        cut = peak + (int(small[0]) if len(small) else len(arr) - peak - 1)
        if (len(t) - cut > 1000 * m) and np.sum(np.abs(arr)) > 1e-5:
            break
    else:
        if near_zero < 1e-5:
            raise Exception("Could not find converging or non-negligibly-valued "
                            "bounds of integration for `int_fn`")
        elif not nowarn:
            WARN("Integrated only from 1e-15 to 0.1 in logspace")
    return integrate.trapezoid(arr[:cut], t[:cut]) + near_zero


def _wavelet_fn(wavelet):
    from .wavelets import Wavelet
    return Wavelet._init_if_not_isinstance(wavelet).fn


_ADM_CACHE = {}


def _adm_cached(kind, wavelet, compute):
    # the quadrature costs ~0.1 ms of host time; the constants depend only on the
    # wavelet's family and parameters
    from .wavelets import Wavelet
    w = wavelet if isinstance(wavelet, Wavelet) else None
    key = None
    if isinstance(wavelet, str):
        key = (kind, wavelet)
    elif w is not None and w.family is not None:
        key = (kind, w.family, tuple(sorted((k, str(v)) for k, v in w.config.items())))
    if key is not None and key in _ADM_CACHE:
        return _ADM_CACHE[key]
    val = compute()
    if key is not None:
        _ADM_CACHE[key] = val
    return val


def adm_ssq(wavelet):
    """Synchrosqueezing admissibility constant ``int_0^inf conj(psih(w)) / w dw``
    (reference: utils/cwt_utils.py:28-47)."""
    def compute():
        fn = _wavelet_fn(wavelet)
        c = integrate_analytic(lambda w: np.conj(_to_numpy(fn(w))) / w)
        return c.real if abs(c.imag) < 1e-15 else c
    return _adm_cached('ssq', wavelet, compute)


def adm_cwt(wavelet):
    """CWT admissibility constant ``int_0^inf |psih(w)|^2 / w dw``
    (reference: utils/cwt_utils.py:50-64)."""
    def compute():
        fn = _wavelet_fn(wavelet)
        c = integrate_analytic(lambda w: np.conj(_to_numpy(fn(w))) * _to_numpy(fn(w)) / w)
        return c.real if abs(c.imag) < 1e-15 else c
    return _adm_cached('cwt', wavelet, compute)
