# -*- coding: utf-8 -*-
"""ssqueezepy/experimental.py on the MI355X: scale <-> frequency conversions
(`freq_to_scale` :15-85, `scale_to_freq` :88-143; host-side design code, value-exact with the
reference) and `phase_transform` / `phase_ssqueeze` (:146-253) for arbitrary CWT/STFT-like
arrays, composed of this package's device functions (`trigdiff`, `phase_cwt`, `phase_stft`,
`ssqueeze`)."""
import warnings
import numpy as np

from .wavelets import Wavelet, center_frequency
from .scales import cwt_scalebounds
from .padding import p2up

__all__ = ['freq_to_scale', 'scale_to_freq', 'phase_ssqueeze', 'phase_transform']


def freq_to_scale(freqs, wavelet, N, fs=1, n_search_scales=None, kind='peak', base=2):
    """Scales whose centre frequencies (`center_frequency(kind)`) span the ascending
    `freqs` (within [0, fs/2]), log-spaced in `base`. A search, not a closed form:
    `n_search_scales` candidates (default 10 per frequency) between the wavelet's 'maximal'
    scale bounds are mapped to frequencies and the two that land closest to the ends of
    `freqs` delimit the answer. Reference: experimental.py:15-85 (same values)."""
    f = np.asarray(freqs) / fs                      # cycles / sample
    for bad, what in (((f < 0).any(), "frequencies must be positive"),
                      (f.max() > 0.5, "max frequency must be 0.5"),
                      (f.max() != f[-1], "max frequency must be last sample"),
                      (f.min() != f[0], "min frequency must be first sample")):
        if bad:
            raise AssertionError(what)
    n_out = len(f)
    n_grid = 10 * n_out if n_search_scales is None else n_search_scales
    ln_b = np.log(base)
    bounds = cwt_scalebounds(wavelet, N, preset='maximal', use_padded_N=False)
    grid = np.logspace(np.log(bounds[0]) / ln_b, np.log(bounds[1]) / ln_b, n_grid, base=base)
    radians = [min(max(center_frequency(wavelet, s, N, kind=kind), 0), np.pi) for s in grid]
    grid_f = np.array(radians) / (2*np.pi)
    s_first = grid[np.argmin(np.abs(grid_f - f.min()))]      # lowest frequency <-> largest scale
    s_last = grid[np.argmin(np.abs(grid_f - f.max()))]
    return np.logspace(np.log(s_first) / ln_b, np.log(s_last) / ln_b, n_out, base=base)


def scale_to_freq(scales, wavelet, N, fs=1, padtype='reflect'):
    """Frequencies in [0, fs/2] at which the wavelets at `scales` peak on the grid the
    transform uses (the padded length unless `padtype is None`). Reference:
    experimental.py:88-143 (same values)."""
    scales = np.array([scales]) if isinstance(scales, float) else scales
    wavelet = Wavelet._init_if_not_isinstance(wavelet)
    n_grid = N if padtype is None else p2up(N)[0]
    peak = np.argmax(wavelet(scale=scales, N=n_grid), axis=-1)
    off_grid = (peak > n_grid // 2) | (peak == 0)     # negative-frequency or dc "peak"
    if off_grid.any():
        warnings.warn("found potentially ill-behaved wavelets (peak indices at "
                      "negative freqs or at dc); will round idxs to 1 or N/2")
        low_freq_half = np.arange(len(peak)) > len(peak) // 2
        peak = np.where(off_grid, np.where(low_freq_half, 1, n_grid // 2), peak)
    f = peak / n_grid
    if f.min() < 0 or f.max() > 0.5:
        raise AssertionError((f.min(), f.max()))
    return f * fs


def phase_ssqueeze(Wx, dWx=None, ssq_freqs=None, scales=None, Sfs=None, fs=1., t=None,
                   squeezing='sum', maprange=None, wavelet=None, gamma=None,
                   was_padded=True, flipud=False, rpadded=False, padtype=None, N=None,
                   n1=None, difftype=None, difforder=None, get_w=False, get_dWx=False,
                   transform='cwt'):
    """Reassign an arbitrary CWT- / STFT-like `Wx`: `phase_transform` for the instantaneous
    frequencies (or the derivative they are formed from), then `ssqueeze`. Same argument list
    and return tuple ``(Tx, Wx, ssq_freqs, scales, Sfs, w, dWx)`` as the reference's
    (experimental.py:146-187); `maprange` defaults to 'peak' for a CWT and 'maximal' for an
    STFT, and `dWx` is handed back only on request once `w` exists."""
    from .ssqueezing import ssqueeze
    pt = phase_transform(Wx, dWx, difftype, difforder=difforder, gamma=gamma, rpadded=rpadded,
                         padtype=padtype, N=N, n1=n1, get_w=get_w, fs=fs, transform=transform)
    w, Wx, dWx, Sfs, gamma = pt
    keep_dWx = get_dWx or w is None
    Tx, ssq_freqs = ssqueeze(
        Wx, w, ssq_freqs, scales, Sfs, fs=fs, t=t, squeezing=squeezing,
        maprange=maprange or {'cwt': 'peak'}.get(transform, 'maximal'),
        wavelet=wavelet, gamma=gamma, was_padded=was_padded, flipud=flipud,
        dWx=dWx if keep_dWx else None, transform=transform)
    return Tx, Wx, ssq_freqs, scales, Sfs, w, (dWx if keep_dWx else None)


def phase_transform(Wx, dWx=None, difftype='trig', difforder=4, gamma=None, fs=1.,
                    Sfs=None, rpadded=False, padtype='reflect', N=None, n1=None,
                    get_w=False, transform='cwt'):
    """Unified CWT / STFT phase transform (experimental.py:190-253): computes `dWx` by
    `trigdiff` when it is not given (CWT), and `w` when `get_w`. Only `difftype='trig'`
    (`None` counts as 'trig' when `dWx` has to be formed) runs on the device, as in the
    reference's GPU mode. Returns ``w, Wx, dWx, Sfs, gamma``."""
    import torch
    from . import algos
    from .common import trigdiff
    from .configs import EPS32, EPS64
    from ._ssq_cwt import phase_cwt
    from ._ssq_stft import phase_stft, _make_Sfs

    if transform == 'stft' and dWx is None:
        raise NotImplementedError("`phase_transform` without `dWx` for STFT is not "
                                  "currently supported.")
    if rpadded and N is None:
        raise ValueError("`rpadded=True` requires `N`")
    if Wx.ndim > 2 and get_w:
        raise NotImplementedError("`get_w=True` unsupported with batched input.")
    Wx = algos.to_device(Wx)
    if gamma is None:
        gamma = 10 * (EPS64 if Wx.dtype == torch.complex128 else EPS32)

    if transform == 'cwt':
        if N is None and not rpadded:
            N = Wx.shape[-1]
        if n1 is None:
            _, n1, _ = p2up(N)
        if dWx is None:
            dWx = trigdiff(Wx, fs, padtype, rpadded, N=N, n1=n1, transform='cwt')
        w = None
        if get_w:
            if difftype not in (None, 'trig'):
                raise ValueError("`difftype != 'trig'` unsupported with tensor inputs.")
            w = phase_cwt(Wx, dWx, 'trig', gamma)
        Sfs = None
    elif transform == 'stft':
        if Sfs is None:
            rdt = np.float64 if Wx.dtype == torch.complex128 else np.float32
            Sfs = _make_Sfs(Wx.shape[-2], fs, rdt)
        w = phase_stft(Wx, dWx, Sfs, gamma) if get_w else None
    else:
        raise ValueError("`transform` must be one of: cwt, stft (got %s)" % transform)
    return w, Wx, dWx, Sfs, gamma
