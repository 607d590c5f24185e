# -*- coding: utf-8 -*-
"""Scale <-> frequency conversions of ssqueezepy/experimental.py (`freq_to_scale` :15-85,
`scale_to_freq` :88-143). Host-side design code (NumPy), value-exact with the reference;
the transforms themselves (`phase_ssqueeze`, `phase_transform`) are `ssq_cwt` / `ssq_stft` /
`ssqueeze` / `phase_cwt` / `phase_stft` of this package."""
import warnings
import numpy as np

from .wavelets import Wavelet, center_frequency
from .scales import cwt_scalebounds
from .padding import p2up

__all__ = ['freq_to_scale', 'scale_to_freq']


def freq_to_scale(freqs, wavelet, N, fs=1, n_search_scales=None, kind='peak', base=2):
    """Scales whose centre frequencies (`center_frequency(kind)`) span `freqs` (ascending,
    within [0, fs/2]), log-spaced in `base`. Approximate: searches `n_search_scales`
    (default 10 * len(freqs)) candidate scales. Reference: experimental.py:15-85."""
    def log(x):
        return np.log(x) / np.log(base)

    freqs = np.asarray(freqs) / fs  # unitless, [0., 0.5)
    assert np.all(freqs >= 0),       "frequencies must be positive"
    assert freqs.max() <= 0.5,       "max frequency must be 0.5"
    assert freqs.max() == freqs[-1], "max frequency must be last sample"
    assert freqs.min() == freqs[0],  "min frequency must be first sample"

    M = len(freqs)
    if n_search_scales is None:
        n_search_scales = 10 * M
    smin, smax = cwt_scalebounds(wavelet, N, preset='maximal', use_padded_N=False)
    search_scales = np.logspace(log(smin), log(smax), n_search_scales, base=base)

    w_from_scales = []
    for scale in search_scales:
        w = center_frequency(wavelet, scale, N, kind=kind)
        w_from_scales.append(min(max(w, 0), np.pi))
    f_from_scales = np.array(w_from_scales) / (2*np.pi)

    fmin, fmax = freqs.min(), freqs.max()
    smax = search_scales[np.argmin(np.abs(f_from_scales - fmin))]
    smin = search_scales[np.argmin(np.abs(f_from_scales - fmax))]
    return np.logspace(log(smax), log(smin), M, base=base)


def scale_to_freq(scales, wavelet, N, fs=1, padtype='reflect'):
    """Frequencies [0, fs/2] of the peaks of the wavelets at `scales` on the grid the
    transform uses (padded length unless `padtype is None`). Reference:
    experimental.py:88-143."""
    if isinstance(scales, float):
        scales = np.array([scales])
    wavelet = Wavelet._init_if_not_isinstance(wavelet)
    Npad = p2up(N)[0] if padtype is not None else N
    psis = wavelet(scale=scales, N=Npad)
    idxs = np.argmax(psis, axis=-1)
    if np.any(idxs > Npad//2) or 0 in idxs:
        warnings.warn("found potentially ill-behaved wavelets (peak indices at "
                      "negative freqs or at dc); will round idxs to 1 or N/2")
        n_psis = len(psis)
        for i, ix in enumerate(idxs):
            if ix > Npad//2 or ix == 0:
                idxs[i] = 1 if i > n_psis // 2 else Npad//2
    freqs = idxs / Npad
    assert freqs.min() >= 0,   freqs.min()
    assert freqs.max() <= 0.5, freqs.max()
    freqs *= fs
    return freqs
