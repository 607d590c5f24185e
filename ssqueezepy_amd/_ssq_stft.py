# -*- coding: utf-8 -*-
"""Synchrosqueezed Short-Time Fourier Transform on the MI355X.

`ssq_stft` / `phase_stft` keep the signatures and return tuples of the reference
(ssqueezepy/_ssq_stft.py:13-136, 201-257); `stft(derivative=True)`, the phase
transform and the reassignment run as one device plan execution.
"""
import numpy as np
from types import FunctionType
import torch

from . import algos
from .configs import EPS32, EPS64
from ._stft import _stft_setup
from .scales import infer_scaletype
from .ssqueezing import _check_ssqueezing_args, ssq_grid_params

__all__ = ['ssq_stft', 'phase_stft']


_SFS_CACHE = {}


def _make_Sfs(n_rows, fs, dtype):
    """`Sfs = linspace(0, fs/2, n_rows)` (ssqueezepy/_ssq_stft.py:103); the last few grids are kept (a call at one
    signal per launch is host-bound: `linspace` was a quarter of its Python time). A copy is handed out: the caller
    owns what `ssq_stft` returns."""
    key = (int(n_rows), float(fs), str(dtype))
    Sfs = _SFS_CACHE.get(key)
    if Sfs is None:
        if len(_SFS_CACHE) >= 16:
            _SFS_CACHE.clear()
        Sfs = _SFS_CACHE[key] = np.linspace(0, .5 * fs, n_rows, dtype=dtype)
    return Sfs.copy()


def ssq_stft(x, window=None, n_fft=None, win_len=None, hop_len=1, fs=None, t=None,
             modulated=True, ssq_freqs=None, padtype='reflect', squeezing='sum',
             gamma=None, preserve_transform=None, dtype=None, astensor=True,
             flipud=False, get_w=False, get_dWx=False):
    """Synchrosqueezed STFT (Thakur & Wu 2011). Arguments follow
    ``ssqueezepy.ssq_stft`` (ssqueezepy/_ssq_stft.py:17-76). Returns
    ``(Tx, Sx, ssq_freqs, Sfs[, w][, dSx])`` with `Tx`, `Sx` of shape
    ``(n_fft//2 + 1, n_hops)``; `ssq_freqs`, `Sfs` are NumPy vectors."""
    if x.ndim == 2 and get_w:
        raise NotImplementedError("`get_w=True` unsupported with batched input.")
    _check_ssqueezing_args(squeezing)
    if (isinstance(ssq_freqs, np.ndarray) and
            infer_scaletype(ssq_freqs)[0] != 'linear'):
        raise ValueError("`ssq_freqs` must be linearly distributed "
                         "for `ssq_stft`")
    plan, xd, fs, dtype = _stft_setup(x, window, n_fft, win_len, hop_len, fs, t,
                                      padtype, modulated, dtype)
    Sfs = _make_Sfs(plan.rows, fs, dtype)
    if gamma is None:
        gamma = 10 * (EPS64 if dtype == 'float64' else EPS32)
    if ssq_freqs is None:
        ssq_freqs = Sfs
    ssq_freqs = np.asarray(ssq_freqs)
    # 'alpha' of Thakur & Wu: the frequency step (ssqueezing.py:133-134)
    const = (ssq_freqs[1] - ssq_freqs[0])
    grid, params = ssq_grid_params(ssq_freqs, False)
    plan.set_ssq(Sfs, grid, params, const, flipud, gamma)
    if squeezing == 'sum':
        out = plan.execute(xd, want_dSx=get_dWx, want_Tx=True, want_w=get_w)
        Tx, Sx, w, dSx = out['Tx'], out['Sx'], out.get('w'), out.get('dSx')
    else:                       # see ssq_cwt: the reassignment runs as its own launch
        out = plan.execute(xd, want_dSx=True, want_Tx=False, want_w=get_w)
        Sx, w = out['Sx'], out.get('w')
        if isinstance(squeezing, FunctionType):
            Sq = squeezing(Sx)
        elif squeezing == 'lebesgue':
            Sq = algos.ones_like(Sx) / len(Sx)
        else:
            Sq = algos.cabs(Sx)
        if get_w:
            Tx = algos.indexed_sum_onfly(Sq, w, ssq_freqs, const, False, flipud)
        else:
            Tx = algos.ssqueeze_fast(Sq, out['dSx'], ssq_freqs, const, False, flipud,
                                     gamma, Sfs=Sfs)
        dSx = out['dSx'] if get_dWx else None
    if flipud:
        ssq_freqs = ssq_freqs[::-1]
    if not astensor:
        Tx, Sx, w, dSx = [g.cpu().numpy() if g is not None else None
                          for g in (Tx, Sx, w, dSx)]
    if get_w and get_dWx:
        return Tx, Sx, ssq_freqs, Sfs, w, dSx
    elif get_w:
        return Tx, Sx, ssq_freqs, Sfs, w
    elif get_dWx:
        return Tx, Sx, ssq_freqs, Sfs, dSx
    return Tx, Sx, ssq_freqs, Sfs


def phase_stft(Sx, dSx, Sfs, gamma=None, parallel=None):
    """Phase transform of the STFT, ``w[u, k] = |Sfs[k] - Im(dSx/Sx)/(2pi)|``,
    ``inf`` where ``|Sx| < gamma`` (default 10*eps). Reference: ``phase_stft``,
    ssqueezepy/_ssq_stft.py:201-246."""
    Sx = algos.to_device(Sx)
    if gamma is None:
        gamma = 10 * (EPS64 if Sx.dtype == torch.complex128 else EPS32)
    return algos.phase_stft_gpu(Sx, dSx, Sfs, gamma)
