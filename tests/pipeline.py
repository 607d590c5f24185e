# -*- coding: utf-8 -*-
"""Oracle pipelines: host design step (product code, pinned value-exact by
test_design_vs_golden.py) -> oracle transform (NumPy/scipy.fft) -> oracle loop
nests (C). Mirrors ssq_cwt (ssqueezepy/_ssq_cwt.py:190-310) and ssq_stft
(ssqueezepy/_ssq_stft.py:77-136) stage by stage. Test infrastructure."""
import numpy as np

from ssqueezepy_amd.configs import EPS32, EPS64
from ssqueezepy_amd.padding import pad_geometry
from ssqueezepy_amd.scales import process_scales, _process_fs_and_t
from ssqueezepy_amd.ssqueezing import (_compute_associated_frequencies,
                                       ssq_grid_params, ssq_const)
from ssqueezepy_amd.wavelets import Wavelet

GRIDNAME = {0: 'log', 1: 'log-piecewise', 2: 'linear'}


def oracle_ssq_cwt(orc, x, dtype, wavelet='gmw', scales='log', nv=None, fs=None,
                   padtype='reflect', flipud=True, gamma=None, get_w=False,
                   typing=0, ssq=True, l1_norm=True, maprange='peak'):
    if isinstance(wavelet, str):
        opts = {'dtype': dtype}
        if wavelet == 'gmw':
            opts['norm'] = 'bandpass' if l1_norm else 'energy'
        wavelet = Wavelet((wavelet, opts))
    dtype = wavelet.dtype
    N = x.shape[-1]
    dt, fs, _ = _process_fs_and_t(fs, None, N)
    if nv is None and isinstance(scales, str) and scales != 'linear':
        nv = 32
    scales64, st, *_ = process_scales(scales, N, wavelet, nv=nv, get_params=True)
    sc = np.asarray(scales64, dtype=dtype)
    M, n1, n2 = pad_geometry(N) if padtype is not None else (N, 0, 0)
    Psih = wavelet(scale=sc, N=M, nohalf=False)
    xi = wavelet.xifn(1., M).reshape(-1)
    Wx, dWx = orc.cwt(x, Psih, xi, dt, n1, N, derivative=True, padtype=padtype)
    if not l1_norm:
        Wx = Wx * np.sqrt(sc).astype(Wx.dtype)
        dWx = dWx * np.sqrt(sc).astype(Wx.dtype)
    out = dict(Wx=Wx, dWx=dWx, scales=sc.squeeze())
    if not ssq:
        return out
    if gamma is None:
        gamma = 10 * (EPS64 if dtype == 'float64' else EPS32)
    sc_ssq, st2, _, nv2 = process_scales(sc.squeeze(), N, get_params=True)
    ssq_freqs = _compute_associated_frequencies(sc_ssq, N, wavelet, st2, maprange,
                                                padtype is not None, dt, 'cwt')
    const = ssq_const('cwt', st2, nv2, sc_ssq, ssq_freqs)
    grid, p = ssq_grid_params(ssq_freqs, st2.startswith('log'))
    if get_w:
        w = orc.phase_cwt(Wx, dWx, gamma, typing=typing)
        Tx = orc.indexed_sum(Wx, w, GRIDNAME[grid], p, const, flipud, typing=typing)
        out['w'] = w
    else:
        Tx = orc.ssqueeze(Wx, dWx, GRIDNAME[grid], p, const, gamma, flipud,
                          typing=typing)
    out.update(Tx=Tx, ssq_freqs=ssq_freqs[::-1], const=const, grid=grid, params=p,
               gamma=gamma)
    return out


def oracle_ssq_stft(orc, x, dtype, window=None, n_fft=None, win_len=None, hop_len=1,
                    fs=None, modulated=True, padtype='reflect', flipud=False,
                    gamma=None, get_w=False, typing=0, ssq=True):
    from ssqueezepy_amd._stft import get_window
    N = x.shape[-1]
    _, fs, _ = _process_fs_and_t(fs, None, N)
    n_fft = n_fft or min(N // hop_len, 512)
    if win_len is None:
        win_len = len(window) if isinstance(window, np.ndarray) else n_fft
    win, dwin = get_window(window, win_len, n_fft, derivative=True, dtype=dtype)
    Sx, dSx = orc.stft(x, win, dwin, n_fft, hop_len, fs=fs, modulated=modulated,
                       derivative=True, padtype=padtype)
    out = dict(Sx=Sx, dSx=dSx)
    Sfs = np.linspace(0, .5 * fs, len(Sx), dtype=dtype)
    out['Sfs'] = Sfs
    if not ssq:
        return out
    if gamma is None:
        gamma = 10 * (EPS64 if dtype == 'float64' else EPS32)
    const = Sfs[1] - Sfs[0]
    grid, p = ssq_grid_params(Sfs, False)
    if get_w:
        w = orc.phase_stft(Sx, dSx, Sfs, gamma, typing=typing)
        Tx = orc.indexed_sum(Sx, w, 'linear', p, const, flipud, typing=typing)
        out['w'] = w
    else:
        Tx = orc.ssqueeze(Sx, dSx, 'linear', p, const, gamma, flipud, Sfs=Sfs,
                          typing=typing)
    out.update(Tx=Tx, ssq_freqs=Sfs[::-1] if flipud else Sfs, gamma=gamma)
    return out
