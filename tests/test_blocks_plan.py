# -*- coding: utf-8 -*-
"""Host-side planning of the block ("overlap-save zoom") CWT path (ssqueezepy_amd/_blocks.py):
structural invariants of the tables the kernels trust, and the margin measurement on
adaptive grids against the plain full-length measurement. CPU only."""
import numpy as np
import pytest
import scipy.fft as sfft

from ssqueezepy_amd.wavelets import Wavelet
from ssqueezepy_amd.scales import process_scales
from ssqueezepy_amd._bank import banded_bank, support_hull
from ssqueezepy_amd.padding import pad_geometry
from ssqueezepy_amd import _blocks


def _bank(N, nv, dtype, wavelet='gmw'):
    wav = Wavelet((wavelet, {'dtype': dtype}))
    scales = np.asarray(process_scales('log', N, wav, nv=nv), dtype=dtype).reshape(-1)
    M, n1, _ = pad_geometry(N)
    tol = 1e-3 * np.finfo(dtype).eps
    vals, off, lo = banded_bank(wav, scales, M, tol=tol)
    v64 = None
    if dtype == 'float32':
        tw = Wavelet((wavelet, {'dtype': 'float64'}))
        v64, o64, l64 = banded_bank(tw, scales.astype('float64'), M, tol=tol)
        assert np.array_equal(o64, off) and np.array_equal(l64, lo)
    return vals, off, lo, M, n1, v64


def _extension(N, nv, dtype, wavelet='gmw'):
    """what `_cwt._try_blocks` hands to plan_blocks to continue Nyquist-cut rows"""
    wav = Wavelet((wavelet, {'dtype': dtype}))
    scales = np.asarray(process_scales('log', N, wav, nv=nv), dtype=dtype).reshape(-1)
    _, w_hi = support_hull(wav.fn, np.dtype(dtype), 1e-3 * np.finfo(dtype).eps,
                           w_extent=float(scales.max()) * np.pi * 1.01 + 1)
    return Wavelet((wavelet, {'dtype': 'float64'})).fn, scales.astype('float64'), float(w_hi)


@pytest.mark.parametrize('N,nv,dtype', [(6000, 8, 'float32'), (20000, 16, 'float32'),
                                        (50000, 4, 'float64'), (4097, 8, 'float64')])
@pytest.mark.parametrize('extend', [False, True])
def test_plan_invariants(N, nv, dtype, extend):
    vals, off, lo, M, n1, v64 = _bank(N, nv, dtype)
    bp = _blocks.plan_blocks(vals, off, lo, M, N, n1, dtype, vals64=v64,
                             extension=_extension(N, nv, dtype) if extend else None)
    assert bp is not None
    points = _blocks.POINTS_PER_WG[dtype]
    cls, rows = bp['classes'], bp['rows']
    cut = (lo + np.diff(off)) == M // 2 + 1
    ext = bp['extended']
    if extend and M > _blocks.P_MIN:
        # the Nyquist-cut rows of a GMW bank are continued and run as block rows -- all of them at
        # the usual densities; a coarse bank's first row (nv = 4: its peak lies well past Nyquist)
        # is refused by the gain limit and stays on the exact path
        assert cut.any() and not np.any(ext & ~cut)
        assert ext.sum() >= cut.sum() - (1 if nv <= 4 else 0)
        assert np.array_equal(np.sort(bp['generic_rows']), np.nonzero(cut & ~ext)[0])
        assert np.all(cls[rows[ext, 0], 4] == 1) and np.all(cls[rows[~ext & (rows[:, 0] >= 0), 0], 4] == 0)
        assert np.all(np.diff(cls[:, 4]) >= 0)            # analytic classes come last
        # the continued band: the bank's values (Nyquist bin un-halved), then the continuation
        for i in np.nonzero(ext)[0]:
            n_in = off[i + 1] - off[i]
            got = bp['band_vals'][bp['band_off'][i]:bp['band_off'][i + 1]]
            assert np.array_equal(got[:n_in - 1], vals[off[i]:off[i + 1] - 1])
            assert got[n_in - 1] == 2 * vals[off[i + 1] - 1] and len(got) > n_in
            assert lo[i] + len(got) <= M and abs(got[-1]) <= 1e-6 * np.abs(got).max()
    else:
        assert not ext.any() and np.all(cls[:, 4] == 0)
    # (from here on the bands the kernels apply)
    vals, off = bp['band_vals'], bp['band_off']
    lens = np.diff(off)
    for P, m, V, nb, ana in cls:
        assert P & (P - 1) == 0 and _blocks.P_MIN <= P <= M
        if P == M:
            assert (m, V, nb) == (0, M, 1)
        else:
            assert V == P - 2 * m and m >= P // 32 and nb == -(-N // V)
    n_items = 0
    for i, (c, klo, KP, Lp, G, pboff) in enumerate(rows):
        if c < 0:
            assert i in bp['generic_rows']
            continue
        P, m, V, nb = cls[c, :4]
        S = M // P
        assert Lp * G == points and KP <= Lp and Lp <= _blocks.L_MAX and P // Lp >= G
        # the P-grid band covers every P-grid bin inside the row's M-grid band
        assert klo * S >= lo[i] and (klo + KP - 1) * S <= lo[i] + lens[i] - 1
        assert (klo - 1) * S < lo[i] and (klo + KP) * S > lo[i] + lens[i] - 1
        # the margin covers the measured response (the single-block class needs none)
        assert P == M or bp['margins'][i] <= m
        # band values are the M-grid bank sampled every S-th bin
        sel = np.arange(klo, klo + KP) * S - lo[i] + off[i]
        assert np.array_equal(bp['pbank'][pboff:pboff + KP], vals[sel])
        n_items += nb * (P // Lp) // G
    assert n_items == sum(len(v) for v in bp['items'].values())
    for Lp, it in bp['items'].items():
        r = rows[it[:, 0]]
        assert np.all(r[:, 3] == Lp) and np.all(it[:, 3] == r[:, 0])
        assert np.all(it[:, 2] % r[:, 4] == 0)
    assert len(bp['generic_rows']) + np.count_nonzero(rows[:, 0] >= 0) == len(rows)


def _margins_plain(vals, off, lo, M, tol):
    na, lens, half = len(lo), np.diff(off), M // 2
    out = np.empty(na, np.int64)
    for i in range(na):
        D = np.zeros(M, np.complex128)
        D[lo[i]:lo[i] + lens[i]] = vals[off[i]:off[i + 1]]
        h = np.abs(sfft.ifft(D))
        f = h[:half + 1].copy()
        f[1:half] += h[:half:-1]
        cs = np.cumsum(f[::-1])[::-1]
        ok = np.nonzero(cs <= tol * f.sum())[0]
        out[i] = ok[0] if len(ok) else half
    return out


def test_adaptive_margins_match_the_full_length_measurement():
    vals, off, lo, M, n1, v64 = _bank(20000, 16, 'float32')
    fast = _blocks._margins(v64, off, lo, M, 1e-9)
    plain = _margins_plain(v64, off, lo, M, 1e-9)
    nyq = (lo + np.diff(off)) >= M // 2 + 1          # sent to the exact path unmeasured
    assert np.all(fast[nyq] == M // 2)
    d = (fast - plain)[~nyq]
    # never short by more than the slack the classes have anyway, never long by > 2 %
    assert d.min() >= -2 and np.all(d <= 0.02 * plain[~nyq] + 8)


def test_margins_of_rows_continued_past_nyquist():
    """The rows continued past the Nyquist bin are measured on the adaptive grids like any other wide
    band: against the plain full-length measurement of the same (continued) band, and they come out
    compact (tens of samples) where the cut rows themselves never do."""
    N, nv = 20000, 16
    vals, off, lo, M, n1, v64 = _bank(N, nv, 'float32')
    fn, scales, w_hi = _extension(N, nv, 'float32')
    vx, ox, v64x, ext = _blocks.extend_past_nyquist(fn, scales, w_hi, vals, off, lo, M, v64)
    assert ext.sum() >= 5 and np.array_equal(ext, (lo + np.diff(off)) == M // 2 + 1)
    fast = _blocks._margins(v64x, ox, lo, M, 1e-9, extended=ext)
    plain = _margins_plain(v64x, ox, lo, M, 1e-9)
    d = (fast - plain)[ext]
    assert d.min() >= -2 and np.all(d <= 0.02 * plain[ext] + 8), (fast[ext], plain[ext])
    assert fast[ext].max() <= 128                      # fits the shortest block class (P = 4096, margin P / 32)
    # the same rows as the reference cuts them: response ~ 1 / t, no margin below M / 4 at 1e-9
    cut = _margins_plain(v64, off, lo, M, 1e-9)
    assert cut[ext].min() > M // 8


def test_rows_peaking_past_nyquist_are_not_continued():
    """A row is continued past the Nyquist bin only while the continuation stays within
    NYQ_EXT_GAIN of its in-band weights: the analytic signal's bins above Nyquist are rounding
    noise, not zeros, and a continuation hundreds of times the in-band weights (the smallest
    'bump' scales: the wavelet's peak itself lies past Nyquist) would amplify that noise into
    the row -- a float32 simulation of such a row loses three digits (1e-3 of the row's maximum
    against 2e-5 on the full-length path). Those rows stay on the exact path; the default GMW
    bank is unaffected (`test_plan_invariants`)."""
    N = 5000
    for wavelet, nv, some_refused in (('bump', 8, True), ('bump', 32, False), ('gmw', 16, False), ('morlet', 4, True)):
        vals, off, lo, M, n1, v64 = _bank(N, nv, 'float32', wavelet)
        fn, scales, w_hi = _extension(N, nv, 'float32', wavelet)
        vx, ox, v64x, ext = _blocks.extend_past_nyquist(fn, scales, w_hi, vals, off, lo, M, v64)
        cut = (lo + np.diff(off)) == M // 2 + 1
        assert cut.any() and not np.any(ext & ~cut)
        assert np.any(cut & ~ext) == some_refused, wavelet
        for i in np.nonzero(ext)[0]:
            n_in = off[i + 1] - off[i]
            row = v64x[ox[i]:ox[i + 1]]
            assert np.abs(row[n_in:]).max() <= _blocks.NYQ_EXT_GAIN * np.abs(row[:n_in]).max(), (wavelet, i)
        # a refused row keeps its band as the reference cuts it
        for i in np.nonzero(cut & ~ext)[0]:
            assert np.array_equal(vx[ox[i]:ox[i + 1]], vals[off[i]:off[i + 1]])
        # float32 simulation of the continued rows against the full-length float64 transform of
        # the cut bank: per-row error stays at the level of the block rows (a few 1e-6 of the row)
        rng = np.random.default_rng(0)
        x = rng.standard_normal(M)
        X = np.fft.fft(x)
        Xa = X.copy(); Xa[M // 2 + 1:] = 0; Xa[M // 2] *= 0.5
        xa32 = np.fft.ifft(Xa).astype(np.complex64)            # analytic signal as the device holds it
        Xa32 = np.fft.fft(xa32.astype(np.complex128))
        for i in np.nonzero(ext)[0][:6]:
            n_in = off[i + 1] - off[i]
            g = np.zeros(M); g[lo[i]:lo[i] + (ox[i + 1] - ox[i])] = v64x[ox[i]:ox[i + 1]]
            got = np.fft.ifft(g * Xa32)
            ref_band = np.zeros(M); ref_band[lo[i]:lo[i] + n_in] = v64[off[i]:off[i + 1]]
            ref = np.fft.ifft(ref_band * X)
            err = np.abs(got - ref).max() / np.abs(ref).max()
            assert err <= 2e-5, (wavelet, i, err)
