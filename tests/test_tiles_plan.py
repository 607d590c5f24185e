# -*- coding: utf-8 -*-
"""Host planning of the column-tile path (ssqueezepy_amd/_tiles.py): structure of the tables
the kernel walks, and the mathematics they encode -- a NumPy restatement of what the tile
kernel computes from them (decimated baseband samples -> 8-tap Kaiser-Bessel interpolation and
its derivative -> modulation) against the reference's definition (dense bank, full-length
inverse FFT, spectral derivative), in float64 so that only the method's own error shows."""
import numpy as np
import scipy.fft as sfft

from ssqueezepy_amd._bank import banded_bank
from ssqueezepy_amd._blocks import plan_blocks
from ssqueezepy_amd._tiles import plan_tiles, W_TAPS, RSUB, KIND_INTERP
from ssqueezepy_amd.padding import pad_geometry
from ssqueezepy_amd.scales import process_scales
from ssqueezepy_amd.wavelets import Wavelet
from conftest import two_chirps


def _plan(N, nv, group=2, dt=1.0):
    wav = Wavelet()
    scales = np.asarray(process_scales('log', N, wav, nv=nv), dtype='float32').reshape(-1)
    M, n1, n2 = pad_geometry(N)
    vals, off, lo = banded_bank(wav, scales, M)
    bp = plan_blocks(vals, off, lo, M, N, n1, 'float32')
    tp = plan_tiles(vals, off, lo, M, N, n1, dt, bp['rows'][:, 0] >= 0, group)
    return wav, scales, M, n1, n2, vals, off, lo, bp, tp


def test_tile_tables_cover_every_row_once_in_order():
    wav, scales, M, n1, n2, vals, off, lo, bp, tp = _plan(6000, 16)
    na = len(scales)
    rows, segs = tp['rows'], tp['segs']
    from ssqueezepy_amd._tiles import STEPS_PER_TICKET
    assert len(rows) % (STEPS_PER_TICKET * RSUB) == 0        # whole groups of steps
    real = rows[rows[:, 0] >= 0, 0]
    assert np.array_equal(real, np.arange(na))               # ascending: the summation order
    pads = rows[rows[:, 0] < 0, 0]
    assert np.all((pads & 0xFFFF) < na)
    assert segs[0, 1] == 0 and np.array_equal(segs[1:, 1], np.cumsum(segs[:-1, 2]))
    assert segs[:, 2].sum() == len(rows) // RSUB
    interp = tp['interp_rows']
    assert interp.sum() > 0.6 * na and not interp[bp['rows'][:, 0] < 0].any()
    for kind, first, nsteps, lgR, woff, stride, lmask, base in segs:
        r = rows[first * RSUB:(first + nsteps) * RSUB, 0]
        r = r[r >= 0]                                        # (pad rows repeat a row, flagged)
        assert np.all(interp[r] == (kind == KIND_INTERP))
        if kind == KIND_INTERP:
            L = lmask + 1
            assert L == M >> lgR and np.all(tp['lgR'][r] == lgR)
            assert np.all(2 * np.diff(off)[r] <= L)          # at least 2x oversampled
    # intermediates: classes tile the buffer without gaps
    cls = tp['classes']
    assert cls[0, 2] == 0 and np.array_equal(cls[1:, 2], np.cumsum(cls[:-1, 0] * cls[:-1, 1]))
    assert tp['u_total'] == int((cls[:, 0] * cls[:, 1]).sum())


def test_interpolation_reproduces_the_full_length_transform():
    N, dt = 3000, 0.5
    wav, scales, M, n1, n2, vals, off, lo, bp, tp = _plan(N, 12, dt=dt)
    x = two_chirps(N, seed=4)
    X = sfft.fft(np.pad(x, (n1, n2), mode='reflect'))
    xi = 2 * np.pi * np.fft.fftfreq(M)
    wtab, irows, classes = tp['wtab'].astype(np.float64), tp['irows'], tp['classes']
    woff = {int(c[3]): o for c, o in zip(classes, np.cumsum([0] + [1 << int(c[3]) for c in classes[:-1]]))}
    n = np.arange(n1, n1 + N)
    worst = [0.0, 0.0]
    tb = tp['tbank'].astype(np.float64)
    for row, blo, K, kc, L, tboff, ci, ridx in irows[::3]:
        lgR = int(classes[ci, 3]); R = 1 << lgR
        # the kernel's arithmetic, in double
        Z = np.zeros(L, complex)
        k = np.arange(blo, blo + K)
        Z[(k - kc) % L] = X[k] * tb[tboff:tboff + K]
        u = sfft.ifft(Z) * L                                 # unnormalised inverse, as rocFFT's
        q0, ph = n >> lgR, n & (R - 1)
        w = wtab[woff[lgR] + ph]                             # (N, 16): phi_t, phi'_t interleaved
        a = np.zeros(N, complex); da = np.zeros(N, complex)
        for t in range(W_TAPS):
            s = u[(q0 - (W_TAPS // 2 - 1) + t) % L]
            a += w[:, 2 * t] * s; da += w[:, 2 * t + 1] * s
        theta = 2 * np.pi * kc / M / dt
        tw = np.exp(2j * np.pi * ((kc * n) % M) / M)
        Wk, Dk = tw * a, tw * (1j * theta * a + da)
        # the reference's definition
        Y = np.zeros(M, complex)
        Y[k] = X[k] * vals[off[row]:off[row + 1]].astype(np.float64)
        Wr = sfft.ifft(Y)[n1:n1 + N]
        Dr = sfft.ifft(Y * 1j * xi / dt)[n1:n1 + N]
        worst[0] = max(worst[0], np.abs(Wk - Wr).max() / np.abs(Wr).max())
        worst[1] = max(worst[1], np.abs(Dk - Dr).max() / np.abs(Dr).max())
    # method error only (float32 tables): well under the float32 FFT's own 2.5e-6
    assert worst[0] < 5e-7 and worst[1] < 2e-6, worst
