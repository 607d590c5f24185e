# -*- coding: utf-8 -*-
"""Banded filter bank: the wavelet rows `psih(scale_i * xi_k)` restricted to the
contiguous run of DFT bins on which each row is non-negligible.

The reference evaluates and stores the dense `(na, M)` bank
(``Wavelet.Psih``, ssqueezepy/wavelets.py:135-160; used at _cwt.py:169-171).
Every built-in wavelet is a single bump in frequency, so a row at scale `a` is
non-zero only for ``a * xi_k`` inside the bump's support: at N=160k / 300 scales
that is 12.8 % of the dense array (hard part 4). Values inside
the band are evaluated with the reference's own NumPy operation order in the
wavelet dtype (see wavelets.py), so they equal the dense bank's entries bit for
bit; entries outside are below `tol` times the wavelet's peak and are dropped
(default tol = 1e-3 * eps(dtype): orders of magnitude under the transform's own
rounding error).
"""
import numpy as np

from .wavelets import xi_grid

__all__ = ['banded_bank', 'support_hull']


def support_hull(fn, dtype, tol, w_extent):
    """Smallest interval [w_lo, w_hi] of the real line outside which
    ``|fn(w)| <= tol * max|fn|``, searched on ``[-w_extent, w_extent]`` with a
    1e-3 grid near the origin (coarser far out); padded by one grid step."""
    span = 64.0
    while True:
        step = span / 65536
        w = np.arange(-span, span + step, step)
        with np.errstate(all='ignore'):
            v = np.abs(np.asarray(fn(w.astype(dtype)))).astype(np.float64)
        v[~np.isfinite(v)] = 0
        peak = v.max()
        if peak <= 0:
            raise ValueError("wavelet evaluates to zero on [-%g, %g]" % (span, span))
        keep = np.nonzero(v > tol * peak)[0]
        lo, hi = keep[0], keep[-1]
        if (lo > 0 and hi < len(w) - 1) or span >= w_extent:
            return w[max(lo - 1, 0)], w[min(hi + 1, len(w) - 1)]
        span *= 4


_PAR_MIN = 1 << 22        # elements from which the evaluation is spread over threads


def _evaluate(wavelet, w_flat):
    """``wavelet.fn(w_flat)``. The built-in families are elementwise NumPy expressions, so a
    large argument (config 5: 1.4e8 values, 11 s in one piece) is evaluated in contiguous
    pieces on a thread pool -- NumPy releases the GIL inside its loops; every element goes
    through the same operations as in one piece (checked bit for bit by
    tests/test_design_vs_golden.py::test_bank_evaluation_in_pieces). A user-supplied
    function is called once, whole: nothing is known about it."""
    n = len(w_flat)
    if getattr(wavelet, 'family', None) is None or n < _PAR_MIN:
        return np.asarray(wavelet.fn(w_flat))
    import os
    from concurrent.futures import ThreadPoolExecutor
    from .configs import host_threads
    workers = max(1, min(host_threads(32), n // (_PAR_MIN // 4)))
    edges = np.linspace(0, n, 4 * workers + 1).astype(np.int64)

    def piece(i):
        with np.errstate(all='ignore'):
            return np.asarray(wavelet.fn(w_flat[edges[i]:edges[i + 1]]))
    with ThreadPoolExecutor(workers) as pool:
        parts = list(pool.map(piece, range(len(edges) - 1)))
    return np.concatenate(parts)


def banded_bank(wavelet, scales, M, tol=None, nohalf=False):
    """Evaluate `wavelet` at `scales` (1-D array in the wavelet dtype) on the
    M-point DFT grid and return ``(values, band_off, band_lo)``:
    row i occupies bins ``[band_lo[i], band_lo[i] + len_i)`` with
    ``len_i = band_off[i+1] - band_off[i]`` and values
    ``values[band_off[i]:band_off[i+1]]``. `nohalf=False` halves the Nyquist bin
    (wavelets.py:86-95), as `cwt` requires."""
    dt = np.dtype(wavelet.dtype)
    eps = np.finfo(dt).eps
    tol = (1e-3 * eps) if tol is None else float(tol)
    scales = np.asarray(scales, dtype=dt).reshape(-1)
    na = len(scales)
    M = int(M)
    xi = xi_grid(M, dtype=dt)
    half = M // 2
    h = 2 * np.pi / M

    w_lo, w_hi = support_hull(wavelet.fn, dt, tol,
                              w_extent=float(scales.max()) * np.pi * 1.01 + 1)
    los = np.empty(na, np.int64)
    his = np.empty(na, np.int64)
    dense = np.zeros(na, bool)
    for i, a in enumerate(scales.astype(np.float64)):
        k_lo = int(np.floor(w_lo / (a * h))) - 1
        k_hi = int(np.ceil(w_hi / (a * h))) + 1
        if k_lo < 0 and w_lo < 0:
            # negative-frequency content: not a single run of [0, M) -> dense row
            dense[i] = True
            los[i], his[i] = 0, M
        else:
            los[i] = min(max(k_lo, 0), half)
            his[i] = min(max(k_hi, los[i]), half) + 1
    lens = his - los
    band_off = np.zeros(na + 1, np.int64)
    np.cumsum(lens, out=band_off[1:])
    # flat evaluation, exactly `fn(scale * xi)` of the dense path, band entries only
    row_of = np.repeat(np.arange(na), lens)
    k_of = np.arange(band_off[-1]) - np.repeat(band_off[:-1], lens) + np.repeat(los, lens)
    w_flat = scales[row_of] * xi[k_of]
    with np.errstate(all='ignore'):
        vals = _evaluate(wavelet, w_flat)
    if np.iscomplexobj(vals):
        if vals.imag.sum() / vals.real.sum() < 1e-8:
            vals = vals.real
        else:
            raise NotImplementedError("complex-valued frequency-domain wavelets "
                                      "are not supported by the HIP path")
    vals = np.ascontiguousarray(vals, dtype=dt)
    if not nohalf and M % 2 == 0:
        at_nyq = (k_of == half)
        vals[at_nyq] /= 2
    return vals, band_off, los.astype(np.int32)
