# -*- coding: utf-8 -*-
"""Block ("overlap-save zoom") decomposition of the CWT -- host-side planning.

The reference computes every row as one length-M inverse FFT of
``psih(scale*xi) * xh`` (ssqueezepy/_cwt.py:167-177). On the device that is ~14 GFLOP
and several HBM round trips per transform at N=160k. Two properties of analytic
wavelet filter banks make a far cheaper *equivalent* evaluation possible:

  (1) each row is band-limited: non-negligible on K bins of the M-point grid;
  (2) each row's impulse response is short: all but a 1e-9 fraction of its L1 mass
      lies within +-m samples, m ~ 30 * scale for the default GMW (measured per
      row, in double, from the bank itself -- nothing is assumed about the wavelet).

By (2) the circular convolution over M samples can be evaluated block by block
(overlap-save): a block of P >= 8m consecutive samples of the padded signal,
filtered circularly with the P-periodised wavelet, equals the true output on its
central V = 3P/4 samples up to the truncated tail mass. The P-periodised wavelet's
spectrum is the M-point bank sampled every M/P-th bin -- the very same numbers. By
(1) the block's P-point inverse FFT has only K*P/M non-zero inputs, so it splits
into R' = P/L' independent L'-point FFTs (L' = 128...2048, a "zoom" FFT): output
sample t = q*R' + c is entry q of the FFT of column c, whose inputs are the band
entries times exp(2i*pi*kappa*c/P). L'-point FFTs fit in LDS, adjacent columns are
adjacent output samples (coalesced stores), and nothing but the returned arrays
touches HBM.

Rows whose pass-band is cut by the Nyquist frequency (the few smallest scales) fail (2) as
the reference states them: the cut makes their impulse response decay like 1/t. But the cut
can be moved from the filter to the signal. With x_a the analytic signal of the padded input
(spectrum X_a = xh on bins [0, M/2), half of it at the Nyquist bin, zero above: one length-M
inverse FFT per signal), a row is  ifft(psih * xh) = ifft(g * X_a)  for EVERY g that equals
the un-halved bank on [0, M/2] -- whatever g holds on the bins above M/2 multiplies zeros.
Taking for g the wavelet's own smooth continuation past Nyquist, psih(scale * 2 pi k / M) for
k > M/2 (times a smooth taper where that continuation would be too wide for the LDS FFT),
the filter has a compact response again (margins of 25-40 samples for the default GMW) and
the row runs through the same block kernels, fed with the block spectra of x_a (complex
blocks, P bins each) instead of those of x: `extend_past_nyquist`, classes with
`analytic = 1`. The identity is exact; what differs from the reference is rounding only.

Rows of a non-analytic / non-power-of-two configuration, and Nyquist-cut rows that cannot be
extended, stay on the exact full-length path.

This module measures m per row, picks the block class of every row and builds the
tables the kernel needs (`ssq_cwt_plan_set_blocks`, include/ssq_hip.h).
"""
import numpy as np
import scipy.fft as sfft
from .configs import host_threads

__all__ = ['plan_blocks']

POINTS_PER_WG = {'float32': 4096, 'float64': 2048}   # D = L' * G complex points per workgroup
L_MIN, L_MAX = 128, 2048
P_MIN = 4096


def _is_pow2(v):
    return v >= 1 and (v & (v - 1)) == 0


def _tail_margin(f, tol, denoise):
    """Smallest s with sum_{s' >= s} f[s'] <= tol * sum(f) for every row of the folded
    magnitude response `f` (rows, half + 1); -1 where no such s exists."""
    half = f.shape[1] - 1
    tot = f.sum(1)
    if denoise:
        # the far half is rounding noise for a compact response, but real signal for a
        # long one: never subtract more than 64 eps of the peak
        floor = np.minimum(4 * np.median(f[:, half // 2:], axis=1, keepdims=True),
                           64 * np.finfo(np.float64).eps * f.max(1, keepdims=True))
        f = np.maximum(f - floor, 0)
    cs = np.cumsum(f[:, ::-1], axis=1)[:, ::-1]          # cs[s] = sum_{s' >= s}
    out = np.full(len(f), -1, np.int64)
    for r in range(len(f)):
        ok = np.nonzero(cs[r] <= tol * tot[r])[0]
        if len(ok):
            out[r] = ok[0]
    return out


def _margins(vals, off, lo, M, tol, chunk=32, denoise=False, extended=None):
    """Per row: smallest m such that the impulse response's L1 mass outside
    [-m, m] (circularly, on the M-point grid) is <= tol * total. Double precision.

    `denoise`: subtract 4x the median magnitude of the far half of the response (the
    rounding-noise floor of the double-precision FFT, ~1e-17 of the peak per sample but
    ~1e-13 of the L1 norm once summed over M samples; capped at 64 eps of the peak) before
    accumulating -- needed to resolve tails at the 1e-14 level a float64 transform asks
    for.

    Cost control (the plain way is an M-point FFT per row: 10+ s of host time at
    M = 2^21). The response of a row is measured on the shortest grid that represents it:
      * narrow band (ends at bin hi): the response sampled every 2^d-th point is the
        (M >> d)-point inverse FFT of the same band -- grid 8x the band, >= 8192 points;
      * wide band, short response: the band sampled every s-th bin is the response
        periodised with period M / s -- start at 16384 points and double until the
        margin is below 1/8 of the period (a compact response is then unaffected);
      * a band that reaches the Nyquist bin is cut there by the reference, its response
        decays like 1/t: such rows go to the exact path without being measured -- unless
        `extended[i]` says the band handed in is the row's continuation past Nyquist
        (`extend_past_nyquist`), which is measured like any other wide band."""
    na = len(lo)
    lens = np.diff(off)
    out = np.full(na, M // 2, np.int64)
    todo = {}                                  # (kind, grid length) -> rows
    for i in range(na):
        hi = int(lo[i] + lens[i])
        if lens[i] == 0 or (hi >= M // 2 + 1 and not (extended is not None and extended[i])):
            continue                           # empty, or cut at Nyquist: exact path
        d = 0
        while (M >> (d + 1)) >= 8192 and 8 * hi <= (M >> (d + 1)):
            d += 1
        if d:
            todo.setdefault(('time', M >> d), []).append(i)
        else:
            todo.setdefault(('freq', min(M, 16384)), []).append(i)
    while todo:
        (kind, Md), sel = todo.popitem()
        half, step = Md // 2, M // Md
        ch = max(1, chunk * max(1, (1 << 18) // Md))
        for c0 in range(0, len(sel), ch):
            rows = sel[c0:c0 + ch]
            D = np.zeros((len(rows), Md), np.complex128)
            for r, i in enumerate(rows):
                band = vals[off[i]:off[i + 1]]
                if kind == 'time':
                    D[r, lo[i]:lo[i] + lens[i]] = band
                else:                          # every `step`-th bin of the M-grid
                    k0 = -(-int(lo[i]) // step)
                    sub = band[k0 * step - int(lo[i])::step]
                    D[r, k0:k0 + len(sub)] = sub
            h = np.abs(sfft.ifft(D, axis=-1, workers=host_threads(256)))
            f = h[:, :half + 1].copy()
            f[:, 1:half] += h[:, :half:-1]
            m = _tail_margin(f, tol, denoise)
            for r, i in enumerate(rows):
                if kind == 'time':
                    d = int(np.log2(step))
                    out[i] = min(M // 2, ((int(m[r]) + 1) << d)) if m[r] >= 0 else M // 2
                elif m[r] >= 0 and (8 * m[r] <= Md or Md == M):
                    out[i] = min(M // 2, int(m[r]) + 4)     # periodisation folds a few samples
                elif Md < M:                   # response not compact on this period: retry longer
                    todo.setdefault(('freq', Md * 2), []).append(i)
    return out


NYQ_EXT_GAIN = 8.0    # a continued row's weights past Nyquist: at most this many times its largest in-band weight
                      # (banks at the usual densities stay below 3.5; a coarse bank's first row reaches 16 .. 400)


def extend_past_nyquist(fn, scales, w_hi, vals, off, lo, M, vals64=None):
    """Continue the Nyquist-cut rows of the banded bank past the Nyquist bin (module
    docstring). `fn`: the wavelet's frequency-domain function, evaluated here in float64
    (the continuation only has to be smooth: it multiplies the zeros of the analytic
    signal's spectrum); `w_hi`: upper end of the wavelet's support (`_bank.support_hull`).

    Returns ``(vals_x, off_x, vals64_x, extended)``: the bank with every extended row
    replaced by  [its band with the Nyquist bin un-halved | continuation on bins M/2+1 ...],
    in the bank dtype and in float64, and the per-row flag. Rows that are not cut, or whose
    continuation does not fit, are returned unchanged (flag False)."""
    from scipy.special import erfc
    na, half = len(lo), M // 2
    lens = np.diff(off)
    h = 2 * np.pi / M
    S = M // P_MIN
    extended = np.zeros(na, bool)
    bands, bands64 = [], []
    for i in range(na):
        band = vals[off[i]:off[i + 1]]
        b64 = (band if vals64 is None else vals64[off[i]:off[i + 1]]).astype(np.float64)
        # (at M == P_MIN the single-block class takes the cut rows as they are)
        if M > P_MIN and M % 2 == 0 and lens[i] > 0 and lo[i] + lens[i] == half + 1:
            a = float(np.asarray(scales).reshape(-1)[i])
            k_nat = int(np.ceil(w_hi / (a * h))) + 1         # where the continuation has died out
            # the widest band an LDS FFT takes at the shortest block length, and no wrap to DC
            k_lim = min(int(lo[i]) + (L_MAX - 1) * S, M - S)
            k_end = min(k_nat, k_lim)                         # one past the last bin
            # (a taper needs room to be smooth; the wavelet's own tail needs none)
            if k_end - half >= (2 if k_end == k_nat else max(M // 16, 2)):
                k = np.arange(half + 1, k_end)
                with np.errstate(all='ignore'):
                    e = np.asarray(fn(a * h * k.astype(np.float64)))
                e = np.where(np.isfinite(e), e, 0).real.astype(np.float64)
                if k_end < k_nat:
                    # smooth step 1 -> 0 over (M/2, k_end): equals 1 to 1e-17 at the Nyquist
                    # bin, a Gaussian-shaped response in time
                    e = e * (0.5 * erfc(6 * (2 * (k - half) / float(k_end - half) - 1)))
                # The bins of x_a above Nyquist are rounding noise (~eps |X|), not zeros: a
                # continuation much larger than the in-band weights would amplify it -- a wavelet
                # whose PEAK lies past Nyquist (e.g. the smallest 'bump' scales: in-band weights
                # ~2e-3 of the peak) would lose three digits of its row. Such rows keep the exact path.
                inband = float(np.abs(b64).max()) * (2.0 if len(b64) == 1 else 1.0)
                inband = max(inband, float(np.abs(b64[-1]) * 2))
                if len(e) and float(np.abs(e).max()) > NYQ_EXT_GAIN * inband:
                    bands.append(band); bands64.append(b64)
                    continue
                band = band.copy(); b64 = b64.copy()
                band[-1] = band[-1] * 2; b64[-1] = b64[-1] * 2   # the halving moves into X_a (exact)
                band = np.concatenate([band, e.astype(band.dtype)])
                b64 = np.concatenate([b64, e])
                extended[i] = True
        bands.append(band); bands64.append(b64)
    off_x = np.zeros(na + 1, np.int64)
    np.cumsum([len(b) for b in bands], out=off_x[1:])
    return (np.concatenate(bands) if bands else vals, off_x,
            np.concatenate(bands64) if bands64 else vals64, extended)


def plan_blocks(vals, off, lo, M, N, n1, dtype, vals64=None, tail_tol=None, extension=None):
    """Plan the block decomposition.

    vals/off/lo: banded bank on the M-grid (`_bank.banded_bank`). `vals64`: the
    same band evaluated in float64 if available (margins are then measured on the
    clean wavelet rather than on its float32 rounding noise). `extension`:
    ``(fn, scales, w_hi)`` to continue Nyquist-cut rows past the Nyquist bin
    (`extend_past_nyquist`), or None to leave them on the exact path.
    Returns None when no row qualifies, else a dict of NumPy arrays:
      classes  (nc, 5) int64: P, m (margin), V (valid), nb (blocks per signal), analytic
               (1: the class' blocks are cut from the analytic signal, P bins per spectrum)
      rows     (na, 6) int32: class (-1 = exact path), kappa_lo, K_P, L', G, pbank_off
      pbank    concatenated P-grid band values of the block rows (bank dtype)
      pxi      xi (radian frequency) at the same bins, in the transform's dtype
      ctw      per class: exp(2i*pi*q/P), q in [0, P)   (complex, concatenated)
      ctw_off  (nc + 1) int64 offsets into ctw
      ftw      per L': exp(2i*pi*q/L'), concatenated for L' = 128..2048
      items    dict L' -> (n_items, 4) int32: row, block, c0, class
      generic_rows  int32 indices of rows left on the exact path
    """
    dtype = str(np.dtype(dtype))
    if not _is_pow2(M) or M < P_MIN or dtype not in POINTS_PER_WG:
        return None
    f64 = dtype == 'float64'
    if tail_tol is None:
        # truncated-tail budget relative to the row's L1 norm: far below the rounding
        # error of the transform itself (6e-8 / 1e-16)
        tail_tol = 1e-14 if f64 else 1e-9
    points = POINTS_PER_WG[dtype]
    rdt, cdt = (np.float64, np.complex128) if f64 else (np.float32, np.complex64)
    na = len(lo)
    lens = np.diff(off)
    half = M // 2
    if np.any(lo + lens > half + 1):
        return None                                   # negative-frequency content
    # from here on `vals` / `off` / `lens` describe the bands the kernels apply: the bank's,
    # except for the rows continued past Nyquist
    bank_vals, bank_off = vals, off
    extended = np.zeros(na, bool)
    if extension is not None:
        fn, scales, w_hi = extension
        vals, off, vals64, extended = extend_past_nyquist(fn, scales, w_hi, vals, off, lo, M, vals64)
        lens = np.diff(off)
    margins = _margins(vals if vals64 is None else vals64, off, lo, M, tail_tol, denoise=f64,
                       extended=extended)

    # block classes (P, margin): P = 4096, 8192, ..., M/2 with margin P/8 (valid 3P/4),
    # and the "global" class P = M (one block, no margin needed: the circular convolution
    # over the padded signal *is* the reference's definition). A row takes the shortest
    # block whose margin covers its response, so its margin lies in (P/16, P/8] -- except
    # at the smallest block length, where short responses would waste most of a P/8
    # margin: P_MIN also comes with margins P/32 and P/16 (valid 15/16 and 7/8 of P; each
    # class costs a gather + FFT of the signal's blocks, so a P/64 class does not pay).
    cands = []
    P = P_MIN
    while P < M:
        cands += [(P, den) for den in ((32, 16, 8) if P == P_MIN else (8,))]
        P *= 2
    cands.append((M, 1))
    n_real = len(cands)
    cands = cands + cands                             # the same classes over the analytic signal
    cls_of = np.full(na, -1, np.int64)
    rows = np.zeros((na, 6), np.int32)
    rows[:, 0] = -1
    pb, px, pb_off = [], [], 0
    for i in range(na):
        if lens[i] == 0:
            continue
        for c, (P, mden) in enumerate(cands):
            if (c >= n_real) != bool(extended[i]):
                continue
            if P < M and mden * margins[i] > P:
                continue
            S = M // P
            k_lo = -(-int(lo[i]) // S)                       # ceil(lo / S)
            k_hi = (int(lo[i]) + int(lens[i]) - 1) // S       # last P-grid bin in band
            KP = k_hi - k_lo + 1
            if KP <= 0:
                k_lo, KP = int(lo[i]) // S, 0
            Lp = L_MIN
            while Lp < KP:
                Lp *= 2
            if Lp > L_MAX:
                break                      # band too wide for an LDS FFT: exact path
            G = points // Lp
            if P // Lp < G:
                continue                   # fewer columns than a workgroup handles
            cls_of[i] = c
            sel = (np.arange(k_lo, k_lo + KP) * S - int(lo[i])) + int(off[i])
            pb.append(vals[sel])
            # xi at the band's bins, exactly the M-grid values the reference uses
            # (k * 2pi/M formed in double, stored in the wavelet dtype: wavelets.py:473-484)
            px.append((np.arange(k_lo, k_lo + KP) * S * (2 * np.pi / M)
                       ).astype(rdt))
            rows[i] = (c, k_lo, KP, Lp, G, pb_off)
            pb_off += KP
            break
    extended &= cls_of >= 0            # (a continuation too wide for its block length: exact path)
    used = sorted(set(int(c) for c in cls_of if c >= 0))
    if not used:
        return None
    remap = {c: j for j, c in enumerate(used)}
    classes = np.zeros((len(used), 5), np.int64)
    ctw, ctw_off = [], [0]
    for c in used:
        P, mden = cands[c]
        if P == M:
            m, V, nb, t0 = 0, M, 1, 0
        else:
            m = P // mden
            V = P - 2 * m
            nb = -(-N // V)
        classes[remap[c]] = (P, m, V, nb, int(c >= n_real))
        q = np.arange(P)
        w = np.exp(2j * np.pi * q / P)
        ctw.append(w.astype(cdt))
        ctw_off.append(ctw_off[-1] + P)
    for i in range(na):
        if rows[i, 0] >= 0:
            rows[i, 0] = remap[int(rows[i, 0])]
    # FFT twiddles per L'
    ftw, ftw_off = [], {}
    o = 0
    Lp = L_MIN
    while Lp <= L_MAX:
        ftw.append(np.exp(2j * np.pi * np.arange(Lp) / Lp).astype(cdt))
        ftw_off[Lp] = o
        o += Lp
        Lp *= 2
    # work items per L': (row, block, c0, class)
    items = {}
    for i in range(na):
        c = rows[i, 0]
        if c < 0:
            continue
        P, m, V, nb = classes[c, :4]
        Lp, G = int(rows[i, 3]), int(rows[i, 4])
        Rp = int(P) // Lp
        b, c0 = np.meshgrid(np.arange(nb), np.arange(0, Rp, G), indexing='ij')
        it = np.stack([np.full(b.size, i), b.ravel(), c0.ravel(),
                       np.full(b.size, c)], axis=1).astype(np.int32)
        items.setdefault(Lp, []).append(it)
    items = {Lp: np.concatenate(v) for Lp, v in items.items()}
    generic_rows = np.nonzero(rows[:, 0] < 0)[0].astype(np.int32)
    return dict(classes=classes, rows=rows,
                pbank=(np.concatenate(pb) if pb else np.zeros(1, vals.dtype)
                       ).astype(vals.dtype),
                pxi=(np.concatenate(px) if px else np.zeros(1, rdt)),
                ctw=np.concatenate(ctw), ctw_off=np.array(ctw_off, np.int64),
                ftw=np.concatenate(ftw), ftw_off=ftw_off, items=items,
                generic_rows=generic_rows, margins=margins, extended=extended,
                band_vals=vals, band_off=off)
