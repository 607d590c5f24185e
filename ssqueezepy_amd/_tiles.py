# -*- coding: utf-8 -*-
"""Column-tile evaluation of `ssq_cwt` -- host-side planning.

The reference computes every row of `Wx` and `dWx` as a length-M inverse FFT
(ssqueezepy/_cwt.py:167-177) and then scatters `Wx` along the scale axis
(ssqueezing.py:122-146 -> algos.py:859-953). Row by row that forces `Wx` (and a bin
map) through HBM twice: once out of the transform, once into the reassignment. The
reassignment wants the opposite order -- *all rows of a few columns* -- so that the
column's `Tx` bins can live in LDS. The block kernels cannot produce that order (a row
is cheap only as a long run of time samples), but an *oversampled* row can be produced
in any order from a short description:

  every row i is band-limited to K_i bins around bin kc_i of the M-point grid, so its
  demodulated form  a_i[n] = Wx_i[n] * exp(-2i pi kc_i n / M)  is a low-pass signal of
  bandwidth K_i/M: it is fully described by its samples u_i[q] = a_i[q R_i] at any rate
  M / R_i >= sigma * K_i (sigma >= 2 here), i.e. by an (M / R_i)-point inverse FFT of the
  band -- R_i times shorter than the reference's. Between those samples a_i is recovered
  by convolution with a compact kernel phi of W = 8 taps (Kaiser-Bessel, the gridding
  kernel of non-uniform FFTs; its pass-band droop is divided out of the band beforehand):

      Wx_i[n]  = e^{2i pi kc n / M} * sum_t phi(n/R - q0 - t)  u_i[q0 + t]
      dWx_i[n] = e^{2i pi kc n / M} * (i theta a + sum_t phi'(n/R - q0 - t) u_i[q0 + t] / R) / dt

  with theta = 2 pi kc / M (the derivative of the band-limited interpolant *is* the
  reference's spectral derivative 1j*xi, up to the aliasing error of phi, ~1e-7).

A workgroup of the tile kernel (csrc/ssq_cwt_tiles.hip) therefore owns 64 columns x all
`na` bins of `Tx` in LDS, walks the rows in ascending order, evaluates `Wx`, `dWx` for its
columns from 16 samples of u_i per row, writes `Wx` once, reassigns into LDS in the
reference's summation order, and writes `Tx` once: HBM traffic = the returned arrays.

Rows that are not oversampled enough (R < R_MIN: the widest ~20 % of the bank) or
whose band is cut by the Nyquist frequency keep the block / exact kernels, which leave
`Wx` and a bin map in HBM; the tile kernel reads those rows back (`kind 0`).

This module picks R_i per row, builds the compensated band values, the kernel tables, and
the row / step / segment lists the kernel walks
(`ssq_cwt_plan_set_tiles`, include/ssq_hip.h). Measured accuracy of the interpolation on
the N=160k GMW bank: 2.7e-7 of max|Wx| (float32 samples and weights), 9e-7 of max|dWx|
-- below the float32 FFT's own error (tests compare against the oracle at 1e-5).
"""
import numpy as np
from scipy.special import i0, i1

__all__ = ['plan_tiles', 'W_TAPS', 'SIGMA_MIN', 'R_MIN']

W_TAPS = 8            # taps per output sample
SIGMA_MIN = 2.0       # least oversampling of a decimated row
R_MIN = 4             # least decimation for which a row leaves the block kernels
R_MAX = 4096
L_MINLEN = 64         # shortest decimated row
MIN_CLASS_ROWS = 16   # a shorter run of rows with a decimation of its own joins the run before it ...
MERGE_MAX_OCTAVES = 3 # ... when that costs it at most 8 times the samples it needs ...
MERGE_MIN_ROWS = 64   # ... in a plan with at least this many interpolated rows
COLS = 64             # columns per workgroup of the ordered tile kernel (one per lane)
NA_MAX = 512          # rows: a packed record holds 9 bits of row; the float64 tile of the default
                      # kernel (16 B per cell) takes 32 columns up to 318 rows, 16 columns beyond
                      # (the ordered kernel, SSQ_TILE_ORDER=ordered, stops at 318 rows)
RSUB = 4              # rows per step (TILE_G of the kernels)
STEPS_PER_TICKET = 1  # (steps are handed out one at a time)
KIND_READBACK, KIND_INTERP = 0, 1


def _kb_beta(W, sigma):
    # Beatty et al. 2005: Kaiser-Bessel shape parameter for oversampling `sigma`
    return np.pi * np.sqrt((W / sigma) ** 2 * (sigma - 0.5) ** 2 - 0.8)


def kb_phi(t, W=W_TAPS, sigma=SIGMA_MIN):
    """Kaiser-Bessel kernel phi(t), t in decimated samples, support |t| < W/2."""
    beta = _kb_beta(W, sigma)
    a = 1.0 - (2.0 * np.asarray(t, dtype=np.float64) / W) ** 2
    return np.where(a > 0, i0(beta * np.sqrt(np.maximum(a, 0.0))), 0.0) / i0(beta)


def kb_dphi(t, W=W_TAPS, sigma=SIGMA_MIN):
    """d phi / d t (the edge jump of size 1/I0(beta) ~ 1e-7 is ignored)."""
    beta = _kb_beta(W, sigma)
    t = np.asarray(t, dtype=np.float64)
    a = 1.0 - (2.0 * t / W) ** 2
    s = np.sqrt(np.maximum(a, 1e-300))
    return np.where(a > 0, i1(beta * s) * beta * (-4.0 * t / W ** 2) / s, 0.0) / i0(beta)


def kb_phihat(f, W=W_TAPS, sigma=SIGMA_MIN):
    """Fourier transform of phi at f cycles per decimated sample."""
    beta = _kb_beta(W, sigma)
    z = np.sqrt(beta ** 2 - (np.pi * W * np.asarray(f, dtype=np.float64)) ** 2 + 0j)
    return (W * np.sinh(z) / z / i0(beta)).real


def _ilog2(v):
    return int(v).bit_length() - 1


def plan_tiles(vals, off, lo, M, N, n1, dt, block_rows, group, row_scale=None,
               r_min=R_MIN, sigma_min=SIGMA_MIN):
    """Plan the column-tile path of a float32 plan.

    vals/off/lo: banded bank (`_bank.banded_bank`); `block_rows`: bool (na,), rows the
    block kernels evaluate (compact impulse response, band below Nyquist) -- only those
    are candidates; `group`: signals per launch (the plan's launch group).
    Returns None when no row qualifies, else a dict of arrays (see `ssq_cwt_tiles_desc`)
    plus `interp_rows` (bool mask)."""
    na = len(lo)
    lens = np.diff(off).astype(np.int64)
    if M & (M - 1) or M > 2 ** 23 or na >= NA_MAX or na * N >= 2 ** 29:
        return None                              # the Tx tile must fit one CU's LDS (see NA_MAX)
    lgR = np.full(na, -1, np.int64)
    for i in range(na):
        K = int(lens[i])
        if not block_rows[i] or K < 1:
            continue
        lg = _ilog2(max(1, int(M / (sigma_min * K))))
        R = min(1 << lg, R_MAX, M // L_MINLEN)
        if R >= r_min and (M // R) >= sigma_min * K:
            lgR[i] = _ilog2(R)
    interp = lgR >= 0
    if not interp.any():
        return None
    # Short classes join their predecessor. The kernels keep one class's interpolation weights in registers and
    # re-read them at a class change, and the host deals whole classes to wavefronts (csrc/ssq_cwt_tiles.hip): a run
    # of a few rows with a decimation of its own -- 'log-piecewise' scales double every 8 rows in their upper part --
    # costs more in class changes than its rows cost in arithmetic. A row may always be sampled MORE densely than it
    # needs (smaller R: sigma grows, the kernel's error shrinks), so a run shorter than MIN_CLASS_ROWS takes the
    # decimation of the run before it when that one is smaller (rows are in scale order: it is). Its samples take
    # 2^d times the room of a class that had few rows and short rows anyway.
    i = 0 if interp.sum() >= MERGE_MIN_ROWS else na              # (a handful of rows: more wavefronts than items anyway)
    prev_lg, prev_len = None, 0
    while i < na:
        j = i
        while j < na and lgR[j] == lgR[i]:
            j += 1
        lg = int(lgR[i])
        if lg >= 0:
            if prev_lg is not None and 0 <= prev_lg < lg <= prev_lg + MERGE_MAX_OCTAVES and (j - i) < MIN_CLASS_ROWS:
                lgR[i:j] = prev_lg                       # (joins the run before it; the merged run goes on growing)
                prev_len += j - i
            else:
                prev_lg, prev_len = lg, j - i
        else:
            prev_lg, prev_len = None, 0
        i = j

    # ---- classes (one per decimation), intermediates layout, compensated band values
    used = sorted(set(int(v) for v in lgR[interp]))
    classes, cls_index = [], {}
    u_prefix = 0                                  # complex entries per signal before the class
    for lg in used:
        rows_c = np.nonzero(lgR == lg)[0]
        L = M >> lg
        cls_index[lg] = len(classes)
        classes.append([L, len(rows_c), u_prefix, lg])
        u_prefix += len(rows_c) * L
    u_total = u_prefix
    if u_total * group >= 2 ** 31:
        return None
    tb, irows, tb_off = [], [], 0
    r_in_class = {lg: 0 for lg in used}
    kc_of = np.zeros(na, np.int64)
    ubase_of = np.zeros(na, np.int64)
    for i in np.nonzero(interp)[0]:
        lg = int(lgR[i]); L = M >> lg
        K = int(lens[i]); kc = int(lo[i]) + K // 2
        k = np.arange(int(lo[i]), int(lo[i]) + K)
        v = vals[off[i]:off[i + 1]].astype(np.float64)
        v = v / (kb_phihat((k - kc) / L, sigma=sigma_min) * M)
        if row_scale is not None:
            v = v * float(row_scale[i])
        tb.append(v.astype(np.float32))
        r = r_in_class[lg]; r_in_class[lg] += 1
        kc_of[i] = kc
        ubase_of[i] = r * L
        irows.append([i, int(lo[i]), K, kc, L, tb_off, cls_index[lg], r])
        tb_off += K
    irows = np.asarray(irows, dtype=np.int64)

    # ---- weights: per class, per phase r = n mod R, per tap t = 0..W-1: the pair
    # phi(r/R - (t - W/2 + 1)), phi'(...) / (R dt)
    wt, wt_off = [], {}
    o = 0
    taps = np.arange(W_TAPS) - (W_TAPS // 2 - 1)
    for lg in used:
        R = 1 << lg
        fr = (np.arange(R) / R)[:, None] - taps[None, :]
        tab = np.stack([kb_phi(fr, sigma=sigma_min),
                        kb_dphi(fr, sigma=sigma_min) / (R * float(dt))], axis=2)   # (R, W, 2)
        wt.append(tab.reshape(R, 2 * W_TAPS).astype(np.float32))
        wt_off[lg] = o
        o += R
    wtab = np.concatenate(wt)

    # ---- steps (RSUB rows each, one kind / class per step) and segments (runs of steps)
    rowdesc, segs = [], []
    i = 0
    while i < na:
        kind_lg = int(lgR[i])
        j = i
        while j < na and int(lgR[j]) == kind_lg:
            j += 1
        first = len(rowdesc) // RSUB
        for r in range(i, j):
            if kind_lg >= 0:
                theta = np.float32(2.0 * np.pi * kc_of[r] / M / float(dt))
                rowdesc.append([r, int(ubase_of[r]), int(kc_of[r]), int(theta.view(np.int32))])
            else:
                rowdesc.append([r, 0, 0, 0])
        while len(rowdesc) % RSUB:                # pad: the last row again, flagged (sign bit)
            last = rowdesc[-1]
            rowdesc.append([(last[0] & 0xFFFF) - 2 ** 31, last[1], last[2], last[3]])
        nsteps = len(rowdesc) // RSUB - first
        if kind_lg >= 0:
            L, nrows_c, upre, _ = classes[cls_index[kind_lg]]
            segs.append([KIND_INTERP, first, nsteps, kind_lg, wt_off[kind_lg], nrows_c * L,
                         L - 1, group * upre])
        else:
            segs.append([KIND_READBACK, first, nsteps, 0, 0, 0, 0, 0])
        i = j
    while (len(rowdesc) // RSUB) % STEPS_PER_TICKET:       # whole groups of steps: pad with no-op steps
        last = rowdesc[-1]
        for _ in range(RSUB):
            rowdesc.append([(last[0] & 0xFFFF) - 2 ** 31, last[1], last[2], last[3]])
        if segs[-1][0] == KIND_READBACK:
            segs[-1][2] += 1
        else:
            segs.append([KIND_READBACK, len(rowdesc) // RSUB - 1, 1, 0, 0, 0, 0, 0])
    rowdesc = np.asarray(rowdesc, dtype=np.int32)
    segs = np.asarray(segs, dtype=np.int32)

    return dict(segs=segs, rows=rowdesc, wtab=wtab,
                tbank=np.concatenate(tb), irows=irows.astype(np.int64),
                classes=np.asarray(classes, dtype=np.int64), u_total=int(u_total),
                interp_rows=interp, lgR=lgR)
