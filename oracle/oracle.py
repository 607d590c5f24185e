# -*- coding: utf-8 -*-
"""CPU oracle for the cwt / stft / ssq_cwt / ssq_stft forward path.

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py; never by ssqueezepy_amd/ (the product path fails
loudly without its HIP library instead of falling back to this).

Two layers:
  * loop nests (phase transform, fused reassignment, indexed sum, framing):
    `oracle/ssq_oracle.c`, a C restatement of ssqueezepy/algos.py:172-250,
    720-740, 794-816, 859-984 and utils/stft_utils.py:69-98, loaded via ctypes;
  * transforms: NumPy/scipy.fft restatements of `cwt` (ssqueezepy/_cwt.py:167-177,
    255-306: pad -> fft -> Psih*xh -> ifft [-> *1j*xi/dt -> ifft] -> unpad) and
    `stft` (ssqueezepy/_stft.py:127-147, 160-167: pad -> buffer -> *window -> rfft).
The filter-bank / scale / frequency-grid *design* (host NumPy code shared with the
product, ssqueezepy_amd/{wavelets,scales,ssqueezing}.py) is pinned separately,
value for value, against fixtures generated from the reference
(tests/golden/design_*.npz).

Pinned against the reference itself by tests/test_oracle_vs_golden.py using the
fixtures written by oracle/gen_golden.py (run in the build container, where
/root/reference is importable with the numba stand-in under oracle/refshim).
"""
import ctypes
import os
import subprocess
import numpy as np
import scipy.fft as sfft

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(HERE, '_build', 'libssq_oracle.so')

TYPING_NUMBA, TYPING_NUMPY = 0, 1
GRID = {'log': 0, 'log-piecewise': 1, 'log_piecewise': 1, 'linear': 2, 'lin': 2}

_lib = None


def build(force=False):
    """Compile oracle/ssq_oracle.c (gcc) into oracle/_build/."""
    if force or not os.path.isfile(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) <
            os.path.getmtime(os.path.join(HERE, 'ssq_oracle.c'))):
        subprocess.check_call(['make', '-C', HERE, '-s'])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _params(grid_params):
    p = np.zeros(5, dtype=np.float64)
    p[:len(grid_params)] = grid_params
    return p


def _is32(x):
    return x.dtype in (np.complex64, np.float32)


# ------------------------------------------------------------- loop nests
def phase_cwt(Wx, dWx, gamma, typing=TYPING_NUMBA):
    na, n = Wx.shape
    L = lib()
    if _is32(Wx):
        Wx, dWx = _c(Wx, np.complex64), _c(dWx, np.complex64)
        out = np.empty((na, n), np.float32)
        L.orc_phase_cwt_f32(_p(Wx), _p(dWx), _p(out), ctypes.c_int64(na),
                            ctypes.c_int64(n), ctypes.c_float(gamma),
                            ctypes.c_int(typing))
    else:
        Wx, dWx = _c(Wx, np.complex128), _c(dWx, np.complex128)
        out = np.empty((na, n), np.float64)
        L.orc_phase_cwt_f64(_p(Wx), _p(dWx), _p(out), ctypes.c_int64(na),
                            ctypes.c_int64(n), ctypes.c_double(gamma),
                            ctypes.c_int(typing))
    return out


def phase_stft(Sx, dSx, Sfs, gamma, typing=TYPING_NUMBA):
    na, n = Sx.shape
    L = lib()
    if _is32(Sx):
        Sx, dSx = _c(Sx, np.complex64), _c(dSx, np.complex64)
        Sfs = _c(Sfs, np.float32)
        out = np.empty((na, n), np.float32)
        L.orc_phase_stft_f32(_p(Sx), _p(dSx), _p(Sfs), _p(out),
                             ctypes.c_int64(na), ctypes.c_int64(n),
                             ctypes.c_float(gamma), ctypes.c_int(typing))
    else:
        Sx, dSx = _c(Sx, np.complex128), _c(dSx, np.complex128)
        Sfs = _c(Sfs, np.float64)
        out = np.empty((na, n), np.float64)
        L.orc_phase_stft_f64(_p(Sx), _p(dSx), _p(Sfs), _p(out),
                             ctypes.c_int64(na), ctypes.c_int64(n),
                             ctypes.c_double(gamma), ctypes.c_int(typing))
    return out


def _const_vec(const, na, is32):
    """Per-row weights as the reference materialises them (algos.py:66-79):
    a scalar becomes a vector in the data dtype; a float64 vector with complex64
    data stays float64 (and the accumulate happens in double)."""
    const = np.asarray(const)
    if const.size != na:
        return _c(np.full(na, float(const)),
                  np.float32 if is32 else np.float64), 0
    const = const.reshape(-1)
    if is32 and const.dtype == np.float64:
        return _c(const, np.float64), 1
    return _c(const, np.float32 if is32 else np.float64), 0


def ssqueeze(Wx, dWx, grid, grid_params, const, gamma, flipud=False, Sfs=None,
             typing=TYPING_NUMBA, out=None, get_k=False, parallel=False):
    """Fused phase + bin + accumulate (`ssqueeze_fast`, algos.py:126-150).
    grid: 'log' (vlmin, dvl) | 'log-piecewise' (vlmin0, vlmin1, dvl0, dvl1, idx1)
    | 'linear' (vmin, dv). Returns Tx (and the per-point bin map if `get_k`)."""
    na, n = Wx.shape
    L = lib()
    is32 = _is32(Wx)
    cdt = np.complex64 if is32 else np.complex128
    Wx, dWx = _c(Wx, cdt), _c(dWx, cdt)
    if out is None:
        out = np.zeros((na, n), cdt)
    cst, cst64 = _const_vec(const, na, is32)
    p = _params(grid_params)
    kmap = np.empty((na, n), np.int32) if get_k else None
    kp = _p(kmap) if get_k else None
    sfs = None
    if Sfs is not None:
        sfs = _c(Sfs, np.float32 if is32 else np.float64)
    if is32:
        L.orc_ssq_f32(_p(Wx), _p(dWx), _p(sfs) if sfs is not None else None,
                      _p(out), _p(cst), ctypes.c_int(cst64), ctypes.c_int64(na),
                      ctypes.c_int64(n), ctypes.c_double(gamma),
                      ctypes.c_int(GRID[grid]), _p(p), ctypes.c_int(bool(flipud)),
                      ctypes.c_int(typing), kp, ctypes.c_int(bool(parallel)))
    else:
        L.orc_ssq_f64(_p(Wx), _p(dWx), _p(sfs) if sfs is not None else None,
                      _p(out), _p(cst), ctypes.c_int64(na), ctypes.c_int64(n),
                      ctypes.c_double(gamma), ctypes.c_int(GRID[grid]), _p(p),
                      ctypes.c_int(bool(flipud)), ctypes.c_int(typing), kp,
                      ctypes.c_int(bool(parallel)))
    return (out, kmap) if get_k else out


def indexed_sum(Wx, w, grid, grid_params, const, flipud=False,
                typing=TYPING_NUMBA, out=None, parallel=False):
    """Bin + accumulate from a given phase transform `w` (`indexed_sum_onfly`,
    algos.py:153-169)."""
    na, n = Wx.shape
    L = lib()
    is32 = _is32(Wx)
    cdt = np.complex64 if is32 else np.complex128
    rdt = np.float32 if is32 else np.float64
    Wx, w = _c(Wx, cdt), _c(w, rdt)
    if out is None:
        out = np.zeros((na, n), cdt)
    cst, cst64 = _const_vec(const, na, is32)
    p = _params(grid_params)
    if is32:
        L.orc_indexed_sum_f32(_p(Wx), _p(w), _p(out), _p(cst),
                              ctypes.c_int(cst64), ctypes.c_int64(na),
                              ctypes.c_int64(n), ctypes.c_int(GRID[grid]), _p(p),
                              ctypes.c_int(bool(flipud)), ctypes.c_int(typing),
                              ctypes.c_int(bool(parallel)))
    else:
        L.orc_indexed_sum_f64(_p(Wx), _p(w), _p(out), _p(cst),
                              ctypes.c_int64(na), ctypes.c_int64(n),
                              ctypes.c_int(GRID[grid]), _p(p),
                              ctypes.c_int(bool(flipud)),
                              ctypes.c_int(bool(parallel)))
    return out


def replace_under_abs(w, ref, value, replacement):
    L = lib()
    w = np.array(w, copy=True)
    if _is32(ref):
        ref = _c(ref, np.complex64)
        w = _c(w, np.float32)
        L.orc_replace_under_abs_f32(_p(w), _p(ref), ctypes.c_int64(w.size),
                                    ctypes.c_double(value),
                                    ctypes.c_float(replacement))
    else:
        ref = _c(ref, np.complex128)
        w = _c(w, np.float64)
        L.orc_replace_under_abs_f64(_p(w), _p(ref), ctypes.c_int64(w.size),
                                    ctypes.c_double(value),
                                    ctypes.c_double(replacement))
    return w


def buffer(x, seg_len, n_overlap, modulated=False):
    """STFT framing -> (seg_len, n_segs) (utils/stft_utils.py:20-98)."""
    L = lib()
    hop = seg_len - n_overlap
    n_segs = (len(x) - seg_len) // hop + 1
    if x.dtype == np.float32:
        x = _c(x, np.float32)
        out = np.empty((seg_len, n_segs), np.float32)
        fn = L.orc_buffer_f32
    else:
        x = _c(x, np.float64)
        out = np.empty((seg_len, n_segs), np.float64)
        fn = L.orc_buffer_f64
    fn(_p(x), _p(out), ctypes.c_int64(len(x)), ctypes.c_int64(seg_len),
       ctypes.c_int64(n_overlap), ctypes.c_int(bool(modulated)))
    return out


# ------------------------------------------------------------- transforms
def reflect_pad(x, n1, n2, padtype='reflect'):
    mode = {'zero': 'constant', 'reflect': 'reflect', 'replicate': 'edge',
            'wrap': 'wrap', 'symmetric': 'symmetric'}[padtype]
    width = (n1, n2) if x.ndim == 1 else [(0, 0), (n1, n2)]
    return np.pad(x, width, mode=mode)


def cwt(x, Psih, xi, dt, n1, N, derivative=True, padtype='reflect',
        workers=None):
    """`Wx[, dWx]` of 1-D/2-D `x` given the dense bank `Psih (na, M)` (already
    Nyquist-halved, in the transform dtype) and `xi (M,)` in the same dtype.
    Follows ssqueezepy/_cwt.py:255-306 with `vectorized=True`; FFTs via
    scipy.fft (pocketfft) as the reference does (utils/fft_utils.py:156-208)."""
    rdt = Psih.dtype
    x = np.asarray(x).astype(rdt)
    M = Psih.shape[-1]
    n2 = M - N - n1
    xp = reflect_pad(x, n1, n2, padtype) if padtype is not None else x
    xh = sfft.fft(xp, axis=-1, workers=workers)
    if x.ndim == 2:
        xh = xh[:, None]
    prod = Psih * xh
    Wx = sfft.ifft(prod, axis=-1, workers=workers)
    dWx = None
    if derivative:
        prod *= (1j * xi / dt)
        dWx = sfft.ifft(prod, axis=-1, workers=workers)
        dWx = np.ascontiguousarray(dWx[..., n1:n1 + N])
    Wx = np.ascontiguousarray(Wx[..., n1:n1 + N])
    return Wx, dWx


def stft(x, window, diff_window, n_fft, hop_len, fs=1., modulated=True,
         derivative=True, padtype='reflect', workers=None):
    """`Sx[, dSx]` (n_fft//2+1, n_hops) of 1-D `x`; `window`, `diff_window` are
    the length-n_fft arrays from `get_window` in the transform dtype
    (ssqueezepy/_stft.py:127-147,160-170)."""
    rdt = window.dtype
    x = np.asarray(x).astype(rdt)
    N = x.shape[-1]
    padlength = N + n_fft - 1
    tot = padlength - N
    n2 = tot // 2
    n1 = n2 if tot % 2 == 0 else n2 + 1
    xp = reflect_pad(x, n1, n2, padtype)
    outs = []
    wins = [window] + ([diff_window] if derivative else [])
    for idx, win in enumerate(wins):
        frames = buffer(xp, n_fft, n_fft - hop_len, modulated)
        if modulated:
            win = sfft.ifftshift(win)
            if idx == 1:
                win = win * fs
        frames *= win.reshape(-1, 1)
        outs.append(sfft.rfft(frames, axis=0, workers=workers))
    return (outs[0], outs[1]) if derivative else (outs[0], None)


# --------------------------------------------------------- ridge extraction
def ridge_design(Tf_dtype, scales, penalty, transform='cwt'):
    """(dtype, eps, penalty matrix) of extract_ridges (ssqueezepy/ridge_extraction.py:
    78-89, 113-128): float64 only for complex128 input, scales log'd for 'cwt',
    `penalty * subtract.outer(scales, scales)**2` in that dtype."""
    c128 = np.dtype(Tf_dtype) == np.complex128
    dtype = np.float64 if c128 else np.float32
    eps = np.asarray(np.finfo(dtype).eps, dtype=dtype)
    scales = np.asarray(scales, dtype=dtype)
    penalty = np.asarray(penalty, dtype=dtype)
    sc = (np.log(scales) if transform == 'cwt' else scales).squeeze()
    P = (penalty * np.subtract.outer(sc, sc)**2).squeeze()
    return dtype, eps, P


def ridge_track(E, P, eps):
    """Forward-backward tracking (ridge_extraction.py:91-111, 143-232) on the
    negative-log energy `E (na, n)`: returns (ridge indices int64 (n,), penalised
    energy (na, n))."""
    E = np.ascontiguousarray(E)
    f32 = E.dtype == np.float32
    T = np.float32 if f32 else np.float64
    Pm = _c(P, T)                      # float32 -> float64 is exact
    na, n = E.shape
    pe = E.copy()
    L = lib()
    fw = L.orc_ridge_fw_f32 if f32 else L.orc_ridge_fw_f64
    bw = L.orc_ridge_bw_f32 if f32 else L.orc_ridge_bw_f64
    fw(_p(pe), _p(Pm), ctypes.c_int64(na), ctypes.c_int64(n))
    # np.unravel_index(np.argmin(pe, axis=0), pe.shape)[1]  (:157-158)
    ridge = np.unravel_index(np.argmin(pe, axis=0), pe.shape)[1].astype(np.int64)
    ridge = np.ascontiguousarray(ridge)
    bw(_p(E), _p(Pm), _p(pe), _p(ridge), ctypes.c_double(float(eps)),
       ctypes.c_int64(na), ctypes.c_int64(n))
    return ridge, pe


def extract_ridges(Tf, scales, penalty=2., n_ridges=1, bw=15, transform='cwt',
                   get_params=False, neglog=None):
    """ridge_extraction.py:11-141. `neglog(energy, energy_max, eps)` optionally replaces
    the NumPy expression `-log(energy / energy_max + eps)` (tests feed the device's
    values through it to compare the tracking bit for bit)."""
    Tf = np.asarray(Tf)
    dtype, eps, P = ridge_design(Tf.dtype, scales, penalty, transform)
    scales_orig = np.asarray(scales, dtype=dtype).copy()
    energy = np.abs(Tf)**2
    n = Tf.shape[1]
    ridge_idxs = np.zeros((n, n_ridges), dtype=int)
    ridge_f = np.zeros((n, n_ridges), dtype=dtype)
    ridge_e = np.zeros((n, n_ridges), dtype=dtype)
    for i in range(n_ridges):
        energy_max = energy.max(axis=0)
        if neglog is None:
            E = -np.log(energy / energy_max + eps)
        else:
            E = neglog(energy, energy_max, eps)
        ridge_idxs[:, i], _ = ridge_track(E, P, eps)
        ridge_f[:, i] = scales_orig.squeeze()[ridge_idxs[:, i]]
        ridge_e[:, i] = energy[ridge_idxs[:, i], range(n)]
        for t in range(n):
            r = ridge_idxs[t, i]
            energy[int(r - bw):int(r + bw), t] = 0
    return (ridge_idxs, ridge_f, ridge_e) if get_params else ridge_idxs
