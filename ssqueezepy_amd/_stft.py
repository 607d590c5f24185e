# -*- coding: utf-8 -*-
"""Short-Time Fourier Transform on the MI355X.

`stft` / `get_window` keep the signatures and semantics of the reference
(ssqueezepy/_stft.py:13-181, 259-335). The window design (DPSS by default, its
frequency-domain derivative, denormal flushing, NOLA warnings) is host NumPy /
SciPy as in the reference; framing, windowing and the real FFT run on the device
through a cached plan (`ssq_stft_*` in include/ssq_hip.h).
"""
import ctypes
import logging
import numpy as np
import scipy.signal as sig
import torch

from . import _lib
from ._lib import check, F32, F64, StftDesc
from . import algos
from .configs import defaults
from .padding import PADTYPES
from .scales import _process_fs_and_t
from .wavelets import xi_grid

WARN = lambda msg: logging.warning("WARNING: %s" % msg)

__all__ = ['stft', 'get_window', 'StftPlan', 'get_stft_plan']

_TDT = {'float32': torch.float32, 'float64': torch.float64}
_CDT = {'float32': torch.complex64, 'float64': torch.complex128}


def _flush_denormals(x):
    """Zero entries with ``|x| < 1000 * tiny(dtype)`` in place (reference
    `zero_denormals`, ssqueezepy/algos.py:593-613)."""
    tiny = 1000 * np.finfo(x.dtype).tiny
    x[(x < tiny) & (x > -tiny)] = 0
    return x


def _centered(win, win_len, n_fft):
    """`win` zero-padded by the `n_fft - win_len` samples a window of the nominal length lacks,
    the shorter half on the left (an array of another length keeps that padding: it was
    reported, not refused)."""
    spare = n_fft - win_len
    left = spare // 2
    return np.pad(win, [left, spare - left]) if len(win) < n_fft else win


def _spectral_derivative(win):
    """d/dt of a sampled window by multiplication with 1j * xi in the DFT domain; the
    Nyquist bin of an even length carries no derivative."""
    import scipy.fft as sfft
    n = len(win)
    xi = xi_grid(n)
    if n % 2 == 0:
        xi[n // 2] = 0
    return sfft.ifft(sfft.fft(win) * (1j * xi)).real


def get_window(window, win_len, n_fft=None, derivative=False, dtype=None):
    """The analysis window at length `n_fft` (default: `win_len`), optionally with its time
    derivative. `window`: None -> DPSS with time-bandwidth product max(4, win_len // 8); a
    name -> `scipy.signal.get_window(name, win_len, fftbins=True)`; an array -> taken as is
    (a length other than `win_len` is reported, not refused). Values, padding side,
    derivative and denormal flushing as the reference's (ssqueezepy/_stft.py:259-310);
    pinned by tests/test_design_vs_golden.py::test_windows."""
    if n_fft is not None and win_len > n_fft:
        raise ValueError("Can't have `win_len > n_fft` ({} > {})".format(win_len, n_fft))
    if window is None:
        win = sig.windows.dpss(win_len, max(4, win_len // 8), sym=False)
    elif isinstance(window, str):
        win = sig.get_window(window, win_len, fftbins=True)
    elif isinstance(window, np.ndarray):
        win = window
        if len(win) != win_len:
            WARN("len(window) != win_len (%s != %s)" % (len(win), win_len))
    else:
        raise ValueError("`window` must be string or np.ndarray (got %s)" % window)
    if n_fft is not None:
        win = _centered(win, win_len, n_fft)
    out_dtype = win.dtype if dtype is None else dtype
    dwin = _spectral_derivative(win) if derivative else None
    win = _flush_denormals(np.asarray(win).astype(out_dtype))
    if not derivative:
        return win
    return win, _flush_denormals(np.asarray(dwin).astype(out_dtype))


def _check_NOLA(window, hop_len, dtype=None, imprecision_strict=False):
    """Warn when the window / hop pair makes `istft` impossible (NOLA violated, or a hop longer
    than the window) or -- float32 only -- inaccurate at the signal's right end (NOLA barely
    met: tolerance 1e-3, 0.15 when `imprecision_strict`). Reference: _stft.py:313-335."""
    n = len(window)
    overlap = n - hop_len
    if hop_len > n:
        WARN("`hop_len > len(window)`; STFT not invertible")
    elif not sig.check_NOLA(window, n, overlap):
        WARN("`window` fails Non-zero Overlap Add (NOLA) criterion; STFT not invertible")
    single = (str(window.dtype) if dtype is None else dtype) == 'float32'
    if single and hop_len <= n:
        if not sig.check_NOLA(window, n, overlap, tol=0.15 if imprecision_strict else 1e-3):
            WARN("Imprecision expected at right-most hop of signal, in inversion. "
                 "Lower `hop_len`, choose wider `window`, or use `dtype='float64'`.")


_WINDOW_CACHE = {}


def _window_design(window, win_len, n_fft, hop_len, dtype):
    """`get_window(derivative=True)` + `_check_NOLA` memoised per configuration: the
    DPSS design and the overlap-add checks cost far more host time than the device
    transform itself (the analogue of the reference's wavelet cache)."""
    wkey = (window.tobytes(), window.dtype.str) if isinstance(window, np.ndarray) else window
    key = (wkey, int(win_len), int(n_fft), int(hop_len), dtype)
    hit = _WINDOW_CACHE.get(key)
    if hit is None:
        msgs = []
        import logging

        class _Collect(logging.Handler):
            def emit(self, record):
                msgs.append(record.getMessage())
        w, dw = get_window(window, win_len, n_fft, derivative=True, dtype=dtype)
        root, h = logging.getLogger(), _Collect()
        root.addHandler(h)
        try:
            _check_NOLA(w, hop_len, dtype)
        finally:
            root.removeHandler(h)
        if len(_WINDOW_CACHE) >= 32:
            _WINDOW_CACHE.pop(next(iter(_WINDOW_CACHE)))
        hit = _WINDOW_CACHE[key] = (w, dw, tuple(msgs))
        return hit[0], hit[1], ()       # first call: _check_NOLA has already warned
    return hit


class StftPlan():
    """Host handle of a device STFT plan for one
    ``(N, n_fft, hop_len, window, padtype, modulated, dtype)`` configuration."""

    def __init__(self, N, n_fft, hop_len, window, diff_window, fs=1.,
                 padtype='reflect', modulated=True, dtype='float32', max_batch=1):
        self.lib = _lib.load()
        algos._require_gpu()
        self.dtype = dtype
        self.N, self.n_fft, self.hop_len = int(N), int(n_fft), int(hop_len)
        self.max_batch = int(max_batch)
        win = np.asarray(window, dtype=dtype)
        dwin = None if diff_window is None else np.asarray(diff_window, dtype=dtype)
        if modulated:
            # _stft.py:132-135: the window follows the frame rotation; the `* fs`
            # on the differentiated window happens only on this branch
            win = np.fft.ifftshift(win)
            if dwin is not None:
                dwin = np.fft.ifftshift(dwin) * fs
        win = np.ascontiguousarray(win, dtype=dtype)
        dwin = None if dwin is None else np.ascontiguousarray(dwin, dtype=dtype)
        desc = StftDesc()
        desc.dtype = F32 if dtype == 'float32' else F64
        desc.padtype = _lib.PAD[padtype]
        desc.n, desc.n_fft, desc.hop_len = self.N, self.n_fft, self.hop_len
        desc.modulated = int(bool(modulated))
        desc.window = win.ctypes.data
        desc.diff_window = dwin.ctypes.data if dwin is not None else None
        desc.max_batch = self.max_batch
        self._h = ctypes.c_void_p()
        check(self.lib.ssq_stft_plan_create(ctypes.byref(self._h), ctypes.byref(desc)))
        rows, hops = ctypes.c_int64(), ctypes.c_int64()
        check(self.lib.ssq_stft_plan_shape(self._h, ctypes.byref(rows),
                                           ctypes.byref(hops)))
        self.rows, self.n_hops = rows.value, hops.value
        self._ssq_key = None

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            try:
                self.lib.ssq_stft_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    @property
    def algo(self):
        """route of the framing + window + FFT stage: 'fused', 'fused-mixed-radix' or 'rocfft'"""
        return self.lib.ssq_stft_plan_algo(self._h).decode()

    def set_ssq(self, Sfs, grid, params, const, flipud, gamma):
        Sfs = np.ascontiguousarray(np.asarray(Sfs, dtype=self.dtype))
        const = np.asarray(const)
        if const.size != self.rows:
            const = np.full(self.rows, float(const))
        const = const.reshape(-1)
        c64 = 0      # the STFT weight is a scalar -> always the data dtype
        const = np.ascontiguousarray(const.astype(self.dtype))
        key = (Sfs.tobytes(), int(grid), tuple(float(v) for v in params),
               const.tobytes(), bool(flipud), float(gamma))
        if key == self._ssq_key:
            return
        check(self.lib.ssq_stft_plan_set_ssq(self._h, Sfs.ctypes.data, int(grid),
                                             _lib.params5(params), const.ctypes.data,
                                             c64, int(bool(flipud)), float(gamma)))
        self._ssq_key = key

    def execute(self, x, want_dSx=False, want_Tx=False, want_w=False):
        batched = (x.ndim == 2)
        B = x.shape[0] if batched else 1
        if B > self.max_batch:
            raise ValueError("batch %d exceeds the plan's max_batch %d"
                             % (B, self.max_batch))
        shape = ((B, self.rows, self.n_hops) if batched else
                 (self.rows, self.n_hops))
        cdt, rdt = _CDT[self.dtype], _TDT[self.dtype]
        out = {'Sx': torch.empty(shape, dtype=cdt, device=x.device)}
        if want_dSx:
            out['dSx'] = torch.empty(shape, dtype=cdt, device=x.device)
        if want_Tx:
            out['Tx'] = torch.empty(shape, dtype=cdt, device=x.device)
        if want_w:
            out['w'] = torch.empty(shape, dtype=rdt, device=x.device)
        p = lambda k: out[k].data_ptr() if k in out else None
        check(self.lib.ssq_stft_execute(self._h, x.data_ptr(), B, p('Sx'), p('dSx'),
                                        p('Tx'), p('w'), algos.stream()))
        return out


_PLAN_CACHE = {}


def clear_plan_cache():
    """Drop the cached STFT plans (their device workspace is freed with them)."""
    _PLAN_CACHE.clear()


def get_stft_plan(N, n_fft, hop_len, window, diff_window, fs, padtype, modulated,
                  dtype, batch):
    key = (int(N), int(n_fft), int(hop_len), np.asarray(window).tobytes(),
           None if diff_window is None else np.asarray(diff_window).tobytes(),
           float(fs), padtype, bool(modulated), dtype, torch.cuda.current_device())
    plan = _PLAN_CACHE.get(key)
    if plan is not None and plan.max_batch >= batch:
        return plan
    plan = StftPlan(N, n_fft, hop_len, window, diff_window, fs, padtype, modulated,
                    dtype, batch)
    _PLAN_CACHE.pop(key, None)
    if len(_PLAN_CACHE) >= 8:
        _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
    _PLAN_CACHE[key] = plan
    return plan


def _stft_setup(x, window, n_fft, win_len, hop_len, fs, t, padtype, modulated,
                dtype):
    assert x.ndim in (1, 2)
    if padtype not in PADTYPES:
        raise ValueError("`padtype` must be one of: %s (got %s)"
                         % (', '.join(PADTYPES), padtype))
    N = x.shape[-1]
    _, fs, _ = _process_fs_and_t(fs, t, N)
    n_fft = n_fft or min(N // hop_len, 512)
    if win_len is None:
        win_len = (len(window) if isinstance(window, np.ndarray) else n_fft)
    if dtype is None:
        dtype = defaults('stft')['dtype']
    dtype = str(np.dtype(dtype))
    window, diff_window, nola_msgs = _window_design(window, win_len, n_fft, hop_len,
                                                    dtype)
    for msg in nola_msgs:            # the reference warns on every call
        logging.warning(msg)
    xd = algos.to_device(x, _TDT[dtype])
    B = xd.shape[0] if xd.ndim == 2 else 1
    plan = get_stft_plan(N, n_fft, hop_len, window, diff_window, fs, padtype,
                         modulated, dtype, B)
    return plan, xd, fs, dtype


def stft(x, window=None, n_fft=None, win_len=None, hop_len=1, fs=None, t=None,
         padtype='reflect', modulated=True, derivative=False, dtype=None,
         astensor=True):
    """Short-Time Fourier Transform; arguments follow ``ssqueezepy.stft``
    (ssqueezepy/_stft.py:13-99). Returns `Sx` ``(n_fft//2 + 1, n_hops)`` with
    ``n_hops = (len(x) - 1)//hop_len + 1`` (batched: leading signal dim), plus
    `dSx` if `derivative`. Torch GPU tensors by default (`astensor=True`), NumPy
    arrays otherwise."""
    plan, xd, fs, dtype = _stft_setup(x, window, n_fft, win_len, hop_len, fs, t,
                                      padtype, modulated, dtype)
    out = plan.execute(xd, want_dSx=derivative)
    Sx, dSx = out['Sx'], out.get('dSx')
    if not astensor:
        Sx = Sx.cpu().numpy()
        dSx = dSx.cpu().numpy() if dSx is not None else None
    return (Sx, dSx) if derivative else Sx
