# -*- coding: utf-8 -*-
"""Inverse transforms on the MI355X: `icwt`, `issq_cwt`, `istft`, `issq_stft`.

Same signatures and return values as the reference (ssqueezepy/_cwt.py:323-497,
_ssq_cwt.py:313-417, _stft.py:184-256, _ssq_stft.py:139-198). The O(na * N) work --
reductions over the scale / frequency axis, the inverse real FFTs and the overlap-add --
runs on the device through the C ABI (`ssq_colsum`, `ssq_band_colsum`, `ssq_istft`,
include/ssq_hip.h); admissibility constants and windows are host design values.
A NumPy input gives a NumPy result, a torch GPU tensor gives a torch GPU tensor.
"""
import logging
import numpy as np
import torch

from . import algos
from ._cwt import _process_gmw_wavelet
from ._stft import get_window, _check_NOLA
from .scales import (process_scales, logscale_transition_idx, adm_ssq, adm_cwt)
from .wavelets import Wavelet

WARN = lambda msg: logging.warning("WARNING: %s" % msg)

__all__ = ['icwt', 'issq_cwt', 'istft', 'issq_stft']


_WIN_CACHE = {}


def _window_and_checks(window, win_len, n_fft, hop_len, dtype):
    """`get_window` + `_check_NOLA`, memoised per configuration (the checks cost more
    host time than the device work); warnings are replayed on every call."""
    wkey = ((window.tobytes(), window.dtype.str) if isinstance(window, np.ndarray)
            else window)
    key = (wkey, int(win_len), int(n_fft), int(hop_len), dtype)
    hit = _WIN_CACHE.get(key)
    if hit is None:
        msgs = []

        class _Collect(logging.Handler):
            def emit(self, record):
                msgs.append(record.getMessage())
        w = get_window(window, win_len, n_fft=n_fft, dtype=dtype)
        root, h = logging.getLogger(), _Collect()
        root.addHandler(h)
        try:
            _check_NOLA(w, hop_len, dtype=dtype)
        finally:
            root.removeHandler(h)
        if len(_WIN_CACHE) >= 32:
            _WIN_CACHE.pop(next(iter(_WIN_CACHE)))
        _WIN_CACHE[key] = (w, tuple(msgs))
        return w
    for msg in hit[1]:
        logging.warning(msg)
    return hit[0]


def _is_tensor(x):
    return isinstance(x, torch.Tensor)


def _finish(x, like):
    return x if _is_tensor(like) else x.cpu().numpy()


def _scale(x, c):
    """``x *= c`` with NumPy's rules for an in-place product: a NumPy float64 scalar is
    not a weak type, so a float32 array is multiplied in float64 and rounded back; a
    Python scalar multiplies in the array's precision."""
    if x.dtype == torch.float32 and isinstance(c, np.generic):
        return (x.double() * float(c)).float()
    return x * float(c)


def _add(x, c):
    if isinstance(c, (int, float)) and c == 0:
        return x
    if x.dtype == torch.float32 and isinstance(c, np.generic):
        return (x.double() + float(c)).float()
    return x + float(c)


def _icwt_norm(scaletype, l1_norm):
    # `norm` and `pn` of help(cwt): _cwt.py:479-489
    if l1_norm:
        return None if scaletype == 'log' else (lambda s: s)
    if scaletype == 'log':
        return lambda s: s**.5
    return lambda s: s**1.5


def icwt(Wx, wavelet='gmw', scales='log-piecewise', nv=None, one_int=True, x_len=None,
         x_mean=0, padtype='reflect', rpadded=False, l1_norm=True):
    """Inverse CWT. `one_int=True` (analytic wavelets): ``x = (2 / C_psi) * ln(2**(1/nv)) *
    sum_a Re(Wx[a]) / norm(a) + x_mean``; `one_int=False`: every row is filtered again
    with its wavelet before the sum (double integral, float64 result, single `Wx` only).
    `scales`, `nv`, `l1_norm`, `wavelet`, `padtype` as in the forward call. `Wx`: (na, N) or
    batched (B, na, N)."""
    if not one_int and Wx.ndim == 3:
        raise NotImplementedError("batched `Wx` requires `one_int=True`.")
    *_, na, n = Wx.shape
    x_len = x_len or n
    if not (isinstance(scales, np.ndarray) or _is_tensor(scales)) and nv is None:
        nv = 32
    wavelet = _process_gmw_wavelet(wavelet, l1_norm)
    wavelet = Wavelet._init_if_not_isinstance(wavelet)
    if _is_tensor(scales):
        scales = scales.detach().cpu().numpy()
    scales, scaletype, _, nv = process_scales(scales, x_len, wavelet, nv=nv,
                                              get_params=True)
    assert len(scales) == na, "%s != %s" % (len(scales), na)

    if scaletype == 'log-piecewise':
        # each piece is inverted with its own `nv` (inferred from its scales); the
        # reference adds `x_mean` in both calls (_cwt.py:424-432)
        kw = dict(wavelet=wavelet, one_int=one_int, x_len=x_len, x_mean=x_mean,
                  padtype=padtype, rpadded=rpadded, l1_norm=l1_norm)
        idx = logscale_transition_idx(scales)
        Wd = algos.to_device(Wx)
        x = (icwt(Wd[..., :idx, :], scales=scales[:idx], **kw) +
             icwt(Wd[..., idx:, :], scales=scales[idx:], **kw))
        return _finish(x, Wx)

    Wd = algos.to_device(Wx)
    norm = _icwt_norm(scaletype, l1_norm)
    if not one_int:
        x = _icwt_2int(Wd, np.asarray(scales).reshape(-1), norm, wavelet, x_len, padtype,
                       rpadded)
        Cpsi = adm_cwt(wavelet)
        if scaletype == 'log':
            x = x * float((2 / Cpsi) * np.log(2 ** (1 / nv)))
        else:
            x = x * float((2 / Cpsi) * np.pi / 4)
        x = x + float(x_mean)
        return _finish(x, Wx)
    divisor = None
    if norm is not None:
        divisor = np.asarray(norm(np.asarray(scales).reshape(-1)))
        if divisor.dtype == np.float64 and Wd.dtype == torch.complex64:
            Wd = Wd.to(torch.complex128)      # NumPy promotes `Wx.real / norm` to float64
    x = algos.colsum_real(Wd, divisor)

    Cpsi = adm_ssq(wavelet) if one_int else adm_cwt(wavelet)
    if scaletype == 'log':
        x = _scale(x, (2 / Cpsi) * np.log(2 ** (1 / nv)))
    else:
        x = _scale(x, (2 / Cpsi) * np.pi / 4)
    x = _add(x, x_mean)                       # the CWT does not capture the mean
    return _finish(x, Wx)


def _icwt_2int(Wd, scales, norm, wavelet, x_len, padtype, rpadded):
    """Double-integral iCWT (_cwt.py:448-469): every row is filtered again with its
    wavelet and the rows are summed. The reference's (-1)^k factor and ifftshift cancel
    (even padded length) and the sum over scales commutes with the inverse FFT, so the
    device does one forward FFT per row, a multiply-accumulate over rows and one inverse
    FFT (`ssq_icwt2`). Returns float64, as the reference does."""
    from .padding import pad_geometry
    from . import _lib
    from ._lib import check
    cdt = Wd.dtype
    rdt = torch.float32 if cdt == torch.complex64 else torch.float64
    if not rpadded:
        n_up, n1, n2 = pad_geometry(Wd.shape[-1])
        if cdt == torch.complex64:        # 8-byte elements: the pad kernel only moves them
            Wp = algos.pad_signal_gpu(Wd.view(torch.float64), n1, n2, padtype).view(cdt)
        else:
            ri = torch.view_as_real(Wd).permute(2, 0, 1).contiguous()      # (2, na, N)
            pr = algos.pad_signal_gpu(ri.reshape(-1, ri.shape[-1]), n1, n2, padtype)
            pr = pr.reshape(2, Wd.shape[0], -1)
            Wp = torch.complex(pr[0], pr[1])
    else:
        n_up = Wd.shape[-1]
        n1 = (n_up - x_len) // 2
        Wp = Wd.clone()
    Wp = Wp.contiguous()
    na = Wp.shape[0]
    psih = np.asarray(wavelet(scale=scales.reshape(-1, 1), N=n_up))
    if norm is not None:
        psih = psih / np.asarray(norm(scales)).reshape(-1, 1)
    psih_d = algos.to_device(np.ascontiguousarray(psih), rdt)
    out = torch.empty(n_up, dtype=rdt, device=Wp.device)
    lib = _lib.load()
    check(lib.ssq_icwt2(_lib.F32 if rdt == torch.float32 else _lib.F64, Wp.data_ptr(),
                        psih_d.data_ptr(), out.data_ptr(), int(na), int(n_up),
                        algos.stream()))
    return out[n1:n1 + x_len].double()


# ------------------------------------------------------------ component inversion
def _process_component_inversion_args(cc, cw):
    if cc is None and cw is None:
        return None, None, True
    cc, cw = np.asarray(cc), np.asarray(cw)
    if cc.ndim == 1:
        cc = cc.reshape(-1, 1)
    if cw.ndim == 1:
        cw = cw.reshape(-1, 1)
    return cc.astype('int32'), cw.astype('int32'), False


def _invert_components(Td, cc, cw):
    """Sums of Re(Tx) inside the curve bands ``cc +- cw`` (one per component) and of
    the remainder (_ssq_cwt.py:381-403): `cc == -1` marks "no curve at this time"."""
    na = Td.shape[0]
    upper = np.clip(cc + cw, 0, na)
    lower = np.clip(cc - cw, 0, na)
    upper[cc == -1] = 0
    lower[cc == -1] = 1
    # slice(lower, upper + 1): rows lower .. min(upper, na - 1)
    return algos.band_colsum(Td, lower.T, np.minimum(upper, na - 1).T)


def issq_cwt(Tx, wavelet='gmw', cc=None, cw=None):
    """Inverse synchrosqueezed CWT: ``x = (2 / C_ssq) * sum_k Re(Tx[k])`` or, with curve
    centres `cc` and half-widths `cw` (N x K), the K components and the residual
    ((K + 1) x N, float64)."""
    cc, cw, full_inverse = _process_component_inversion_args(cc, cw)
    Td = algos.to_device(Tx)
    x = algos.colsum_real(Td) if full_inverse else _invert_components(Td, cc, cw)
    wavelet = Wavelet._init_if_not_isinstance(wavelet)
    Css = adm_ssq(wavelet)
    # *2 for the analytic wavelet & the real part
    x = _scale(x, 2 / Css)
    return _finish(x, Tx)


# ------------------------------------------------------------------------- STFT
def istft(Sx, window=None, n_fft=None, win_len=None, hop_len=1, N=None, modulated=True,
          win_exp=1):
    """Inverse STFT (Griffin & Lim least-squares estimate for `win_exp=1`):
    ``x[n] = sum_t y_t[n] w^a[n - tH] / sum_t w^(a+1)[n - tH]``, ``y_t = irfft(Sx[:, t])``."""
    n_fft = n_fft or (Sx.shape[0] - 1) * 2
    win_len = win_len or n_fft
    N = N or hop_len * Sx.shape[1]          # longest possible signal if not given
    dtype = 'float32' if str(Sx.dtype).endswith('complex64') else 'float64'
    window = _window_and_checks(window, win_len, n_fft, hop_len, dtype)
    if len(window) != n_fft:
        raise ValueError("Must have `len(window) == n_fft` (got %s != %s)"
                         % (len(window), n_fft))
    if win_exp == 0:
        win_a = np.ones(n_fft, dtype=dtype)
    elif win_exp == 1:
        win_a = window
    else:
        win_a = window ** win_exp
    win_a1 = window ** (win_exp + 1)
    x = algos.istft_gpu(Sx, win_a, win_a1, n_fft, hop_len, N, modulated)
    return _finish(x, Sx)


def issq_stft(Tx, window=None, cc=None, cw=None, n_fft=None, win_len=None, hop_len=1,
              modulated=True):
    """Inverse synchrosqueezed STFT (hop 1, modulated): ``x = (2 / w[n_fft//2]) *
    sum_k Re(Tx[k])``, optionally per curve band as in `issq_cwt`."""
    if not modulated:
        raise ValueError("inversion with `modulated == False` is unsupported.")
    if hop_len != 1:
        raise ValueError("inversion with `hop_len != 1` is unsupported.")
    cc, cw, full_inverse = _process_component_inversion_args(cc, cw)
    n_fft = n_fft or (Tx.shape[0] - 1) * 2
    win_len = win_len or n_fft
    window = _window_and_checks(window, win_len, n_fft, hop_len, None)
    if abs(np.argmax(window) - len(window) // 2) > 1:
        WARN("`window` maximum not centered; results may be inaccurate.")
    Td = algos.to_device(Tx)
    x = algos.colsum_real(Td) if full_inverse else _invert_components(Td, cc, cw)
    x = _scale(x, 2 / window[len(window) // 2])
    return _finish(x, Tx)
