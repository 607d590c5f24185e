# -*- coding: utf-8 -*-
"""Kernel-level seam: the reference's `algos.py` entry points, on the MI355X.

Same names, argument meaning and error behaviour as the functions the reference's
transforms call (ssqueezepy/algos.py): `ssqueeze_fast` (126-150),
`indexed_sum_onfly` (153-169), `phase_cwt_gpu` (743-781), `phase_stft_gpu`
(818-856), `replace_under_abs` (498-579) and `buffer` (utils/stft_utils.py:20-66).
Inputs may be NumPy arrays (uploaded) or torch tensors; outputs are torch tensors
on the GPU. Every function is a thin marshalling layer over one C-ABI call of
libssq_hip.so, launched on torch's current stream -- there is no CPU
implementation behind them.
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, params5, F32, F64

__all__ = ['ssqueeze_fast', 'indexed_sum_onfly', 'phase_cwt_gpu', 'phase_stft_gpu',
           'replace_under_abs', 'buffer', 'pad_signal_gpu', 'to_device']

_CDT = {torch.complex64: F32, torch.complex128: F64,
        torch.float32: F32, torch.float64: F64}


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("ssqueezepy_amd needs a ROCm GPU (torch.cuda.is_available()"
                           " is False); there is no CPU fallback.")


def device():
    _require_gpu()
    return torch.device('cuda', torch.cuda.current_device())


def stream():
    return torch.cuda.current_stream().cuda_stream


def to_device(x, dtype=None):
    """NumPy array / torch tensor -> contiguous torch tensor on the current GPU."""
    dev = device()
    if isinstance(x, np.ndarray):
        if not x.flags.c_contiguous:
            x = np.ascontiguousarray(x)
        x = torch.from_numpy(x)
    elif not isinstance(x, torch.Tensor):
        raise TypeError("expected numpy array or torch Tensor (got %s)" % type(x))
    if dtype is not None and x.dtype != dtype:
        x = x.to(dtype)
    return x.to(dev).contiguous()


def ones_like(x):
    return torch.ones_like(x)


def cabs(x):
    return torch.abs(x).to(x.dtype)


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _real_of(cdtype):
    return torch.float32 if cdtype == torch.complex64 else torch.float64


def _shape3(t):
    if t.ndim == 2:
        return 1, t.shape[0], t.shape[1]
    if t.ndim == 3:
        return t.shape
    raise ValueError("expected a 2D or 3D array (got ndim=%d)" % t.ndim)


def _const_vector(const, na, cdtype):
    """Per-row weights as the reference materialises them (algos.py:66-79):
    scalar -> vector in the data dtype; a float64 vector weighting complex64 data
    stays float64 (the accumulate is then done in double, like NumPy/numba do)."""
    rdt = _real_of(cdtype)
    if isinstance(const, torch.Tensor):
        const = const.detach().cpu().numpy()
    const = np.asarray(const)
    if const.size != na:
        vec = torch.full((na,), float(const), dtype=rdt)
        return vec.to(device()), 0
    const = const.reshape(-1)
    if cdtype == torch.complex64 and const.dtype == np.float64:
        return torch.from_numpy(np.ascontiguousarray(const)).to(device()), 1
    return torch.from_numpy(np.ascontiguousarray(const)).to(rdt).to(device()), 0


def _grid(ssq_freqs, logscale):
    from .ssqueezing import ssq_grid_params
    kind, p = ssq_grid_params(ssq_freqs, logscale)
    return kind, params5(p)


def ssqueeze_fast(Wx, dWx, ssq_freqs, const, logscale=False, flipud=False,
                  gamma=None, out=None, Sfs=None, parallel=None, get_k=False):
    """Fused phase transform + nearest-bin search + accumulate: for every point
    with ``|Wx| > gamma``, ``Tx[k, j] += Wx[i, j] * const[i]``.
    Reference: `ssqueeze_fast`, ssqueezepy/algos.py:126-150."""
    if gamma is None:
        raise ValueError("`gamma` must not be None")
    lib = _lib.load()
    Wx, dWx = to_device(Wx), to_device(dWx)
    if Wx.dtype not in (torch.complex64, torch.complex128):
        raise TypeError("`Wx` must be complex64 or complex128 (got %s)" % Wx.dtype)
    dWx = dWx.to(Wx.dtype)
    if Wx.shape != dWx.shape:
        raise ValueError("`Wx` and `dWx` shapes differ: %s vs %s"
                         % (tuple(Wx.shape), tuple(dWx.shape)))
    B, na, n = _shape3(Wx)
    if out is None:
        out = torch.empty_like(Wx)
    elif out.shape != Wx.shape or out.dtype != Wx.dtype or not out.is_cuda:
        raise ValueError("`out` must be a GPU tensor of `Wx`'s shape and dtype")
    cst, c64 = _const_vector(const, na, Wx.dtype)
    kind, p = _grid(ssq_freqs, logscale)
    sfs = None
    if Sfs is not None:
        sfs = to_device(Sfs, _real_of(Wx.dtype))
    kmap = (torch.empty((B, na, n), dtype=torch.int32, device=Wx.device)
            if get_k else None)
    check(lib.ssq_ssqueeze(_CDT[Wx.dtype], _ptr(Wx), _ptr(dWx), _ptr(sfs),
                           _ptr(out), _ptr(cst), c64, B, na, n, float(gamma), kind,
                           p, int(bool(flipud)), _ptr(kmap), stream()))
    if get_k:
        return out, kmap.reshape(Wx.shape)
    return out


def indexed_sum_onfly(Wx, w, ssq_freqs, const=1, logscale=False, flipud=False,
                      out=None, parallel=None):
    """Nearest-bin search + accumulate from a precomputed phase transform `w`
    (``inf`` = skip). Reference: `indexed_sum_onfly`, ssqueezepy/algos.py:153-169."""
    lib = _lib.load()
    Wx = to_device(Wx)
    w = to_device(w, _real_of(Wx.dtype))
    if Wx.shape != w.shape:
        raise ValueError("`Wx` and `w` shapes differ")
    B, na, n = _shape3(Wx)
    if out is None:
        out = torch.empty_like(Wx)
    cst, c64 = _const_vector(const, na, Wx.dtype)
    kind, p = _grid(ssq_freqs, logscale)
    check(lib.ssq_indexed_sum(_CDT[Wx.dtype], _ptr(Wx), _ptr(w), _ptr(out),
                              _ptr(cst), c64, B, na, n, kind, p,
                              int(bool(flipud)), stream()))
    return out


def phase_cwt_gpu(Wx, dWx, gamma):
    """``w = inf where |Wx| < gamma else |Im(dWx / Wx)| / 2pi``.
    Reference: `phase_cwt_gpu`, ssqueezepy/algos.py:743-781."""
    lib = _lib.load()
    Wx, dWx = to_device(Wx), to_device(dWx)
    dWx = dWx.to(Wx.dtype)
    B, na, n = _shape3(Wx)
    w = torch.empty(Wx.shape, dtype=_real_of(Wx.dtype), device=Wx.device)
    check(lib.ssq_phase_cwt(_CDT[Wx.dtype], _ptr(Wx), _ptr(dWx), _ptr(w), B, na, n,
                            float(gamma), stream()))
    return w


def phase_stft_gpu(Sx, dSx, Sfs, gamma):
    """``w = inf where |Sx| < gamma else |Sfs[i] - Im(dSx / Sx) / 2pi|``.
    Reference: `phase_stft_gpu`, ssqueezepy/algos.py:818-856."""
    lib = _lib.load()
    Sx, dSx = to_device(Sx), to_device(dSx)
    dSx = dSx.to(Sx.dtype)
    sfs = to_device(Sfs, _real_of(Sx.dtype))
    B, na, n = _shape3(Sx)
    w = torch.empty(Sx.shape, dtype=_real_of(Sx.dtype), device=Sx.device)
    check(lib.ssq_phase_stft(_CDT[Sx.dtype], _ptr(Sx), _ptr(dSx), _ptr(sfs), _ptr(w),
                             B, na, n, float(gamma), stream()))
    return w


def replace_under_abs(x, ref=None, value=1., replacement=0., parallel=None):
    """In place: ``x[abs(ref) < value] = replacement`` (`ref` complex, `x` real GPU
    tensors). Reference: `replace_under_abs`, ssqueezepy/algos.py:498-579."""
    lib = _lib.load()
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise TypeError("`x` must be a GPU tensor (modified in place)")
    ref = to_device(ref)
    if ref.dtype not in (torch.complex64, torch.complex128):
        ref = ref.to(torch.complex64 if x.dtype == torch.float32 else
                     torch.complex128)
    if x.dtype != _real_of(ref.dtype) or not x.is_contiguous():
        raise TypeError("`x` must be contiguous and of `ref`'s real dtype")
    check(lib.ssq_replace_under_abs(_CDT[ref.dtype], _ptr(x), _ptr(ref), x.numel(),
                                    float(value), float(replacement), stream()))
    return x


def buffer(x, seg_len, n_overlap, modulated=False, parallel=None):
    """Frames of `x` as columns: ``(seg_len, n_segs)`` or batched
    ``(B, seg_len, n_segs)``. Reference: `buffer`, utils/stft_utils.py:20-66."""
    lib = _lib.load()
    x = to_device(x)
    assert x.ndim in (1, 2)
    if x.dtype not in (torch.float32, torch.float64):
        x = x.to(torch.float32)
    hop = seg_len - n_overlap
    n_x = x.shape[-1]
    n_segs = (n_x - seg_len) // hop + 1
    B = 1 if x.ndim == 1 else x.shape[0]
    out = torch.empty((B, seg_len, n_segs), dtype=x.dtype, device=x.device)
    check(lib.ssq_buffer(_CDT[x.dtype], _ptr(x), _ptr(out), B, n_x, seg_len,
                         n_overlap, int(bool(modulated)), stream()))
    return out[0] if x.ndim == 1 else out


def pad_signal_gpu(x, n1, n2, padtype='reflect'):
    """Device signal extension (the kernel behind `cwt`/`stft` padding)."""
    lib = _lib.load()
    x = to_device(x)
    B = 1 if x.ndim == 1 else x.shape[0]
    n = x.shape[-1]
    out = torch.empty((B, n1 + n + n2), dtype=x.dtype, device=x.device)
    check(lib.ssq_pad_signal(_CDT[x.dtype], _ptr(x), _ptr(out), B, n, n1, n2,
                             _lib.PAD[padtype], stream()))
    return out[0] if x.ndim == 1 else out


# ------------------------------------------------------------------ inverses
def colsum_real(Z, divisor=None):
    """``sum_i Re(Z[..., i, :]) [/ divisor[i]]`` over the scale / frequency axis, rows in
    ascending order in `Z`'s own precision (bit-identical to NumPy's
    ``(Z.real / d).sum(axis=-2)``). `Z`: (na, n) or (B, na, n) complex."""
    lib = _lib.load()
    Z = to_device(Z)
    B, na, n = _shape3(Z)
    rdt = _real_of(Z.dtype)
    d = None if divisor is None else to_device(np.ascontiguousarray(
        np.asarray(divisor).reshape(-1)), rdt)
    if d is not None and d.numel() != na:
        raise ValueError("`divisor` must have one entry per row (%d != %d)"
                         % (d.numel(), na))
    out = torch.empty(Z.shape[:-2] + (n,), dtype=rdt, device=Z.device)
    check(lib.ssq_colsum(_CDT[Z.dtype], _ptr(Z), _ptr(d) if d is not None else None,
                         _ptr(out), B, na, n, stream()))
    return out


def band_colsum(Z, lo, hi):
    """Per-component sums of ``Re(Z)`` over the row bands ``lo[k, j] .. hi[k, j]`` of every
    column (float64), plus the sum of the rows no band covers as the last row.
    `Z`: (na, n) complex; `lo`, `hi`: (K, n) integer arrays, inclusive."""
    lib = _lib.load()
    Z = to_device(Z)
    if Z.ndim != 2:
        raise ValueError("component inversion takes a single (na, n) transform")
    na, n = Z.shape
    lo = torch.as_tensor(np.ascontiguousarray(lo, dtype=np.int32), device=Z.device)
    hi = torch.as_tensor(np.ascontiguousarray(hi, dtype=np.int32), device=Z.device)
    K = lo.shape[0]
    out = torch.empty((K + 1, n), dtype=torch.float64, device=Z.device)
    check(lib.ssq_band_colsum(_CDT[Z.dtype], _ptr(Z), _ptr(lo), _ptr(hi), K, _ptr(out),
                              na, n, stream()))
    return out


def istft_gpu(Sx, win_a, win_a1, n_fft, hop_len, N, modulated=True):
    """irfft of every column of `Sx` + overlap-add with `win_a`, divided by the
    overlap-added `win_a1`, trimmed to `N` samples (`ssq_istft`, include/ssq_hip.h)."""
    lib = _lib.load()
    Sx = to_device(Sx)
    rdt = _real_of(Sx.dtype)
    rows, n_hops = Sx.shape
    wa, wa1 = to_device(win_a, rdt), to_device(win_a1, rdt)
    x = torch.empty(int(N), dtype=rdt, device=Sx.device)
    check(lib.ssq_istft(_CDT[Sx.dtype], _ptr(Sx), _ptr(wa), _ptr(wa1), _ptr(x),
                        int(n_fft), int(n_hops), int(hop_len), int(N),
                        int(bool(modulated)), stream()))
    return x
