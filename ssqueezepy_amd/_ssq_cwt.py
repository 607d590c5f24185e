# -*- coding: utf-8 -*-
"""Synchrosqueezed Continuous Wavelet Transform on the MI355X.

`ssq_cwt` keeps the signature, return tuple and error behaviour of the reference's
``ssqueezepy.ssq_cwt`` (ssqueezepy/_ssq_cwt.py:12-310). The reference runs
``cwt(derivative=True)`` and then ``ssqueeze`` as separate passes over full-size
`Wx`, `dWx` (and `w`) arrays; here both stages are one plan execution on the
device (`ssq_cwt_execute`), which writes only the arrays that are returned.
"""
import numpy as np
from types import FunctionType
import torch

from . import algos
from .configs import EPS32, EPS64
from ._cwt import get_cwt_plan, _process_gmw_wavelet, _zero_nonfinite_inplace, _TDT
from .padding import PADTYPES
from .scales import process_scales, infer_scaletype, _process_fs_and_t
from .ssqueezing import (_check_ssqueezing_args, _compute_associated_frequencies, GRID_LIN,
                         ssq_grid_params, ssq_const)
from .wavelets import Wavelet

__all__ = ['ssq_cwt', 'phase_cwt']


_DESIGN_CACHE = {}


def _hashable(v):
    if isinstance(v, np.ndarray):
        return (v.dtype.str, v.shape, v.tobytes())
    if hasattr(v, 'detach'):
        return _hashable(v.detach().cpu().numpy())
    return v


def _ssq_design(wavelet, scales, nv, N, dt, ssq_freqs, maprange, was_padded):
    """Host design step of `ssq_cwt`, memoised per configuration: scale vector,
    frequency axis, reassignment weights and bin-map parameters. Sequenced exactly
    as the reference does it: float64 scale design -> wavelet dtype for the
    transform -> scale type / nv re-inferred from the rounded values for the
    weights (_ssq_cwt.py:243-254, ssqueezing.py:168-188). The reference recomputes
    all of this on every call (tens of ms of NumPy at N=160k); the values depend
    only on the configuration, so they are cached next to the plan."""
    key = (wavelet.key(), _hashable(scales), nv, int(N), float(dt),
           _hashable(ssq_freqs), maprange if not isinstance(maprange, list)
           else tuple(maprange), was_padded)
    hit = _DESIGN_CACHE.get(key)
    if hit is not None:
        return hit
    dtype = wavelet.dtype
    scales64, cwt_scaletype, *_ = process_scales(scales, N, wavelet, nv=nv,
                                                 get_params=True)
    scales_dt = np.asarray(scales64, dtype=dtype)
    scales_ssq, cwt_scaletype2, _, nv_ssq = process_scales(scales_dt.squeeze(), N,
                                                           get_params=True)
    if ssq_freqs is None:
        ssq_freqs = cwt_scaletype
    if not isinstance(ssq_freqs, np.ndarray) and not hasattr(ssq_freqs, 'detach'):
        ssq_scaletype = ssq_freqs if isinstance(ssq_freqs, str) else cwt_scaletype2
        if ((maprange == 'maximal' or isinstance(maprange, tuple)) and
                ssq_scaletype == 'log-piecewise'):
            raise ValueError("can't have `ssq_scaletype = log-piecewise` or "
                             "tuple with `maprange = 'maximal'` "
                             "(got %s)" % str(maprange))
        ssq_freqs = _compute_associated_frequencies(
            scales_ssq, N, wavelet, ssq_scaletype, maprange, was_padded, dt, 'cwt')
    else:
        ssq_freqs = np.asarray(ssq_freqs.detach().cpu().numpy()
                               if hasattr(ssq_freqs, 'detach') else ssq_freqs)
        ssq_scaletype, _ = infer_scaletype(ssq_freqs)
    const = ssq_const('cwt', cwt_scaletype2, nv_ssq, scales_ssq, ssq_freqs)
    grid, params = ssq_grid_params(ssq_freqs, ssq_scaletype.startswith('log'))
    out = (scales_dt, ssq_freqs, const, grid, params)
    if len(_DESIGN_CACHE) >= 16:
        _DESIGN_CACHE.pop(next(iter(_DESIGN_CACHE)))
    _DESIGN_CACHE[key] = out
    return out


def ssq_cwt(x, wavelet='gmw', scales='log-piecewise', nv=None, fs=None, t=None,
            ssq_freqs=None, padtype='reflect', squeezing='sum', maprange='peak',
            difftype='trig', difforder=None, gamma=None, vectorized=True,
            preserve_transform=None, astensor=True, order=0, nan_checks=None,
            patience=0, flipud=True, cache_wavelet=None,
            get_w=False, get_dWx=False):
    """Synchrosqueezed CWT (Daubechies, Lu, Wu 2011; Thakur et al. 2013), computed
    on the GPU in the wavelet's dtype.

    Arguments follow ``ssqueezepy.ssq_cwt`` (ssqueezepy/_ssq_cwt.py:18-189).
    Returns ``(Tx, Wx, ssq_freqs, scales[, w][, dWx])``:

        Tx: (na, N) synchrosqueezed CWT (rows = `ssq_freqs`, highest first by
            default, `flipud=True`); (B, na, N) for batched `x`
        Wx: (na, N) CWT of `x`
        ssq_freqs: (na,) float64 NumPy, frequencies of the rows of `Tx`
        scales: (na,) NumPy in the wavelet dtype
        w: phase transform (if `get_w`);  dWx: time derivative (if `get_dWx`)

    `Tx`, `Wx`, `w`, `dWx` are torch GPU tensors (`astensor=True`) or NumPy arrays.
    Supported on the device path: `difftype='trig'` (other values raise, as in the
    reference's GPU mode); every `squeezing` mode; `order` > 0 / tuple (higher-order GMWs).
    """
    if x.ndim == 2 and get_w:
        raise NotImplementedError("`get_w=True` unsupported with batched input.")
    if not hasattr(x, 'ndim'):
        raise TypeError("`x` must be a numpy array or torch Tensor "
                        "(got %s)" % type(x))
    elif x.ndim not in (1, 2):
        raise ValueError("`x` must be 1D or 2D (got x.ndim == %s)" % x.ndim)
    difforder = _check_ssqueezing_args(squeezing, maprange, wavelet, difftype,
                                       difforder, get_w, transform='cwt')
    if padtype is not None and padtype not in PADTYPES:
        raise ValueError("`padtype` must be one of: %s (got %s)"
                         % (', '.join(PADTYPES), padtype))
    if nv is None and not isinstance(scales, np.ndarray):
        nv = 32
    N = x.shape[-1]
    dt, fs, t = _process_fs_and_t(fs, t, N)
    if nan_checks is None:
        nan_checks = bool(isinstance(x, np.ndarray))
    if nan_checks:
        if not isinstance(x, np.ndarray):
            raise ValueError("`nan_checks=True` requires NumPy input.")
        _zero_nonfinite_inplace(x)

    wavelet = _process_gmw_wavelet(wavelet, True)
    wavelet = Wavelet._init_if_not_isinstance(wavelet, N=N)
    dtype = wavelet.dtype
    if gamma is None:
        gamma = 10 * (EPS64 if dtype == 'float64' else EPS32)
    was_padded = bool(padtype is not None)

    design = _ssq_design(wavelet, scales, nv, N, dt, ssq_freqs, maprange,
                         was_padded)
    scales_dt, ssq_freqs, const, grid, params = design

    use_cache = True if cache_wavelet is None else bool(cache_wavelet)
    xd = algos.to_device(x, _TDT[dtype])
    B = xd.shape[0] if xd.ndim == 2 else 1
    hi_order = isinstance(order, (tuple, list, range)) or order > 0
    if not hi_order:
        plan = get_cwt_plan(wavelet, scales_dt, N, padtype, dt, True, B, cache=use_cache)
        plan.set_ssq(grid, params, const, flipud, gamma)
    if hi_order:
        # higher-order GMWs: the (averaged) transform and its derivative, then the
        # reassignment as its own launch (_ssq_cwt.py:227-241; the reference
        # differentiates the averaged padded transform, which is the average of the
        # derivatives)
        from ._cwt import cwt as _cwt_fn
        Wx, _, dWx_h = _cwt_fn(xd, wavelet, scales=scales_dt, fs=fs, nv=nv, l1_norm=True,
                               derivative=True, padtype=padtype, astensor=True,
                               cache_wavelet=cache_wavelet, nan_checks=False, order=order,
                               average=isinstance(order, (tuple, list, range)))
        out = {'Wx': Wx, 'dWx': dWx_h}
        w = algos.phase_cwt_gpu(Wx, dWx_h, gamma) if get_w else None
        Wq = Wx
        if isinstance(squeezing, FunctionType):
            Wq = squeezing(Wx)
        elif squeezing == 'lebesgue':
            Wq = algos.ones_like(Wx) / len(Wx)
        elif squeezing == 'abs':
            Wq = algos.cabs(Wx)
        logscale = grid != GRID_LIN
        if get_w:
            Tx = algos.indexed_sum_onfly(Wq, w, ssq_freqs, const, logscale, flipud)
        else:
            Tx = algos.ssqueeze_fast(Wq, dWx_h, ssq_freqs, const, logscale, flipud, gamma)
        dWx = dWx_h if get_dWx else None
    elif squeezing == 'sum':
        out = plan.execute(xd, want_dWx=get_dWx, want_Tx=True, want_w=get_w)
        Tx, Wx, w, dWx = out['Tx'], out['Wx'], out.get('w'), out.get('dWx')
    else:
        # 'lebesgue' / 'abs' / callable replace the summed quantity (and, as in the
        # reference, the `Wx` the fused phase transform sees: ssqueezing.py:197-202);
        # the transform stays fused, the reassignment runs as its own launch
        out = plan.execute(xd, want_dWx=True, want_Tx=False, want_w=get_w)
        Wx, w = out['Wx'], out.get('w')
        if isinstance(squeezing, FunctionType):
            Wq = squeezing(Wx)
        elif squeezing == 'lebesgue':
            Wq = algos.ones_like(Wx) / len(Wx)
        else:
            Wq = algos.cabs(Wx)
        logscale = grid != GRID_LIN
        if get_w:
            Tx = algos.indexed_sum_onfly(Wq, w, ssq_freqs, const, logscale, flipud)
        else:
            Tx = algos.ssqueeze_fast(Wq, out['dWx'], ssq_freqs, const, logscale,
                                     flipud, gamma)
        dWx = out['dWx'] if get_dWx else None

    # `scales` go high -> low, so frequencies are returned high -> low
    ssq_freqs = ssq_freqs[::-1]
    scales_out = scales_dt.squeeze()
    if not astensor:
        Tx, Wx, w, dWx = [g.cpu().numpy() if g is not None else None
                          for g in (Tx, Wx, w, dWx)]
    if get_w and get_dWx:
        return Tx, Wx, ssq_freqs, scales_out, w, dWx
    elif get_w:
        return Tx, Wx, ssq_freqs, scales_out, w
    elif get_dWx:
        return Tx, Wx, ssq_freqs, scales_out, dWx
    return Tx, Wx, ssq_freqs, scales_out


def phase_cwt(Wx, dWx, difftype='trig', gamma=None, parallel=None):
    """Phase transform ``w[a, b] = Im((1/2pi) * d/db(Wx[a,b]) / Wx[a,b])``;
    ``inf`` where ``|Wx| < gamma``. Reference: ``phase_cwt``,
    ssqueezepy/_ssq_cwt.py:420-509 (only `difftype='trig'` exists on the GPU
    there too; `gamma` defaults to sqrt(eps) when called standalone)."""
    if difftype != 'trig':
        raise ValueError("`difftype != 'trig'` unsupported with tensor inputs.")
    Wx = algos.to_device(Wx)
    if gamma is None:
        gamma = np.sqrt(EPS64 if Wx.dtype == torch.complex128 else EPS32)
    return algos.phase_cwt_gpu(Wx, dWx, gamma)
