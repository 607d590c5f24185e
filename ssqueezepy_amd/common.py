# -*- coding: utf-8 -*-
"""`trigdiff` of ssqueezepy/utils/common.py:161-245 on the MI355X: frequency-domain
differentiation of the rows of a CWT-like array (`ssq_trigdiff`, include/ssq_hip.h)."""
import numpy as np
import torch

from . import _lib, algos
from ._lib import check
from .padding import p2up, pad_geometry
from .wavelets import xi_grid

__all__ = ['trigdiff']


def trigdiff(A, fs=1., padtype=None, rpadded=None, N=None, n1=None, window=None,
             transform='cwt'):
    """Trigonometric / frequency-domain differentiation along the last axis:
    ``ifft(fft(A) * 1j * xi * fs)``, un-padded (reference: utils/common.py:161-245).

    # Arguments
        A: (na, n) or (B, na, n) complex torch.Tensor / np.ndarray.
        fs: sampling frequency (scales the derivative to physical units).
        padtype: pad `A` along time before differentiating ('reflect' by default
            when `A` is not already padded).
        rpadded: `A` is already padded (then `N` is required); the result is trimmed
            as ``[..., n1:n1+N]``.
        transform: 'cwt' ('stft' is not supported, as in the reference).

    # Returns
        A_diff: torch.Tensor on the GPU, shape of the un-padded `A`.
    """
    if transform == 'stft':
        raise NotImplementedError("`transform='stft'` is currently not supported.")
    if not isinstance(A, (np.ndarray, torch.Tensor)):
        raise TypeError("`A` must be np.ndarray or torch.Tensor (got %s)" % type(A))
    if A.ndim not in (2, 3):
        raise ValueError("`A` must be 2D or 3D (got %dD)" % A.ndim)
    if rpadded and N is None:
        raise ValueError("must pass `N` if `rpadded`")
    rpadded = rpadded or False
    padtype = padtype or ('reflect' if not rpadded else None)

    A = algos.to_device(A)
    if A.dtype not in (torch.complex64, torch.complex128):
        A = A.to(torch.complex128 if A.dtype == torch.float64 else torch.complex64)
    cdt = A.dtype
    rdt = torch.float32 if cdt == torch.complex64 else torch.float64
    lead = A.shape[:-1]
    rows = int(np.prod(lead))
    A2 = A.reshape(rows, A.shape[-1])

    if padtype is not None:
        if A.ndim == 3:                   # the reference's padsignal takes 1D / 2D only
            raise ValueError("`x` must be 1D or 2D (got x.ndim == 3)")
        n_up, n1p, n2p = pad_geometry(A2.shape[-1])
        if cdt == torch.complex64:        # 8-byte elements: the pad kernel only moves them
            Ap = algos.pad_signal_gpu(A2.contiguous().view(torch.float64), n1p, n2p,
                                      padtype).view(cdt)
        else:
            ri = torch.view_as_real(A2).permute(2, 0, 1).contiguous()        # (2, rows, n)
            pr = algos.pad_signal_gpu(ri.reshape(-1, ri.shape[-1]), n1p, n2p, padtype)
            pr = pr.reshape(2, rows, -1)
            Ap = torch.complex(pr[0], pr[1])
        Ap = Ap.reshape(rows, n_up).contiguous()
        n1 = n1p                          # padsignal's own offset (utils/common.py:213-214)
    else:
        Ap = A2.clone().contiguous()
    n_up = Ap.shape[-1]

    if rpadded or padtype is not None:
        if N is None:
            N = n_up      # as the reference: its `A` is the padded array by now (:236-237)
        if n1 is None:
            _, n1, _ = p2up(N)
        off = min(int(n1), n_up)
        n_out = max(min(off + int(N), n_up) - off, 0)       # NumPy slice clipping
        if n_out == 0:
            return torch.empty(lead + (0,), dtype=cdt, device=Ap.device)
    else:
        n_out, off = n_up, 0

    xi = algos.to_device(np.ascontiguousarray(
        xi_grid(n_up, np.float32 if rdt == torch.float32 else np.float64)), rdt)
    out = torch.empty((rows, n_out), dtype=cdt, device=Ap.device)
    lib = _lib.load()
    check(lib.ssq_trigdiff(_lib.F32 if rdt == torch.float32 else _lib.F64, Ap.data_ptr(),
                           xi.data_ptr(), float(fs), out.data_ptr(), rows, int(n_up), off,
                           n_out, algos.stream()))
    return out.reshape(*lead, n_out)
