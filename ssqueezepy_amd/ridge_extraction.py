# -*- coding: utf-8 -*-
"""Ridge extraction from time-frequency representations, on the MI355X.

Mirror of ssqueezepy/ridge_extraction.py (`extract_ridges`, :11-141): same arguments,
same outputs, the loop nests replaced by `ssq_ridge_*` (include/ssq_hip.h,
csrc/ssq_ridge.hip) so that a transform produced on the device is tracked without
leaving it. The design values (dtype, eps, log-scales, penalty) are formed on the host
with the reference's NumPy expressions; the forward / backward recurrences reproduce the
reference's loops bit for bit on the same negative-log energy, whose `log` is the
device's (within an ulp of NumPy's).
"""
import numpy as np
import torch

from . import _lib
from ._lib import check, F32, F64
from .algos import to_device, stream, _ptr
from .configs import EPS32, EPS64

__all__ = ['extract_ridges']

# transforms per launch: the batch entry points put the transform index on a grid dimension (blockIdx.y / x),
# which holds at most 65535 workgroups; longer batches are walked in chunks of this size
MAX_BATCH_PER_LAUNCH = 32768


def extract_ridges(Tf, scales, penalty=2., n_ridges=1, bw=15, transform='cwt',
                   get_params=False, parallel=True):
    """Tracks `n_ridges` time-frequency ridges by forward-backward penalised
    tracking (ssqueezepy/ridge_extraction.py:11-141).

    # Arguments
        Tf: (n_scales, n_timeshifts) torch.Tensor / np.ndarray, complex or real; or a batch of
            transforms of one scale vector, (batch, n_scales, n_timeshifts) (an extension: the reference
            takes one transform) -- every output gains the leading batch axis, and each transform's
            result is what the 2D call returns for it.
        scales: frequency scales (CWT: scales, log'd inside; STFT: frequencies).
        penalty, n_ridges, bw, transform ('cwt' | 'stft'), get_params: as in the
            reference.
        parallel: accepted for signature compatibility (the device loops have one
            form; it equals the reference's serial variant).

    # Returns
        ridge_idxs (n_timeshifts, n_ridges) integer [, ridge_f, ridge_e]: torch tensors
        on the device when `Tf` is a tensor, NumPy arrays otherwise.
    """
    if transform not in ('cwt', 'stft'):
        raise ValueError("`transform` must be one of: cwt, stft (got %s)" % transform)
    as_tensor = isinstance(Tf, torch.Tensor)
    if not as_tensor:
        Tf = np.asarray(Tf)
        if Tf.dtype.kind in 'biu':            # np.abs(ints)**2 / max -> float64
            Tf = Tf.astype(np.float64)
        elif Tf.dtype == np.float16:
            Tf = Tf.astype(np.float32)
    if Tf.ndim not in (2, 3):
        raise ValueError("`Tf` must be 2D, (scales, timeshifts), or a batch of such, 3D (got shape %s)"
                         % (tuple(Tf.shape),))
    Tf = to_device(Tf)
    if Tf.dtype not in (torch.complex64, torch.complex128, torch.float32, torch.float64):
        Tf = Tf.to(torch.float64)
    # a batch (3D; not in the reference, whose extract_ridges takes one transform): every step below runs
    # over all transforms in one launch -- a tracking pass is one workgroup's walk over time, so a batch
    # costs about what one transform does
    batched = Tf.ndim == 3
    if batched and Tf.shape[0] > MAX_BATCH_PER_LAUNCH:
        # (the chunks are device tensors whatever the caller passed: their results are joined as tensors and converted
        # once, like the results of a single launch below)
        parts = [extract_ridges(Tf[b0:b0 + MAX_BATCH_PER_LAUNCH], scales, penalty, n_ridges, bw, transform,
                                get_params, parallel)
                 for b0 in range(0, Tf.shape[0], MAX_BATCH_PER_LAUNCH)]
        if get_params:
            out = tuple(torch.cat([p[k] for p in parts]) for k in range(3))
            return out if as_tensor else (out[0].cpu().numpy().astype(int), out[1].cpu().numpy(), out[2].cpu().numpy())
        out = torch.cat(parts)
        return out if as_tensor else out.cpu().numpy().astype(int)
    if not batched:
        Tf = Tf[None]
    Tf = Tf.contiguous()
    B, na, n = Tf.shape

    # ridge_extraction.py:113-121: float64 only for complex128 input
    c128 = Tf.dtype == torch.complex128
    pdt = np.float64 if c128 else np.float32
    eps = EPS64 if c128 else EPS32
    if isinstance(scales, torch.Tensor):
        scales = scales.detach().cpu().numpy()
    scales_orig = np.asarray(scales, dtype=pdt).copy()
    sc = (np.log(scales_orig) if transform == 'cwt' else scales_orig).squeeze()
    sc = np.ascontiguousarray(np.atleast_1d(sc))
    if sc.ndim != 1 or len(sc) != na:
        raise ValueError("`scales` must have one entry per row of `Tf` (%s vs %d)"
                         % (sc.shape, na))
    pen = float(np.asarray(penalty, dtype=pdt))

    lib = _lib.load()
    is_cplx = Tf.is_complex()
    rdt = torch.float64 if Tf.dtype in (torch.complex128, torch.float64) else torch.float32
    code = F64 if rdt == torch.float64 else F32
    dev = Tf.device
    sc_d = torch.from_numpy(sc).to(dev)
    energy = torch.empty((B, na, n), dtype=rdt, device=dev)
    E = torch.empty_like(energy)
    pe = torch.empty_like(energy)
    ridge = torch.empty((B, n), dtype=torch.int64, device=dev)
    ridge_idxs = torch.zeros((B, n, n_ridges), dtype=torch.int64, device=dev)
    ridge_e = torch.zeros((B, n, n_ridges), dtype=rdt, device=dev)
    e_col = torch.empty((B, n), dtype=rdt, device=dev)
    st = stream()
    check(lib.ssq_ridge_energy(code, int(is_cplx), _ptr(Tf), _ptr(energy), B * na, n, st))
    for i in range(n_ridges):
        check(lib.ssq_ridge_neglog_batch(code, _ptr(energy), _ptr(E), float(eps), na, n, B, st))
        check(lib.ssq_ridge_track_batch(code, int(not c128), _ptr(E), _ptr(pe), _ptr(sc_d), pen,
                                        float(eps), na, n, _ptr(ridge), B, st))
        check(lib.ssq_ridge_clear_batch(code, _ptr(energy), _ptr(ridge), float(bw), _ptr(e_col),
                                        na, n, B, st))
        ridge_idxs[:, :, i] = ridge
        ridge_e[:, :, i] = e_col
    if not batched:
        ridge_idxs, ridge_e = ridge_idxs[0], ridge_e[0]

    if get_params:
        so = torch.from_numpy(np.ascontiguousarray(scales_orig.reshape(-1))).to(dev)
        ridge_f = so[ridge_idxs]
        ridge_e = ridge_e.to(torch.float64 if c128 else torch.float32)
    if as_tensor:
        return (ridge_idxs, ridge_f, ridge_e) if get_params else ridge_idxs
    ridge_idxs = ridge_idxs.cpu().numpy().astype(int)
    return ((ridge_idxs, ridge_f.cpu().numpy(), ridge_e.cpu().numpy()) if get_params
            else ridge_idxs)
