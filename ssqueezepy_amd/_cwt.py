# -*- coding: utf-8 -*-
"""Continuous Wavelet Transform on the MI355X.

`cwt` keeps the signature, argument meaning, return values and error behaviour of
the reference's ``ssqueezepy.cwt`` (ssqueezepy/_cwt.py:12-320). What differs is
where the work happens: the reference broadcasts ``Psih * xh`` into dense
``(na, M)`` temporaries and runs two batched iFFTs through torch; here a cached
*plan* (libssq_hip.so, `ssq_cwt_*` in include/ssq_hip.h) owns the banded filter
bank, the FFT plans and the workspace on the device, and `cwt` is a thin host
wrapper: design step (cached) -> one `ssq_cwt_execute` call.
"""
import ctypes
import logging
import os
import numpy as np
import torch

from . import _lib
from ._lib import check, F32, F64, CwtDesc, CwtBlocksDesc, CwtTilesDesc
from . import algos
from ._bank import banded_bank, support_hull
from ._blocks import plan_blocks, L_MIN
from ._tiles import plan_tiles, RSUB as _tiles_rsub, R_MIN as _tiles_rmin
from .padding import pad_geometry, PADTYPES
from .scales import process_scales, _process_fs_and_t
from .wavelets import Wavelet

WARN = lambda msg: logging.warning("WARNING: %s" % msg)

__all__ = ['cwt', 'cwt_higher_order', 'CwtPlan', 'get_cwt_plan', 'clear_plan_cache']

_TDT = {'float32': torch.float32, 'float64': torch.float64}
_CDT = {'float32': torch.complex64, 'float64': torch.complex128}


class CwtPlan():
    """Host handle of a device CWT plan for one
    ``(wavelet, scales, N, padtype, dt, l1_norm)`` configuration."""

    def __init__(self, wavelet, scales, N, padtype='reflect', dt=1., l1_norm=True,
                 max_batch=1, band_tol=None, algo=0):
        self.lib = _lib.load()
        algos._require_gpu()
        self.dtype = wavelet.dtype
        self.N = int(N)
        self.padtype = padtype
        if padtype is not None:
            self.M, self.n1, self.n2 = pad_geometry(self.N)
        else:
            self.M, self.n1, self.n2 = self.N, 0, 0
        rdt = np.dtype(self.dtype)
        self.scales = np.ascontiguousarray(np.asarray(scales, dtype=rdt).reshape(-1))
        self.na = len(self.scales)
        self.max_batch = int(max_batch)
        vals, off, lo = banded_bank(wavelet, self.scales, self.M, tol=band_tol,
                                    nohalf=False)
        self._band_tol = band_tol
        self.bank_nnz = int(off[-1])
        row_scale = None
        if not l1_norm:
            row_scale = np.ascontiguousarray(np.sqrt(self.scales).astype(rdt))
        desc = CwtDesc()
        desc.dtype = F32 if self.dtype == 'float32' else F64
        desc.padtype = _lib.PAD[padtype]
        desc.n, desc.m, desc.n1, desc.na = self.N, self.M, self.n1, self.na
        desc.bank = vals.ctypes.data
        desc.band_off = off.ctypes.data
        desc.band_lo = lo.ctypes.data
        desc.dt = float(dt)
        desc.row_scale = row_scale.ctypes.data if row_scale is not None else None
        desc.max_batch = self.max_batch
        desc.algo = int(algo)
        self._h = ctypes.c_void_p()
        check(self.lib.ssq_cwt_plan_create(ctypes.byref(self._h), ctypes.byref(desc)))
        self._bank = (vals, off, lo, row_scale)     # host copy: dense rows for `backward`
        self._psih_dev = None
        self._pad_src = None
        self._ssq_key = None
        self.block_rows = 0
        self.extended_rows = 0           # Nyquist-cut rows run as block rows (analytic signal)
        self.tile_rows = 0
        self.dt = float(dt)
        if algo == 0 and os.environ.get('SSQ_DEBUG_CWT_ALGO', 'auto') != 'generic':
            self._try_blocks(wavelet, vals, off, lo)

    def _try_blocks(self, wavelet, vals, off, lo):
        """Plan and install the block ("overlap-save zoom") fast path when the
        configuration admits it (padded power-of-two length, analytic
        bank); see _blocks.py. Impulse-response margins are measured on the
        wavelet evaluated in float64 when it is a built-in family."""
        if self.padtype is None or os.environ.get('SSQ_DEBUG_CWT_ALGO') == 'generic':
            return
        vals64 = None
        fn64 = wavelet.fn if self.dtype == 'float64' else None
        if self.dtype == 'float32' and wavelet.family is not None:
            cfg = {k: v for k, v in wavelet.config.items() if k != 'dtype'}
            try:
                twin = Wavelet((wavelet.family, dict(cfg, dtype='float64')))
                v64, o64, l64 = banded_bank(twin, self.scales.astype('float64'),
                                            self.M, tol=1e-3 * np.finfo('float32').eps)
                if np.array_equal(o64, off) and np.array_equal(l64, lo):
                    vals64 = v64
                    fn64 = twin.fn
            except Exception:
                vals64 = None
        # rows cut by the Nyquist bin are continued past it and run as block rows over the
        # analytic signal (_blocks.extend_past_nyquist) when the wavelet can be evaluated in
        # float64 (built-in families); SSQ_DEBUG_CWT_NYQ_EXT=0 keeps them on the exact path
        extension = None
        if fn64 is not None and os.environ.get('SSQ_DEBUG_CWT_NYQ_EXT', '1') != '0':
            tol = self._band_tol if self._band_tol is not None else 1e-3 * np.finfo(self.dtype).eps
            try:
                _, w_hi = support_hull(wavelet.fn, np.dtype(self.dtype), tol,
                                       w_extent=float(self.scales.max()) * np.pi * 1.01 + 1)
                extension = (fn64, self.scales.astype('float64'), float(w_hi))
            except Exception:
                extension = None
        bp = plan_blocks(vals, off, lo, self.M, self.N, self.n1, self.dtype,
                         vals64=vals64, extension=extension)
        if bp is None:
            return
        cls = np.ascontiguousarray(bp['classes'], dtype=np.int64)
        # the single-block class spans the whole padded signal: its "margin" is the
        # left pad, so that block sample t maps to output t - n1 (see kernel)
        cls[cls[:, 0] == self.M, 1] = self.n1
        rows = np.ascontiguousarray(bp['rows'], dtype=np.int32)
        rdt, cdt = (('float32', 'complex64') if self.dtype == 'float32' else
                    ('float64', 'complex128'))
        pbank = np.ascontiguousarray(bp['pbank'], dtype=rdt)
        pxi = np.ascontiguousarray(bp['pxi'], dtype=rdt)
        ctw = np.ascontiguousarray(bp['ctw'], dtype=cdt)
        ctw_off = np.ascontiguousarray(bp['ctw_off'], dtype=np.int64)
        ftw = np.ascontiguousarray(bp['ftw'], dtype=cdt)
        gen = np.ascontiguousarray(bp['generic_rows'], dtype=np.int32)
        # column-tile path of the fused ssq form (_tiles.py): rows it interpolates leave the
        # block kernels when `Tx` is requested -- their items go to the end of each list
        tp = None
        if self.dtype == 'float32' and os.environ.get('SSQ_CWT_TILES', '1') != '0':
            # (SSQ_DEBUG_TILE_RMIN: least decimation for which a row leaves the block kernels; tuning aid)
            # (candidates: block rows whose band lies below Nyquist -- the rows continued past it are
            # described by another band than the one plan_tiles is given, and stay block rows)
            tp = plan_tiles(vals, off, lo, self.M, self.N, self.n1, self.dt,
                            (rows[:, 0] >= 0) & ~np.asarray(bp['extended'], bool),
                            self.group, row_scale=self._bank[3],
                            r_min=int(os.environ.get('SSQ_DEBUG_TILE_RMIN', _tiles_rmin)))
        n_items_tile = [0] * 5
        keep = [cls, rows, pbank, pxi, ctw, ctw_off, ftw, gen]
        d = CwtBlocksDesc()
        d.n_classes = len(cls)
        d.classes, d.rows = cls.ctypes.data, rows.ctypes.data
        d.pbank, d.n_pbank = pbank.ctypes.data, len(pbank)
        d.pxi = pxi.ctypes.data
        d.ctw, d.ctw_off = ctw.ctypes.data, ctw_off.ctypes.data
        d.ftw, d.n_ftw = ftw.ctypes.data, len(ftw)
        for slot in range(5):
            Lp = L_MIN << slot
            it = np.ascontiguousarray(bp['items'].get(Lp, np.zeros((0, 4))),
                                      dtype=np.int32)
            if tp is not None and len(it):
                stay = ~tp['interp_rows'][it[:, 0]]
                it = np.ascontiguousarray(np.concatenate([it[stay], it[~stay]]))
                n_items_tile[slot] = int(stay.sum())
            keep.append(it)
            d.items[slot] = it.ctypes.data if len(it) else None
            d.n_items[slot] = len(it)
            d.ftw_off[slot] = bp['ftw_off'][Lp]
        d.generic_rows = gen.ctypes.data if len(gen) else None
        d.n_generic = len(gen)
        check(self.lib.ssq_cwt_plan_set_blocks(self._h, ctypes.byref(d)))
        self.block_rows = int((rows[:, 0] >= 0).sum())
        self.extended_rows = int(bp['extended'].sum())
        if tp is not None:
            self._set_tiles(tp, n_items_tile, len(gen))
        self.block_plan = {k: bp[k] for k in ('classes', 'margins')}

    def _set_tiles(self, tp, n_items_tile, n_exact):
        c = lambda a, dt: np.ascontiguousarray(a, dtype=dt)
        segs, rws = c(tp['segs'], np.int32), c(tp['rows'], np.int32)
        wtab, tbank = c(tp['wtab'], np.float32), c(tp['tbank'], np.float32)
        irows, classes = c(tp['irows'], np.int64), c(tp['classes'], np.int64)
        if self.lib.ssq_cwt_tile_rows_per_step() != _tiles_rsub:
            raise RuntimeError("tile tables built for %d rows per step, the library walks %d"
                               % (_tiles_rsub, self.lib.ssq_cwt_tile_rows_per_step()))
        d = CwtTilesDesc()
        d.n_segs, d.segs = len(segs), segs.ctypes.data
        d.n_steps, d.rows = len(rws) // _tiles_rsub, rws.ctypes.data
        d.wtab, d.n_phases = wtab.ctypes.data, len(wtab)
        d.tbank, d.n_tbank = tbank.ctypes.data, len(tbank)
        d.n_irows, d.irows = len(irows), irows.ctypes.data
        d.n_classes, d.classes = len(classes), classes.ctypes.data
        d.u_total = tp['u_total']
        d.reserved = _tiles_rsub                 # rows per step of `rows` (the library checks it against its own)
        for slot in range(5):
            d.n_items_tile[slot] = n_items_tile[slot]
        check(self.lib.ssq_cwt_plan_set_tiles(self._h, ctypes.byref(d)))
        self.tile_rows = int(tp['interp_rows'].sum())
        self.tile_plan = {k: tp[k] for k in ('lgR', 'interp_rows', 'classes')}

    def __del__(self):
        h = getattr(self, '_h', None)
        if h is not None and h.value:
            try:
                self.lib.ssq_cwt_plan_destroy(h)
            except Exception:
                pass
            self._h = None

    @property
    def algo(self):
        return self.lib.ssq_cwt_plan_algo(self._h).decode()

    def tiles_done(self):
        """Column tiles (`tile_cols` columns each) the column-tile kernel has finished on this
        plan so far (what actually executed, as opposed to what `algo` says was planned);
        synchronises."""
        return int(self.lib.ssq_cwt_plan_tiles_done(self._h, algos.stream()))

    @property
    def tile_cols(self):
        """columns per tile of the tile kernel the next execute launches (0: no tile path)"""
        return int(self.lib.ssq_cwt_plan_tile_cols(self._h))

    @property
    def tile_kernel(self):
        """which column-tile kernel the next execute launches: 0 none, 1 ordered (float32 tile), 2 float64 tile
        with a column per lane, 3 float64 tile with a column pair per lane (`ssq_cwt_plan_tile_kernel`)"""
        return int(self.lib.ssq_cwt_plan_tile_kernel(self._h))

    def set_bin_dump(self, kmap):
        """Diagnostic (`ssq_cwt_plan_set_bin_dump`): `kmap` -- a GPU int16 / uint16 tensor of at least
        ``max_batch * na * N`` elements, kept alive by the plan -- receives the bin index of every point
        of every following fused execute (0xFFFF: below gamma); ``None`` switches it off."""
        if kmap is not None and (kmap.element_size() != 2 or kmap.numel() < self.max_batch * self.na * self.N
                                 or not kmap.is_contiguous()):
            raise ValueError("bin dump: a contiguous 2-byte tensor of max_batch * na * N elements is needed")
        check(self.lib.ssq_cwt_plan_set_bin_dump(self._h, kmap.data_ptr() if kmap is not None else None))
        self._kdump = kmap

    def tiles_per_signal(self, N):
        if not self.tile_cols:
            return 0
        # (the pair kernel's tiles start a column early when the left padding is odd: csrc/ssq_tile_pair.hip)
        lead = (self.n1 & 1) if self.tile_kernel == 3 else 0
        return -(-(int(N) + lead) // self.tile_cols)

    @property
    def device_bytes(self):
        return int(self.lib.ssq_cwt_plan_bytes(self._h))

    @property
    def group(self):
        """signals per kernel launch (the batch is walked in groups of this size)"""
        return int(self.lib.ssq_cwt_plan_group(self._h))

    def timing(self, enable=-1):
        """Per-stage HIP-event timing (see `ssq_cwt_plan_timing`): returns
        ``(stage_ms[4], n_signals)`` accumulated so far; `enable` 1/0 switches it on/off
        and resets, -1 only reads."""
        ms = (ctypes.c_double * 4)()
        n = ctypes.c_int64()
        check(self.lib.ssq_cwt_plan_timing(self._h, int(enable), ms, ctypes.byref(n)))
        return list(ms), int(n.value)

    def set_ssq(self, grid, params, const, flipud, gamma):
        """Synchrosqueezing parameters for subsequent `execute(..., Tx=...)`."""
        # as the reference materialises it (algos.py:66-79): a scalar becomes a
        # vector in the data dtype; a float64 *vector* with float32 data stays
        # float64 and the accumulate is then done in double
        const = np.asarray(const)
        if const.size != self.na:
            const = np.full(self.na, float(const)).astype(self.dtype)
        const = const.reshape(-1)
        c64 = int(self.dtype == 'float32' and const.dtype == np.float64)
        if not c64:
            const = const.astype(self.dtype)
        const = np.ascontiguousarray(const)
        key = (int(grid), tuple(float(v) for v in params), const.tobytes(),
               bool(flipud), float(gamma))
        if key == self._ssq_key:
            return
        check(self.lib.ssq_cwt_plan_set_ssq(self._h, int(grid), _lib.params5(params),
                                            const.ctypes.data, c64,
                                            int(bool(flipud)), float(gamma)))
        self._ssq_key = key

    def execute(self, x, want_dWx=False, want_Tx=False, want_w=False, rpadded=False):
        """x: GPU tensor (N,) or (B, N) in the plan dtype. Returns a dict of GPU
        tensors: 'Wx' always, 'dWx' / 'Tx' / 'w' as requested."""
        batched = (x.ndim == 2)
        B = x.shape[0] if batched else 1
        if B > self.max_batch:
            raise ValueError("batch %d exceeds the plan's max_batch %d"
                             % (B, self.max_batch))
        cols = self.M if rpadded else self.N
        shape = (B, self.na, cols) if batched else (self.na, cols)
        dev = x.device
        cdt, rdt = _CDT[self.dtype], _TDT[self.dtype]
        out = {'Wx': torch.empty(shape, dtype=cdt, device=dev)}
        if want_dWx:
            out['dWx'] = torch.empty(shape, dtype=cdt, device=dev)
        if want_Tx:
            out['Tx'] = torch.empty(shape, dtype=cdt, device=dev)
        if want_w:
            out['w'] = torch.empty(shape, dtype=rdt, device=dev)
        p = lambda k: out[k].data_ptr() if k in out else None
        check(self.lib.ssq_cwt_execute(self._h, x.data_ptr(), B, p('Wx'), p('dWx'),
                                       p('Tx'), p('w'), int(bool(rpadded)),
                                       algos.stream()))
        return out


    # ---- adjoint (autograd) -------------------------------------------------------------
    def dense_bank(self, device):
        """(na, M) real: the rows the plan applies (band-limited, Nyquist-halved, times
        sqrt(scale) when L2-normalised), on `device`."""
        if self._psih_dev is None or self._psih_dev.device != device:
            vals, off, lo, row_scale = self._bank
            P = np.zeros((self.na, self.M), dtype=vals.dtype)
            for a in range(self.na):
                P[a, lo[a]:lo[a] + (off[a + 1] - off[a])] = vals[off[a]:off[a + 1]]
            if row_scale is not None:
                P *= row_scale[:, None]
            self._psih_dev = torch.from_numpy(P).to(device)
        return self._psih_dev

    def pad_sources(self, device):
        """Index of the input sample every padded position copies (-1: a zero)."""
        if self._pad_src is None or self._pad_src.device != device:
            if self.padtype is None:
                src = np.arange(self.N)
            elif self.padtype == 'zero':
                src = np.full(self.M, -1, dtype=np.int64)
                src[self.n1:self.n1 + self.N] = np.arange(self.N)
            else:
                mode = {'reflect': 'reflect', 'replicate': 'edge', 'wrap': 'wrap',
                        'symmetric': 'symmetric'}[self.padtype]
                src = np.pad(np.arange(self.N), (self.n1, self.n2), mode=mode)
            self._pad_src = torch.from_numpy(np.ascontiguousarray(src, dtype=np.int64)).to(device)
        return self._pad_src

    def adjoint(self, gW, rpadded=False):
        """Gradient w.r.t. the real input of a real loss whose gradient w.r.t. `Wx` is `gW`
        ((na, N) / (B, na, N), or padded width if `rpadded`): with A = unpad . ifft . diag(psih)
        . fft . pad, ``Re(A^H gW)`` = pad^T Re ifft(sum_a psih_a fft(zero-extended gW_a)) -- the
        double-integral inverse's kernel (`ssq_icwt2`) followed by the adjoint of the signal
        extension."""
        cdt, rdt = _CDT[self.dtype], _TDT[self.dtype]
        batched = gW.ndim == 3
        g3 = gW if batched else gW[None]
        dev = g3.device
        psih = self.dense_bank(dev)
        src = self.pad_sources(dev)
        code = F32 if self.dtype == 'float32' else F64
        out = torch.zeros((g3.shape[0], self.N), dtype=rdt, device=dev)
        v = torch.empty(self.M, dtype=rdt, device=dev)
        keep = src >= 0
        for b in range(g3.shape[0]):
            if rpadded:
                Gp = g3[b].to(cdt).contiguous().clone()
            else:
                Gp = torch.zeros((self.na, self.M), dtype=cdt, device=dev)
                Gp[:, self.n1:self.n1 + self.N] = g3[b]
            check(self.lib.ssq_icwt2(code, Gp.data_ptr(), psih.data_ptr(), v.data_ptr(),
                                     self.na, self.M, algos.stream()))
            out[b].index_add_(0, src[keep], v[keep])
        return out if batched else out[0]


class _CwtFunction(torch.autograd.Function):
    """`Wx = plan(x)` with a backward through `Wx` (`dWx`, when requested, carries no
    gradient), so that `cwt` of a tensor that requires grad is differentiable as in the
    reference's GPU mode (examples/reconstruction.py:1-70)."""

    @staticmethod
    def forward(ctx, x, plan, want_dWx, rpadded):
        out = plan.execute(x.detach(), want_dWx=want_dWx, rpadded=rpadded)
        ctx.plan, ctx.rpadded = plan, rpadded
        if want_dWx:
            ctx.mark_non_differentiable(out['dWx'])
            return out['Wx'], out['dWx']
        return out['Wx']

    @staticmethod
    def backward(ctx, gW, *unused):
        return ctx.plan.adjoint(gW, ctx.rpadded), None, None, None


_PLAN_CACHE = {}
_PLAN_CACHE_MAX = 8
# plans hold their workspace outside torch's allocator (hipMalloc): bound the cache by bytes
# too -- the oldest plans go first
_PLAN_CACHE_MAX_BYTES = int(os.environ.get('SSQ_PLAN_CACHE_GB', '32')) << 30


def clear_plan_cache():
    """Drop every cached plan of this package -- CWT and STFT (the plans' device memory lives
    outside torch's allocator; `torch.cuda.empty_cache()` does not reach it)."""
    _PLAN_CACHE.clear()
    from . import _stft
    _stft.clear_plan_cache()


def get_cwt_plan(wavelet, scales, N, padtype, dt, l1_norm, batch, cache=True):
    """Plan lookup/creation. Plans are cached per configuration and device (the
    analogue of the reference's ``cache_wavelet`` / ``Wavelet.Psih`` cache, and of
    CuPy's kernel memoisation, utils/gpu_utils.py:17)."""
    scales = np.asarray(scales).reshape(-1)
    key = (wavelet.key(), scales.tobytes(), int(N), padtype, float(dt),
           bool(l1_norm), torch.cuda.current_device())
    plan = _PLAN_CACHE.get(key) if cache else None
    if plan is not None and plan.max_batch >= batch:
        return plan
    plan = CwtPlan(wavelet, scales, N, padtype=padtype, dt=dt, l1_norm=l1_norm,
                   max_batch=batch)
    if cache:
        _PLAN_CACHE.pop(key, None)          # a smaller-batch plan of the same configuration
        while _PLAN_CACHE and (len(_PLAN_CACHE) >= _PLAN_CACHE_MAX or
                               sum(p.device_bytes for p in _PLAN_CACHE.values())
                               + plan.device_bytes > _PLAN_CACHE_MAX_BYTES):
            _PLAN_CACHE.pop(next(iter(_PLAN_CACHE)))
        _PLAN_CACHE[key] = plan
    return plan


def _process_gmw_wavelet(wavelet, l1_norm):
    """Keep a GMW's `norm` consistent with `l1_norm` (_cwt.py:497-513)."""
    norm = 'bandpass' if l1_norm else 'energy'
    if isinstance(wavelet, str) and wavelet.lower()[:3] == 'gmw':
        wavelet = ('gmw', {'norm': norm})
    elif isinstance(wavelet, tuple) and wavelet[0].lower()[:3] == 'gmw':
        name, opts = wavelet
        opts = dict(opts)
        opts['norm'] = opts.get('norm', norm)
        wavelet = (name, opts)
    elif isinstance(wavelet, Wavelet) and wavelet.family == 'gmw':
        is_l2 = wavelet.config.get('norm') == 'energy'
        if is_l2 and l1_norm:
            raise ValueError("using GMW L2 wavelet with `l1_norm=True`")
        elif not is_l2 and not l1_norm:
            raise ValueError("using GMW L1 wavelet with `l1_norm=False`")
    return wavelet


def _zero_nonfinite_inplace(x):
    bad = ~np.isfinite(x)
    if bad.any():
        WARN("found NaN or inf values in `x`; will zero")
        x[bad] = 0.


def cwt(x, wavelet='gmw', scales='log-piecewise', fs=None, t=None, nv=32,
        l1_norm=True, derivative=False, padtype='reflect', rpadded=False,
        vectorized=True, astensor=True, cache_wavelet=None, order=0, average=None,
        nan_checks=None, patience=0):
    """Continuous Wavelet Transform via FFT convolution with frequency-domain
    wavelets matching the (padded) input's length; computed on the GPU in the
    wavelet's dtype.

    Arguments and returns follow ``ssqueezepy.cwt`` (ssqueezepy/_cwt.py:12-165):

        x: np.ndarray / torch.Tensor, 1D or 2D (2D = batch of signals along dim 0)
        wavelet: str / tuple[str, dict] / Wavelet
        scales: 'log' | 'log-piecewise' | 'linear' | 'log:maximal' ... | np.ndarray
        fs, t: sampling rate / time vector (for `dWx`)
        nv: voices per octave;  l1_norm: L1 (True) or L2 normalisation
        derivative: also return `dWx` (frequency-domain time derivative)
        padtype: 'reflect' | 'symmetric' | 'replicate' | 'wrap' | 'zero' | None
        rpadded: return the padded-width transform
        astensor: True -> torch tensors on the GPU, False -> NumPy arrays

    Returns ``(Wx, scales)`` or ``(Wx, scales, dWx)``; `Wx` is ``(na, N)``
    (``(B, na, N)`` for batched input), `scales` a NumPy vector in the wavelet
    dtype. `vectorized`, `patience` are accepted and ignored (the device path has
    one execution strategy); `cache_wavelet=False` bypasses the plan cache.
    `order > 0` / a tuple of orders: see `cwt_higher_order`.
    """
    if isinstance(order, (tuple, list, range)) or order > 0:
        kw = dict(wavelet=wavelet, scales=scales, fs=fs, t=t, nv=nv, l1_norm=l1_norm,
                  derivative=derivative, padtype=padtype, rpadded=rpadded,
                  vectorized=vectorized, patience=patience, cache_wavelet=cache_wavelet)
        return cwt_higher_order(x, order=order, average=average, astensor=astensor,
                                **kw)
    if not hasattr(x, 'ndim'):
        raise TypeError("`x` must be a numpy array or torch Tensor "
                        "(got %s)" % type(x))
    elif x.ndim not in (1, 2):
        raise ValueError("`x` must be 1D or 2D (got x.ndim == %s)" % x.ndim)
    if padtype is not None and padtype not in PADTYPES:
        raise ValueError("`padtype` must be one of: %s (got %s)"
                         % (', '.join(PADTYPES), padtype))
    if nan_checks is None:
        nan_checks = bool(isinstance(x, np.ndarray))
    if nan_checks:
        if not isinstance(x, np.ndarray):
            raise ValueError("`nan_checks=True` requires NumPy input.")
        _zero_nonfinite_inplace(x)
    if not isinstance(scales, str):
        nv = None
    N = x.shape[-1]
    dt, *_ = _process_fs_and_t(fs, t, N=N)

    use_cache = True if cache_wavelet is None else bool(cache_wavelet)
    wavelet = _process_gmw_wavelet(wavelet, l1_norm)
    wavelet = Wavelet._init_if_not_isinstance(wavelet)
    dtype = wavelet.dtype

    scales = process_scales(scales, N, wavelet, nv=nv)
    scales = np.asarray(scales, dtype=dtype)

    xd = algos.to_device(x, _TDT[dtype])
    B = xd.shape[0] if xd.ndim == 2 else 1
    plan = get_cwt_plan(wavelet, scales, N, padtype, dt, l1_norm, B, cache=use_cache)
    rp = bool(rpadded and padtype is not None)
    if isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled():
        res = _CwtFunction.apply(xd, plan, bool(derivative), rp)
        Wx, dWx = res if derivative else (res, None)
    else:
        out = plan.execute(xd, want_dWx=derivative, rpadded=rp)
        Wx, dWx = out['Wx'], out.get('dWx')
    scales = scales.squeeze()
    if not astensor:
        Wx = Wx.cpu().numpy()
        dWx = dWx.cpu().numpy() if dWx is not None else None
    return (Wx, scales, dWx) if derivative else (Wx, scales)


def cwt_higher_order(x, wavelet='gmw', order=1, average=None, astensor=True, **kw):
    """`cwt` with generalized Morse wavelets of the given order(s): lower variance, more
    noise-robust (Olhede & Walden 2002). `order`: int or tuple of ints; with a tuple and
    `average` (default True for a tuple) the transforms (and derivatives) are averaged,
    else returned as lists. `kw`: arguments of `cwt`; string `scales` are designed with
    the order-0 wavelet and shared by all orders. Returns ``(Wx, scales[, dWx])``.
    Reference: ``cwt_higher_order``, ssqueezepy/_cwt.py:517-610."""
    if isinstance(order, (list, range)):
        order = tuple(order)
    if not isinstance(order, (list, tuple)):
        order = [order]
    if len(order) == 1 and average:
        WARN("`average` ignored with single `order`")
        average = False
    wavelet = Wavelet._init_if_not_isinstance(wavelet)
    if not wavelet.name.lower().startswith('gmw'):
        raise ValueError("`wavelet` must be GMW for higher-order transforms "
                         "(got %s)" % wavelet.name)
    wavopts = wavelet.config.copy()
    wavopts.pop('order')
    wavelets = [Wavelet(('gmw', dict(order=k, **wavopts))) for k in order]
    scales = kw.get('scales', 'log-piecewise')
    if isinstance(scales, str):
        wav = Wavelet(('gmw', dict(order=0, **wavopts)))
        scales = process_scales(scales, x.shape[-1], wavelet=wav, nv=kw.get('nv', 32))
        scales = np.asarray(scales, dtype=wav.dtype)
    kw['scales'] = scales

    derivative = kw.get('derivative', False)
    Wx_all, dWx_all = [], []
    for wav in wavelets:
        out = cwt(x, wav, order=0, astensor=True, **kw)
        Wx_all.append(out[0])
        if derivative:
            dWx_all.append(out[-1])
    dWx = None
    if average or (average is None and isinstance(order, tuple)):
        Wx = torch.stack(Wx_all).mean(dim=0)
        if derivative:
            dWx = torch.stack(dWx_all).mean(dim=0)
    elif len(Wx_all) == 1:
        Wx = Wx_all[0]
        if derivative:
            dWx = dWx_all[0]
    else:
        Wx, dWx = Wx_all, (dWx_all if derivative else None)
    if not astensor:
        tonp = lambda g: ([t.cpu().numpy() for t in g] if isinstance(g, list)
                          else (g.cpu().numpy() if g is not None else None))
        Wx, dWx = tonp(Wx), tonp(dWx)
    return (Wx, scales, dWx) if derivative else (Wx, scales)
