# -*- coding: utf-8 -*-
"""Frequency-domain analytic wavelets and the filter-bank design step (host side).

This is the *design* half of the CWT hot path: it decides which Fourier bins of
which scale carry a non-negligible wavelet value and evaluates those values.
It runs once per ``(wavelet, scales, padded length)`` configuration and is cached
(the reference caches the same thing in ``Wavelet.Psih``,
``ssqueezepy/wavelets.py:135-160``); the per-call work is done by the HIP kernels.

Value parity with the reference matters here because these numbers define the
transform: sampling grid ``xi`` follows ``ssqueezepy/wavelets.py:473-484``, the
GMW formula and its float32 operation order follow ``ssqueezepy/_gmw.py:187-219``
(L1) / ``:228-252`` (L2), the other families ``ssqueezepy/wavelets.py:498-607``,
Nyquist halving ``:86-95``, centre frequency ``:611-750`` and the 1-D searches
``ssqueezepy/algos.py:625-703``. All arithmetic is NumPy in the wavelet dtype,
operation for operation, so the bank matches the reference's CPU bank bit-for-bit
on the same host.
"""
import numpy as np
from types import FunctionType
from scipy import integrate
from scipy.special import gamma as _gamma_fn

from .configs import fill_defaults

pi = np.pi

__all__ = ['Wavelet', 'center_frequency', 'xi_grid', 'find_maximum',
           'find_first_occurrence', 'morsefreq']


# --------------------------------------------------------------------- grids
def xi_grid(N, dtype=np.float64, scale=1.):
    """Radian frequencies of an N-point DFT, scaled: ``k*h`` for ``k <= N//2``,
    ``(k-N)*h`` above, ``h = scale*2pi/N`` evaluated in double and then stored in
    `dtype` (reference ``_xifn``, ssqueezepy/wavelets.py:473-484)."""
    N = int(N)
    h = scale * (2 * pi) / N
    k = np.arange(N, dtype=np.int64)
    k[N // 2 + 1:] -= N
    return (k * h).astype(dtype)


def _centered_grid(N):
    """`xi_grid(N)` rotated so negative frequencies come first (the reference's
    `aifftshift`, ssqueezepy/wavelets.py:951-965): for even N the Nyquist bin is
    kept on the positive side."""
    xi = xi_grid(N)
    if N % 2 == 0:
        out = np.zeros(N, dtype=xi.dtype)
        out[N // 2 - 1:] = xi[:N // 2 + 1]
        out[:N // 2 - 1] = xi[N // 2 + 1:]
        return out
    return np.fft.ifftshift(xi)


# ------------------------------------------------------------ wavelet families
def _as0d(*vals, dtype):
    return [np.asarray(v).astype(dtype) for v in vals]


def morsefreq(gamma, beta):
    """Peak radian frequency of a generalized Morse wavelet,
    ``(beta/gamma)**(1/gamma)`` (ssqueezepy/_gmw.py:640)."""
    return (beta / gamma) ** (1 / gamma)


def _make_gmw(gamma=None, beta=None, norm=None, order=None, centered_scale=None,
              dtype=None):
    cfg = fill_defaults('gmw', gamma=gamma, beta=beta, norm=norm, order=order,
                        centered_scale=centered_scale, dtype=dtype)
    gamma, beta, norm = cfg['gamma'], cfg['beta'], cfg['norm']
    if gamma <= 0:
        raise ValueError("`gamma` must be positive (got %s)" % gamma)
    if beta <= 0:
        raise ValueError("`beta` must be positive (got %s)" % beta)
    if norm not in ('bandpass', 'energy'):
        raise ValueError("`norm` must be 'energy' or 'bandpass' (got %s)" % norm)
    order = int(cfg['order'])
    if order < 0:
        raise ValueError("`order` must be >= 0 (got %s)" % order)
    dt = str(np.dtype(cfg['dtype']))
    if norm == 'energy' and dt == 'float32':
        raise ValueError("`norm='energy'` w/ `dtype='float32'` is unsupported; "
                         "use 'float64' instead.")
    wc_py = morsefreq(gamma, beta)
    centered = bool(cfg['centered_scale'])
    if order > 0:
        return _make_gmw_k(gamma, beta, order, norm, wc_py, centered, dt), cfg

    if norm == 'bandpass':
        g, b, wc, wcl = _as0d(gamma, beta, wc_py, np.log(wc_py), dtype=dt)

        def gmw_l1(w):
            w = np.atleast_1d(np.asarray(w * wc if centered else w, dtype=dt))
            if not w.flags.writeable or w.base is not None:
                w = w.copy()
            keep = (w >= 0)
            w *= keep                       # zero negative w to avoid nans
            with np.errstate(divide='ignore', invalid='ignore'):
                return 2 * np.exp(- b * wcl + wc**g
                                  + b * np.log(w) - w**g) * keep
        return gmw_l1, cfg

    r_py = (2 * beta + 1) / gamma
    g, b, wc, r, rg = _as0d(gamma, beta, wc_py, r_py, _gamma_fn(r_py), dtype=dt)

    def gmw_l2(w):
        w = np.atleast_1d(np.asarray(w * wc if centered else w, dtype=dt))
        if not w.flags.writeable or w.base is not None:
            w = w.copy()
        keep = (w >= 0)
        w *= keep
        return np.sqrt(2. * pi * g * 2.**r / rg) * w**b * np.exp(-w**g) * keep
    return gmw_l2, cfg


def _gmw_k_constants(gamma, beta, k, norm, dt):
    """Coefficients of the order-k generalized Morse wavelet: generalized Laguerre
    polynomial L_k^(c)(2 w^gamma), c = (2 beta + 1)/gamma - 1, times the normalisation
    (Olhede & Walden 2002; ssqueezepy/_gmw.py:366-394)."""
    from scipy.special import gammaln
    r = (2 * beta + 1) / gamma
    c = r - 1
    if norm == 'bandpass':
        coeff = np.sqrt(np.exp(gammaln(r) + gammaln(k + 1) - gammaln(k + r)))
    else:
        coeff = np.sqrt(2 * pi * gamma * (2**r) * np.exp(gammaln(k + 1) - gammaln(k + r)))
    L = np.zeros(k + 1, dtype=dt)
    for m in range(k + 1):
        fact = np.exp(gammaln(k + c + 1) - gammaln(c + m + 1) - gammaln(k - m + 1))
        L[m] = (-1)**m * fact / _gamma_fn(m + 1)
    kc = L * coeff
    if norm == 'bandpass':
        kc *= 2
    return kc.astype(dt)


def _make_gmw_k(gamma, beta, k, norm, wc_py, centered, dt):
    """Order-k GMW in the frequency domain (ssqueezepy/_gmw.py:268-365): the order-0
    envelope times the Laguerre polynomial in ``2 w**gamma``."""
    kc = _gmw_k_constants(gamma, beta, k, norm, dt)
    g, b, wc = _as0d(gamma, beta, wc_py, dtype=dt)

    def gmw_k(w):
        w = np.atleast_1d(np.asarray(w * wc if centered else w, dtype=dt))
        if not w.flags.writeable or w.base is not None:
            w = w.copy()
        keep = (w >= 0)
        w *= keep                           # zero negative w to avoid nans
        C = np.zeros(w.shape, dtype=w.dtype)
        with np.errstate(divide='ignore', invalid='ignore'):
            for m in range(len(kc)):
                C += kc[m] * (2 * w**g)**m
            if norm == 'bandpass':
                return C * np.exp(- b * np.log(wc) + wc**g + b * np.log(w) - w**g) * keep
            return C * np.exp(b * np.log(w) - w**g) * keep
    return gmw_k


def _make_morlet(mu=None, dtype=None):
    cfg = fill_defaults('morlet', mu=mu, dtype=dtype)
    mu, dt = cfg['mu'], str(np.dtype(cfg['dtype']))
    cs = (1 + np.exp(-mu**2) - 2 * np.exp(-3 / 4 * mu**2)) ** (-.5)
    ks = np.exp(-.5 * mu**2)
    mu, cs, ks = _as0d(mu, cs, ks, dtype=dt)
    C = np.asarray([-.5, np.sqrt(2) * cs * pi**.25], dtype=dt)

    def morlet(w):
        w = np.atleast_1d(np.asarray(w, dtype=dt))
        return C[1] * (np.exp(C[0] * (w - mu)**2) - ks * np.exp(C[0] * w**2))
    return morlet, cfg


def _make_bump(mu=None, s=None, om=None, dtype=None):
    cfg = fill_defaults('bump', mu=mu, s=s, om=om, dtype=dtype)
    rdt = str(np.dtype(cfg['dtype']))
    cdt = 'complex64' if rdt == 'float32' else 'complex128'
    mu, s, om = [np.asarray(g, cdt) for g in (cfg['mu'], cfg['s'], cfg['om'])]
    C = np.asarray([2 * pi * 1j * om, .443993816053287], dtype=cdt)
    C0 = np.asarray(.999, dtype=rdt)

    def bump(w):
        w = np.atleast_1d(np.asarray(w, dtype=cdt))
        _w = (w - mu) / s
        with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
            inside = (np.abs(_w) < C0)
            return (np.exp(C[0] * w) / s * inside
                    * np.exp(-1 / (1 - (_w * inside)**2)) / C[1])
    return bump, cfg


def _make_cmhat(mu=None, s=None, dtype=None):
    cfg = fill_defaults('cmhat', mu=mu, s=s, dtype=dtype)
    dt = str(np.dtype(cfg['dtype']))
    mu, s = _as0d(cfg['mu'], cfg['s'], dtype=dt)
    C = np.asarray([5 / 2, 2 * np.sqrt(2 / 3) * pi**(-1 / 4)], dtype=dt)

    def cmhat(w):
        _w = np.atleast_1d(np.asarray(w, dtype=dt)) - mu
        return C[1] * (s**C[0] * _w**2 * np.exp(-s**2 * _w**2 / 2) * (_w >= 0))
    return cmhat, cfg


def _make_hhhat(mu=None, dtype=None):
    cfg = fill_defaults('hhhat', mu=mu, dtype=dtype)
    dt = str(np.dtype(cfg['dtype']))
    mu, = _as0d(cfg['mu'], dtype=dt)
    C = np.asarray([-1 / 2, 2 / np.sqrt(5) * pi**(-1 / 4)], dtype=dt)

    def hhhat(w):
        _w = np.atleast_1d(np.asarray(w, dtype=dt)) - mu
        return C[1] * (_w * (1 + _w) * np.exp(C[0] * _w**2)) * (1 + np.sign(_w))
    return hhhat, cfg


_FAMILIES = {'gmw': _make_gmw, 'morlet': _make_morlet, 'bump': _make_bump,
             'cmhat': _make_cmhat, 'hhhat': _make_hhhat}


# ------------------------------------------------------------------- Wavelet
class _FnIdentity:
    """Cache-key component for a user-supplied wavelet function: equal only to a key holding
    the *same* function object, and it keeps that object alive, so that CPython cannot hand
    its id() to a different function while a cached plan / design still refers to it."""
    __slots__ = ('fn',)

    def __init__(self, fn):
        self.fn = fn

    def __hash__(self):
        return id(self.fn)

    def __eq__(self, other):
        return isinstance(other, _FnIdentity) and other.fn is self.fn


class Wavelet():
    """Frequency-domain wavelet, sampled as ``psih(scale * xi)``.

    Drop-in for the reference's ``wavelets.Wavelet`` on the forward-transform
    path (ssqueezepy/wavelets.py:14-470): same constructor
    (``'gmw'`` / ``('morlet', {'mu': 5})`` / custom function), same ``dtype``
    rule (the transform computes in the wavelet's dtype), same call convention.
    Values are NumPy arrays on the host; the device copy of the bank is owned by
    the transform plan (see `_cwt.py`).
    """
    SUPPORTED = {'gmw', 'morlet', 'bump', 'cmhat', 'hhhat'}
    DTYPES = {'float32', 'float64'}

    def __init__(self, wavelet='gmw', N=1024, dtype=None):
        self._dtype = self._norm_dtype(dtype) if dtype is not None else None
        self._set_wavelet(wavelet)
        self.N = N

    # -- construction
    @staticmethod
    def _norm_dtype(dtype):
        name = str(dtype).split('.')[-1].strip("<>' ")
        if name.startswith('class'):
            name = name.split("'")[-2]
        name = str(np.dtype(name)) if name not in Wavelet.DTYPES else name
        if name not in Wavelet.DTYPES:
            raise ValueError("`dtype` must be one of: %s (got %s)"
                             % (', '.join(sorted(Wavelet.DTYPES)), dtype))
        return name

    @classmethod
    def _init_if_not_isinstance(cls, wavelet, **kw):
        if isinstance(wavelet, Wavelet):
            return wavelet
        return Wavelet(wavelet, **kw)

    def _set_wavelet(self, wavelet):
        if isinstance(wavelet, FunctionType):
            self.fn, self.config, self.family = wavelet, {}, None
            out = np.asarray(self.fn(np.asarray([1.], dtype='float32')))
            od = str(out.dtype)
            self._dtype = ('float32' if od in ('float32', 'complex64')
                           else 'float64')
            return
        err = ("`wavelet` must be one of: (1) string name of supported wavelet; "
               "(2) tuple of (1) and dict of wavelet parameters (e.g. {'mu': 5}); "
               "(3) custom function taking `scale * xi` as input. (got: %s)"
               % str(wavelet))
        if isinstance(wavelet, tuple):
            if not (len(wavelet) == 2 and isinstance(wavelet[1], dict)):
                raise TypeError(err)
            name, opts = wavelet[0], dict(wavelet[1])
        elif isinstance(wavelet, str):
            name, opts = wavelet, {}
        else:
            raise TypeError(err)
        name = name.lower()
        if name not in Wavelet.SUPPORTED:
            raise ValueError("`wavelet` must be one of: %s (got %s)"
                             % (', '.join(sorted(Wavelet.SUPPORTED)), name))
        passed32 = any('float32' in str(t) for t in
                       (self._dtype, opts.get('dtype', 0)))
        if name == 'gmw' and opts.get('norm', 'bandpass') == 'energy':
            if passed32:
                import logging
                logging.warning("WARNING: `norm='energy'` w/ `dtype='float32'` is"
                                " unsupported; will use 'float64' instead.")
            opts['dtype'] = 'float64'
            self._dtype = 'float64'
        elif self._dtype is not None:
            opts['dtype'] = self._dtype
        if 'dtype' in opts and opts['dtype'] is not None:
            opts['dtype'] = self._norm_dtype(opts['dtype'])
        self.fn, self.config = _FAMILIES[name](**opts)
        self.family = name
        if self._dtype is None:
            self._dtype = self._norm_dtype(self.config['dtype'])

    # -- properties
    @property
    def dtype(self):
        return self._dtype

    @property
    def N(self):
        return self._N

    @N.setter
    def N(self, value):
        self._N = int(value)
        self._xi = xi_grid(self._N, dtype=self._dtype)

    @property
    def xi(self):
        return self._xi

    @property
    def name(self):
        return self.family or getattr(self.fn, '__name__', 'custom')

    def key(self):
        """Hashable identity of the underlying function (plan-cache key)."""
        if self.family is None:
            return ('fn', _FnIdentity(self.fn), self._dtype)
        return (self.family, self._dtype,
                tuple(sorted((k, str(v)) for k, v in self.config.items())))

    # -- evaluation
    def xifn(self, scale=None, N=None):
        if isinstance(scale, np.ndarray) and scale.size > 1:
            if scale.squeeze().ndim > 1:
                raise ValueError("2D `scale` unsupported")
            scale = scale.reshape(-1, 1)
        elif scale is None:
            scale = 1.
        scale = np.asarray(scale, dtype=self._dtype)
        base = self._xi if N is None else xi_grid(N, dtype=self._dtype)
        return scale * base

    def __call__(self, w=None, *, scale=None, N=None, nohalf=True, imag_th=1e-8):
        """``wavelet(w)``, or ``wavelet(scale * xi)`` when called by keyword
        (reference ``Wavelet.__call__``, ssqueezepy/wavelets.py:62-84).
        ``nohalf=False`` halves the Nyquist bin of even-length outputs."""
        if w is not None:
            psih = self.fn(np.asarray(w, dtype=self._dtype))
        else:
            psih = self.fn(self.xifn(scale, N))
        if not nohalf:
            n = psih.shape[-1]
            if n % 2 == 0:
                psih[..., n // 2] /= 2
        if (np.iscomplexobj(psih) and imag_th is not None and
                (psih.imag.sum() / psih.real.sum() < imag_th)):
            psih = psih.real
        return psih

    def Psih(self, scale=None, N=None, nohalf=True):
        """Dense bank with a one-entry cache (reference ``Wavelet.Psih``)."""
        pN = getattr(self, '_Psih_N', -1)
        ps = getattr(self, '_Psih_scale', np.array([-1]))
        n_is_none = N is None
        N = N or self.N
        if ((scale is None and n_is_none) or
                (N == pN and len(scale) == len(ps) and np.allclose(scale, ps))):
            return self._Psih
        self._Psih = self(scale=scale, N=N, nohalf=nohalf)
        self._Psih_N = N
        self._Psih_scale = np.array(scale, copy=True)
        return self._Psih


# ------------------------------------------------------------- 1-D searches
def _search_windows(step_start, step_size, steps_per_search):
    """Successive half-open windows ``[start, start+inc)`` sampled at
    `steps_per_search` points — the scan grid shared by the two searches below
    (ssqueezepy/algos.py:625-703). The grid, not just the optimum, is part of
    the contract: results are grid points."""
    n = int(steps_per_search)
    inc = int(n * step_size)
    idx = 0
    while True:
        lo = step_start + inc * idx
        yield np.linspace(lo, lo + inc, n, endpoint=False)
        idx += 1


def find_maximum(fn, step_size=1e-3, steps_per_search=1e4, step_start=0,
                 step_limit=1000, min_value=-1):
    """Location and value of the (single) maximum of ``|fn|`` scanning upward
    from `step_start` (reference ``find_maximum``, ssqueezepy/algos.py:625-661)."""
    best, best_x = min_value, None
    for xs in _search_windows(step_start, step_size, steps_per_search):
        ys = np.abs(np.asarray(fn(xs)))
        top = ys.max()
        if top > best:
            best, best_x = top, xs[np.argmax(ys)]
        elif top < best:
            return best_x, best
        if xs.max() > step_limit:
            raise ValueError("could not find function maximum with given "
                             "(step_size, steps_per_search, step_start, "
                             "step_limit, min_value)=(%s, %s, %s, %s, %s)"
                             % (step_size, steps_per_search, step_start,
                                step_limit, min_value))


def find_first_occurrence(fn, value, step_size=1e-3, steps_per_search=1e4,
                          step_start=0, step_limit=1000):
    """Earliest grid point where ``|fn|`` comes within one grid-step of `value`
    (reference ``find_first_occurrence``, ssqueezepy/algos.py:664-703)."""
    for xs in _search_windows(step_start, step_size, steps_per_search):
        hit_limit = bool(xs.max() > step_limit)
        if hit_limit:
            xs = np.clip(xs, None, step_limit)
        ys = np.abs(np.asarray(fn(xs))).astype(np.float64)
        tol = np.abs(np.diff(ys)).max()
        dist = np.abs(ys - value)
        if np.any(dist <= tol):
            i = int(np.argmin(dist))
            return xs[i], ys[i]
        if hit_limit:
            raise ValueError("could not find input value to yield function "
                             "output value=%s within step_limit=%s"
                             % (value, step_limit))


# --------------------------------------------------------- centre frequency
def center_frequency(wavelet, scale=None, N=1024, kind='energy', force_int=None):
    """Radian centre frequency of `wavelet` at `scale` on an N-point grid:
    'peak' (argmax of ``|psih|^2`` on the grid), 'energy' (energy-weighted mean)
    or 'peak-ct' (continuous-time peak). Reference ``center_frequency``,
    ssqueezepy/wavelets.py:611-750; 'peak' is what ``maprange='peak'`` uses to
    place the synchrosqueezing frequency axis."""
    if kind not in ('energy', 'peak', 'peak-ct'):
        raise ValueError("`kind` must be one of: energy, peak, peak-ct "
                         "(got %s)" % kind)
    wavelet = Wavelet._init_if_not_isinstance(wavelet)

    def sampled(scale):
        w = _centered_grid(N)
        psih = np.asarray(wavelet(np.asarray(scale) * w))
        return w, np.abs(psih)**2

    if scale is None and kind != 'peak-ct':
        scale = (4 / pi) * find_maximum(wavelet.fn)[0]

    if kind == 'peak-ct':
        return float(find_maximum(wavelet.fn)[0])
    if kind == 'peak':
        w, e = sampled(scale)
        return float(w[np.argmax(e)])
    # 'energy' -- the reference always integrates on the grid at `scale`
    # (`force_int or True`, ssqueezepy/wavelets.py:742-743); `force_int` is
    # accepted for signature compatibility only
    w, e = sampled(scale)
    return float(integrate.trapezoid(e * w) / integrate.trapezoid(e))
