# -*- coding: utf-8 -*-
"""Generate golden fixtures under tests/golden/ by running the REFERENCE itself.

Run only in the build container (needs /root/reference; the GPU box does not have
it):

    PYTHONPATH=oracle/refshim:/root/reference MPLBACKEND=Agg SSQ_GPU=0 \
        SSQ_PARALLEL=0 python oracle/gen_golden.py

The reference imports `numba` unconditionally and numba is not installable here,
so `oracle/refshim/numba` supplies identity decorators -- the reference's own
test-suite does the same for coverage (tests/z_all_test.py:8-20). Consequence,
recorded in every fixture as `typing='numpy'`: the float32 loop nests run with
NumPy-2 scalar promotion (all-float32) rather than numba's (float64 after the
`* 6.283...` literal); float64 runs are unaffected. See oracle/ssq_oracle.c.

Fixtures (all small, committed):
  design_*.npz   scale vectors, filter-bank samples, centre frequencies,
                 synchrosqueezing frequency grids, STFT windows
  kernels_*.npz  phase_cwt/phase_stft/ssqueeze_fast/indexed_sum_onfly/buffer
                 outputs on seeded random inputs (inputs are re-derived from the
                 seed by the tests)
  cwt_*.npz      cwt / ssq_cwt end-to-end on the reference's two-chirp test signal
  stft_*.npz     stft / ssq_stft end-to-end
"""
import os
import sys
import numpy as np

os.environ.setdefault('SSQ_GPU', '0')
os.environ['SSQ_PARALLEL'] = '0'

import ssqueezepy as sp                                        # noqa: E402
from ssqueezepy import Wavelet, cwt, stft, ssq_cwt, ssq_stft   # noqa: E402
from ssqueezepy.utils import process_scales, cwt_scalebounds   # noqa: E402
from ssqueezepy.utils import buffer, padsignal                 # noqa: E402
from ssqueezepy.wavelets import center_frequency               # noqa: E402
from ssqueezepy.ssqueezing import _compute_associated_frequencies  # noqa: E402
from ssqueezepy._stft import get_window                        # noqa: E402
from ssqueezepy import algos                                   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests',
                   'golden')
os.makedirs(OUT, exist_ok=True)


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %7.1f KB" % (name + '.npz', os.path.getsize(path) / 1e3))


def two_chirps(N, seed=0, noise=0.1, f0=None, f1=None):
    """The signal family of SURVEY.md section 8(d): two parallel linear chirps
    (`par_lchirp`, ssqueezepy/_test_signals.py:284-306) plus white noise."""
    rng = np.random.default_rng(seed)
    f0 = rng.uniform(0.01, 0.05) if f0 is None else f0
    f1 = rng.uniform(0.30, 0.45) if f1 is None else f1
    t = np.arange(N) / N
    ph = f0 * N * t + 0.5 * (f1 - f0) * N * t**2
    x = (np.cos(2 * np.pi * ph) + np.cos(2 * np.pi * (ph + 0.04 * N * t))
         + noise * rng.standard_normal(N))
    return x


# ------------------------------------------------------------------ design
def gen_design():
    d = {}
    wavs = {'gmw32': 'gmw', 'gmw64': ('gmw', {'dtype': 'float64'}),
            'morlet32': 'morlet', 'gmw64_b8': ('gmw', {'beta': 8,
                                                       'dtype': 'float64'})}
    for tag, spec in wavs.items():
        wav = Wavelet(spec)
        for N in (1000, 10000, 160000):
            for st in ('log', 'log-piecewise'):
                sc, stype, na, nv = process_scales(st, N, wav, nv=32,
                                                   get_params=True)
                d[f'scales/{tag}/{N}/{st}'] = sc
            d[f'bounds/{tag}/{N}/maximal'] = np.array(
                cwt_scalebounds(wav, N, preset='maximal'))
            d[f'bounds/{tag}/{N}/minimal'] = np.array(
                cwt_scalebounds(wav, N, preset='minimal'))
        sc = process_scales('linear', 1000, wav, get_params=False)
        d[f'scales/{tag}/1000/linear'] = sc
        # bank samples on a 4096 grid, 40 scales
        sc = process_scales('log', 2000, wav, nv=8)
        d[f'bank/{tag}/scales'] = sc
        d[f'bank/{tag}/Psih'] = wav(scale=sc, N=4096, nohalf=False)
        d[f'bank/{tag}/xi'] = wav.xifn(1., 4096)
        for kind in ('peak', 'energy', 'peak-ct'):
            d[f'wc/{tag}/{kind}'] = np.array(
                [center_frequency(wav, scale=float(s), N=4096, kind=kind)
                 for s in (sc[3, 0], sc[20, 0])])
        # ssq frequency grids, as `ssqueeze` builds them (float32 scales come back
        # from cwt in the wavelet dtype: _cwt.py:275)
        for N in (1000, 10000):
            for st in ('log', 'log-piecewise', 'linear'):
                sc = process_scales(st, N, wav, nv=32)
                scd = sc.astype(wav.dtype)
                for mr in ('peak', 'maximal', 'energy'):
                    if mr == 'maximal' and st == 'log-piecewise':
                        continue
                    if mr == 'energy' and N > 1000:
                        continue
                    d[f'ssqf/{tag}/{N}/{st}/{mr}'] = \
                        _compute_associated_frequencies(
                            scd, N, wav, st, mr, was_padded=True, dt=1.,
                            transform='cwt')
    for n_fft, win_len in ((128, 128), (1024, 1024), (256, 200), (127, 127)):
        for dtype in ('float32', 'float64'):
            w, dw = get_window(None, win_len, n_fft, derivative=True, dtype=dtype)
            d[f'window/dpss/{n_fft}/{win_len}/{dtype}'] = w
            d[f'dwindow/dpss/{n_fft}/{win_len}/{dtype}'] = dw
        w, dw = get_window('hann', win_len, n_fft, derivative=True,
                           dtype='float64')
        d[f'window/hann/{n_fft}/{win_len}/float64'] = w
        d[f'dwindow/hann/{n_fft}/{win_len}/float64'] = dw
    save('design', **d)


# ----------------------------------------------------------------- kernels
def make_ssq_freqs(M, scaletype):
    # the grids of the reference's kernel tests (tests/fft_test.py:236-246)
    if scaletype == 'log-piecewise':
        sf = np.logspace(0, np.log10(M), 2 * M)
        return np.hstack([sf[:M // 2], sf[M // 2 + 3 - 1::3]])
    elif scaletype == 'log':
        return np.logspace(0, np.log10(M), M)
    return np.linspace(0, M, M)


def gen_kernels():
    na, n = 48, 200
    gamma = 1e-2
    for dtype in ('float32', 'float64'):
        d = dict(typing=np.array('numpy'), na=na, n=n, gamma=gamma)
        np.random.seed(0)
        Wx = np.random.randn(na, n).astype(dtype) * (1 + 2j)
        dWx = np.random.randn(na, n).astype(dtype) * (2 - 1j)
        w = np.abs(np.random.randn(na, n).astype(dtype))
        w *= (2 * na / w.max())
        Sfs = np.linspace(0, .5, na).astype(dtype)
        # sprinkle sub-threshold points (an exact-zero derivative cannot be run
        # through the un-jitted 'log-piecewise' nest: round(-inf) raises,
        # SURVEY.md section 8(a'), item 7 -- that case is pinned by definition
        # in tests/test_oracle_vs_golden.py instead)
        Wx[3, 5] = 1e-4 * (1 + 1j)
        Wx[7, 9] = 0
        winf = w.copy()
        winf[2, 3] = np.inf
        d['phase_cwt'] = algos.phase_cwt_cpu(Wx, dWx, gamma, parallel=False)
        d['phase_stft'] = algos.phase_stft_cpu(Wx, dWx, Sfs, gamma,
                                               parallel=False)
        for st in ('log-piecewise', 'log', 'linear'):
            ssq_freqs = make_ssq_freqs(na, st)
            logscale = st.startswith('log')
            for flipud in (False, True):
                for ckind in ('scalar', 'vec64', 'vecdt'):
                    if flipud and ckind != 'scalar':
                        continue
                    if ckind == 'scalar':
                        const = np.log(2) / 32
                    elif ckind == 'vec64':
                        const = (np.log(2) / np.linspace(8, 32, na))
                    else:
                        const = (np.log(2) / np.linspace(8, 32, na)
                                 ).astype(dtype)
                    key = f'{st}/{int(flipud)}/{ckind}'
                    d['ssq_cwt/' + key] = algos.ssqueeze_fast(
                        Wx, dWx, ssq_freqs, const, logscale, flipud=flipud,
                        gamma=gamma, parallel=False)
                    d['isum/' + key] = algos.indexed_sum_onfly(
                        Wx, winf, ssq_freqs, const, logscale, flipud=flipud,
                        parallel=False)
        for flipud in (False, True):
            ssq_freqs = Sfs
            const = ssq_freqs[1] - ssq_freqs[0]
            d[f'ssq_stft/{int(flipud)}'] = algos.ssqueeze_fast(
                Wx, dWx, ssq_freqs, const, False, flipud=flipud, gamma=gamma,
                Sfs=Sfs, parallel=False)
        w0 = w.copy()
        algos.replace_under_abs(w0, Wx, 1.5, np.inf, parallel=False)
        d['replace_under_abs'] = w0
        x = np.random.randn(1000).astype(dtype)
        for seg, ov in ((128, 96), (127, 100), (64, 0), (33, 32)):
            for mod in (False, True):
                d[f'buffer/{seg}/{ov}/{int(mod)}'] = np.ascontiguousarray(
                    buffer(x, seg, ov, mod, parallel=False))
        save('kernels_' + dtype, **d)


# --------------------------------------------------------------- transforms
def gen_cwt():
    for dtype in ('float32', 'float64'):
        wav = Wavelet(('gmw', {'dtype': dtype}))
        d = dict(typing=np.array('numpy'))
        for N, nv in ((256, 16), (1000, 8)):
            x = two_chirps(N, seed=N)
            d[f'x/{N}'] = x
            for st in ('log', 'log-piecewise', 'linear'):
                if N != 256 and st != 'log':
                    continue
                if st == 'linear' and dtype == 'float64':
                    continue
                kw = dict(wavelet=wav, scales=st,
                          nv=nv if st != 'linear' else None)
                Tx, Wx, ssq_freqs, scales, dWx = ssq_cwt(x, **kw, get_dWx=True)
                pre = f'{N}/{st}'
                d[f'Tx/{pre}'], d[f'Wx/{pre}'] = Tx, Wx
                if N == 256:
                    d[f'dWx/{pre}'] = dWx
                d[f'ssq_freqs/{pre}'], d[f'scales/{pre}'] = ssq_freqs, scales
                if N == 256 and st == 'log':
                    out = ssq_cwt(x, **kw, get_w=True)
                    d[f'Tx_getw/{pre}'], d[f'w/{pre}'] = out[0], out[4]
                    Tx2 = ssq_cwt(x, **kw, flipud=False)[0]
                    d[f'Tx_noflip/{pre}'] = Tx2
        # fs != 1, other paddings, batched input
        x = two_chirps(300, seed=1)
        d['x/300'] = x
        Tx, Wx, sf, sc, dWx = ssq_cwt(x, wav, scales='log', nv=8, fs=400.,
                                      get_dWx=True)
        d['Tx/300/fs400'], d['Wx/300/fs400'], d['dWx/300/fs400'] = Tx, Wx, dWx
        d['ssq_freqs/300/fs400'], d['scales/300/fs400'] = sf, sc
        if dtype == 'float32':
            for pt in ('zero', 'symmetric', 'replicate', 'wrap'):
                Wx, sc = cwt(x, wav, scales='log', nv=8, padtype=pt)
                d[f'Wx/300/pad_{pt}'] = Wx
            Wx, sc = cwt(x, wav, scales='log', nv=8, padtype=None)
            d['Wx/300/pad_none'] = Wx
        xb = np.vstack([two_chirps(200, seed=s) for s in (5, 6)])
        d['x/batch200'] = xb
        Tx, Wx, sf, sc = ssq_cwt(xb, wav, scales='log', nv=8)
        d['Tx/batch200'], d['Wx/batch200'] = Tx, Wx
        save('cwt_' + dtype, **d)
    # other wavelet families, float32
    d = {}
    x = two_chirps(300, seed=3)
    d['x'] = x
    for name in ('morlet', 'bump', 'cmhat', 'hhhat'):
        Tx, Wx, sf, sc = ssq_cwt(x, name, scales='log', nv=8)
        d[f'Tx/{name}'], d[f'Wx/{name}'] = Tx, Wx
        d[f'ssq_freqs/{name}'], d[f'scales/{name}'] = sf, sc
    Wx, sc = cwt(x, 'morlet', scales='log', nv=8, l1_norm=False)
    d['Wx/morlet_l2'] = Wx
    save('cwt_families', **d)


def gen_stft():
    for dtype in ('float32', 'float64'):
        d = dict(typing=np.array('numpy'))
        for N, n_fft, hop in ((256, 64, 1), (1000, 128, 32), (2000, 256, 64),
                              (777, 100, 7)):
            x = two_chirps(N, seed=N + 1)
            pre = f'{N}/{n_fft}/{hop}'
            d['x/' + pre] = x
            Tx, Sx, sf, Sfs, dSx = ssq_stft(x, n_fft=n_fft, hop_len=hop,
                                            dtype=dtype, get_dWx=True)
            d['Tx/' + pre], d['Sx/' + pre], d['dSx/' + pre] = Tx, Sx, dSx
            d['ssq_freqs/' + pre], d['Sfs/' + pre] = sf, Sfs
        x = two_chirps(600, seed=9)
        d['x/600'] = x
        Sx, dSx = stft(x, n_fft=128, hop_len=16, modulated=False,
                       derivative=True, dtype=dtype, fs=10.)
        d['Sx/600/nomod'], d['dSx/600/nomod'] = Sx, dSx
        out = ssq_stft(x, n_fft=128, hop_len=16, dtype=dtype, get_w=True,
                       fs=10.)
        d['Tx/600/getw'], d['w/600/getw'] = out[0], out[4]
        Sx = stft(x, 'hann', n_fft=128, win_len=100, hop_len=16, dtype=dtype)
        d['Sx/600/hann100'] = Sx
        xb = np.vstack([two_chirps(400, seed=s) for s in (11, 12)])
        d['x/batch400'] = xb
        Tx, Sx, *_ = ssq_stft(xb, n_fft=64, hop_len=8, dtype=dtype)
        d['Tx/batch400'], d['Sx/batch400'] = Tx, Sx
        save('stft_' + dtype, **d)


def gen_inverse():
    """Inverses (reference: icwt _cwt.py:323-497, issq_cwt _ssq_cwt.py:313-378,
    istft _stft.py:184-256, issq_stft _ssq_stft.py:139-198) on the reference's own
    forward outputs, plus the admissibility constants they rely on."""
    from ssqueezepy import icwt, issq_cwt, istft, issq_stft
    from ssqueezepy.utils import adm_ssq, adm_cwt
    d = {}
    for name, spec in (('gmw', 'gmw'), ('gmw_l2', ('gmw', {'norm': 'energy'})),
                       ('morlet', 'morlet'), ('bump', 'bump')):
        d['adm_ssq/' + name] = np.float64(adm_ssq(spec))
        d['adm_cwt/' + name] = np.float64(adm_cwt(spec))
    for dtype in ('float32', 'float64'):
        N = 400
        x = two_chirps(N, seed=21)
        d[f'x/{dtype}'] = x
        wav = Wavelet(('gmw', {'dtype': dtype}))
        for st in ('log', 'log-piecewise', 'linear'):
            nv = 8 if st != 'linear' else None
            Tx, Wx, sf, sc = ssq_cwt(x, wav, scales=st, nv=nv)
            pre = f'{dtype}/{st}'
            d['Tx/' + pre], d['Wx/' + pre], d['scales/' + pre] = Tx, Wx, sc
            d['icwt/' + pre] = icwt(Wx, wav, scales=sc, nv=nv, x_mean=x.mean())
            d['issq/' + pre] = issq_cwt(Tx, wav)
        # L2 norm, double integral, component inversion
        wav2 = Wavelet(('gmw', {'dtype': dtype, 'norm': 'energy'}))
        Wx, sc = cwt(x, wav2, scales='log', nv=8, l1_norm=False)
        d[f'Wx/{dtype}/l2'], d[f'scales/{dtype}/l2'] = Wx, sc
        d[f'icwt/{dtype}/l2'] = icwt(Wx, wav2, scales=sc, nv=8, l1_norm=False)
        Tx = d[f'Tx/{dtype}/log']
        na = len(Tx)
        rng = np.random.default_rng(5)
        cc = np.stack([np.clip(na // 3 + rng.integers(-3, 4, N), 0, na - 1),
                       np.clip(2 * na // 3 + rng.integers(-3, 4, N), 0, na - 1)], 1)
        cc[100:140, 1] = -1                     # "no curve" marker
        cw = np.stack([np.full(N, 4), np.full(N, 6)], 1)
        d[f'cc/{dtype}'], d[f'cw/{dtype}'] = cc, cw
        d[f'issq_comp/{dtype}'] = issq_cwt(Tx, wav, cc, cw)
        xb = np.vstack([two_chirps(300, seed=s) for s in (31, 32)])
        Txb, Wxb, _, scb = ssq_cwt(xb, wav, scales='log', nv=8)
        d[f'xb/{dtype}'], d[f'Wxb/{dtype}'], d[f'scb/{dtype}'] = xb, Wxb, scb
        d[f'icwt_b/{dtype}'] = icwt(Wxb, wav, scales=scb, nv=8)
        # STFT side
        for (N, n_fft, hop, win_len, win, mod, we) in (
                (1024, 128, 16, None, None, True, 1), (1000, 100, 10, 80, 'hann', True, 1),
                (777, 64, 8, None, None, False, 0), (300, 64, 1, None, None, True, 2)):
            x = two_chirps(N, seed=N + 5)
            pre = f'{dtype}/{N}/{n_fft}/{hop}'
            Sx = stft(x, win, n_fft=n_fft, win_len=win_len, hop_len=hop, modulated=mod,
                      dtype=dtype)
            d['xs/' + pre], d['Sx/' + pre] = x, Sx
            d['istft/' + pre] = istft(Sx, win, n_fft=n_fft, win_len=win_len, hop_len=hop,
                                      N=N, modulated=mod, win_exp=we)
        x = two_chirps(300, seed=905)
        Tx, Sx, *_ = ssq_stft(x, n_fft=64, hop_len=1, dtype=dtype)
        d[f'Txs/{dtype}'] = Tx
        d[f'issq_stft/{dtype}'] = issq_stft(Tx, n_fft=64, hop_len=1)
        ccs = np.clip(12 + rng.integers(-2, 3, (Tx.shape[1], 1)), 0, 32)
        cws = np.full((Tx.shape[1], 1), 3)
        d[f'ccs/{dtype}'], d[f'cws/{dtype}'] = ccs, cws
        d[f'issq_stft_comp/{dtype}'] = issq_stft(Tx, cc=ccs, cw=cws, n_fft=64, hop_len=1)
    save('inverse', **d)


def gen_hiorder():
    """Higher-order generalized Morse wavelets (_gmw.py:268-394) and the transforms
    that use them: cwt(order=...) -> cwt_higher_order (_cwt.py:517-610), ssq_cwt(order=...)
    (_ssq_cwt.py:227-241)."""
    d = {}
    w = np.linspace(-1, 12, 527)
    for dtype in ('float32', 'float64'):
        for k in (1, 2, 3):
            wav = Wavelet(('gmw', {'order': k, 'dtype': dtype}))
            d[f'psih_l1/{dtype}/{k}'] = wav.fn(w.astype(dtype))
        N = 400
        x = two_chirps(N, seed=41)
        d[f'x/{dtype}'] = x
        wav = Wavelet(('gmw', {'dtype': dtype}))
        Wx, sc, dWx = cwt(x, wav, scales='log', nv=8, order=1, derivative=True)
        d[f'Wx1/{dtype}'], d[f'sc/{dtype}'], d[f'dWx1/{dtype}'] = Wx, sc, dWx
        Wx, sc = cwt(x, wav, scales='log', nv=8, order=(0, 2), average=True)
        d[f'Wx02/{dtype}'] = Wx
        Tx, Wx, sf, sc = ssq_cwt(x, wav, scales='log', nv=8, order=2)
        d[f'Tx2/{dtype}'], d[f'WxT2/{dtype}'], d[f'sf/{dtype}'] = Tx, Wx, sf
        Tx, Wx, sf, sc = ssq_cwt(x, wav, scales='log', nv=8, order=(0, 1, 2))
        d[f'Tx012/{dtype}'], d[f'WxT012/{dtype}'] = Tx, Wx
    for k in (1, 2):
        wav = Wavelet(('gmw', {'order': k, 'norm': 'energy', 'dtype': 'float64'}))
        d[f'psih_l2/float64/{k}'] = wav.fn(w.copy())
    save('hiorder', **d)


def gen_icwt2():
    """Double-integral inverse CWT, icwt(one_int=False) (_cwt.py:448-469), on the
    reference's own forward transforms."""
    from ssqueezepy import icwt
    d = {}
    N = 400
    x = two_chirps(N, seed=21)
    d['x'] = x
    for dtype in ('float32', 'float64'):
        wav = Wavelet(('gmw', {'dtype': dtype}))
        for st, nv in (('log', 8), ('log-piecewise', 8), ('linear', None)):
            Wx, sc = cwt(x, wav, scales=st, nv=nv)
            d[f'Wx/{dtype}/{st}'], d[f'sc/{dtype}/{st}'] = Wx, sc
            d[f'icwt2/{dtype}/{st}'] = icwt(Wx, wav, scales=sc, nv=nv, one_int=False,
                                            x_len=N, x_mean=x.mean())
    save('icwt2', **d)


def gen_ridges():
    """extract_ridges (ridge_extraction.py:11-141) on the reference's own transforms of
    a two-chirp signal, and its 3x3 example (tests/ridge_extraction_test.py:17-26)."""
    from ssqueezepy import extract_ridges
    d = {}
    tm = np.array([[1, 4, 4], [2, 2, 2], [5, 5, 4]])
    fs = np.exp([1, 2, 3])
    ri, rf, re = extract_ridges(tm, fs, penalty=2.0, get_params=True, parallel=False)
    d['basic/Tf'], d['basic/scales'] = tm, fs
    d['basic/idx'], d['basic/f'], d['basic/e'] = ri, rf, re
    N = 384
    x = two_chirps(N, seed=5, noise=0.02)
    d['x'] = x
    for dtype in ('float32', 'float64'):
        wav = Wavelet(('gmw', {'dtype': dtype}))
        Tx, Wx, ssq_freqs, scales = ssq_cwt(x, wav, nv=8)
        Ts, Sx, sf, Sfs = ssq_stft(x, n_fft=128, dtype=dtype)
        cases = (('cwt', Wx, scales, 'cwt', 15, 2.0), ('ssq_cwt', Tx, ssq_freqs, 'cwt', 4, 0.5),
                 ('stft', Sx, Sfs, 'stft', 4, 2.0), ('ssq_stft', Ts, sf, 'stft', 4, 20.0))
        for name, Tf, sc, tr, bw, pen in cases:
            ri, rf, re = extract_ridges(Tf, sc, penalty=pen, n_ridges=2, bw=bw, transform=tr,
                                        get_params=True, parallel=False)
            k = f'{name}/{dtype}/'
            d[k + 'Tf'], d[k + 'scales'] = Tf, np.asarray(sc)
            d[k + 'idx'], d[k + 'f'], d[k + 'e'] = ri, rf, re
            d[k + 'args'] = np.array([pen, bw, 0 if tr == 'cwt' else 1], dtype=np.float64)
    save('ridges', **d)


def gen_experimental():
    """freq_to_scale / scale_to_freq (experimental.py:15-143)."""
    from ssqueezepy.experimental import freq_to_scale, scale_to_freq
    d = {}
    for name in ('gmw', 'morlet', 'bump'):
        wav = Wavelet(name)
        for N in (512, 2000):
            sc = process_scales('log', N, wav, nv=8)
            d[f's2f/{name}/{N}/scales'] = sc
            d[f's2f/{name}/{N}/reflect'] = scale_to_freq(sc, wav, N, fs=2.0)
            d[f's2f/{name}/{N}/none'] = scale_to_freq(sc, wav, N, padtype=None)
        fr = np.linspace(0.02, 0.45, 24)
        d[f'f2s/{name}/freqs'] = fr
        d[f'f2s/{name}/peak'] = freq_to_scale(fr, wav, 1024)
        d[f'f2s/{name}/energy'] = freq_to_scale(fr * 4, wav, 1024, fs=4, kind='energy',
                                                n_search_scales=100, base=3)
    save('experimental', **d)


def gen_trigdiff():
    """trigdiff (utils/common.py:161-245) on the reference's own CWTs."""
    from ssqueezepy.utils.common import trigdiff
    d = {}
    N = 300
    x = two_chirps(N, seed=9)
    for dtype in ('float32', 'float64'):
        wav = Wavelet(('gmw', {'dtype': dtype}))
        Wx, sc = cwt(x, wav, nv=4)
        Wp, _ = cwt(x, wav, nv=4, rpadded=True)
        d[f'Wx/{dtype}'], d[f'Wp/{dtype}'] = Wx, Wp
        d[f'pad_N/{dtype}'] = trigdiff(Wx, fs=2.0, padtype='reflect', N=N)
        d[f'pad_zero/{dtype}'] = trigdiff(Wx, fs=1.0, padtype='zero', N=N)
        d[f'pad_quirk/{dtype}'] = trigdiff(Wx, fs=1.0, padtype='reflect')       # N=None
        d[f'rpadded/{dtype}'] = trigdiff(Wp, fs=0.5, rpadded=True, N=N)
        d[f'batched/{dtype}'] = trigdiff(np.stack([Wp, 2 * Wp[::-1]]), fs=1.0, rpadded=True, N=N)
    save('trigdiff', **d)


def gen_phase_ssq():
    """experimental.phase_ssqueeze (experimental.py:146-253) on the reference's own CWT / STFT."""
    from ssqueezepy.experimental import phase_ssqueeze
    d = {}
    N = 256
    x = two_chirps(N, seed=13)
    for dtype in ('float32', 'float64'):
        wav = Wavelet(('gmw', {'dtype': dtype}))
        Wx, sc = cwt(x, wav, nv=8)
        d[f'Wx/{dtype}'], d[f'sc/{dtype}'] = Wx, sc
        for get_w in (True, False):
            Tx, _, sf, _, _, w, dWx = phase_ssqueeze(Wx.copy(), None, scales=sc, wavelet=wav,
                                                     padtype='reflect', difftype='trig',
                                                     get_w=get_w, get_dWx=True, transform='cwt')
            k = f'cwt/{dtype}/{int(get_w)}/'
            d[k + 'Tx'], d[k + 'sf'], d[k + 'dWx'] = Tx, sf, dWx
            if get_w:
                d[k + 'w'] = w
        Sx, dSx = stft(x, n_fft=64, hop_len=2, dtype=dtype, derivative=True)
        d[f'Sx/{dtype}'], d[f'dSx/{dtype}'] = Sx, dSx
        for get_w in (True, False):
            Sfs0 = np.linspace(0, .5, len(Sx), dtype=dtype)
            Tx, _, sf, _, Sfs, w, _ = phase_ssqueeze(Sx.copy(), dSx.copy(), ssq_freqs=Sfs0, Sfs=Sfs0,
                                                     fs=1., get_w=get_w, transform='stft')
            k = f'stft/{dtype}/{int(get_w)}/'
            d[k + 'Tx'], d[k + 'sf'], d[k + 'Sfs'] = Tx, sf, Sfs
            if get_w:
                d[k + 'w'] = w
    save('phase_ssq', **d)


if __name__ == '__main__':
    which = sys.argv[1:] or ['design', 'kernels', 'cwt', 'stft', 'inverse', 'hiorder', 'icwt2', 'ridges', 'experimental', 'trigdiff', 'phase_ssq']
    print("reference: ssqueezepy", sp.__version__, "numpy", np.__version__)
    for w in which:
        globals()['gen_' + w]()
