# -*- coding: utf-8 -*-
"""Differential fuzz against the REFERENCE ITSELF (build container only: needs /root/reference).

Runs ssqueezepy (imported with the numba stand-in of oracle/refshim, i.e. its CPU path with the
loop nests as plain Python) and ssqueezepy_amd (its kernels under the CPU emulator, tests/emu/)
in one process on random signals and random option combinations of the public API, and compares
what the two return:
    PYTHONPATH=oracle/refshim:/root/reference MPLBACKEND=Agg SSQ_GPU=0 SSQ_PARALLEL=0 \
        python tools/diff_fuzz_reference.py [n_cases] [seed]
`Wx`, `dWx`, `Sx`, inverses: 1e-5 / 1e-11 relative. `scales`, `ssq_freqs`: exact. `Tx` moves whole
bins under last-bit input changes, so: column sums (assignment-invariant) and the fraction of
entries that differ (<= 1 %)."""
import os, sys, warnings, logging
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('SSQ_GPU', '0'); os.environ['SSQ_PARALLEL'] = '0'
logging.disable(logging.WARNING)
warnings.simplefilter('ignore')
import ssqueezepy as R                      # the reference
import emu_backend
from conftest import two_chirps


def relmax(a, b):
    b = np.asarray(b)
    return float(np.abs(np.asarray(a) - b).max()) / max(float(np.abs(b).max()), 1e-300)


def np_(t):
    return t.detach().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t)


def same_Tx(Tx, ref, tol):
    cs, cr = Tx.sum(-2), ref.sum(-2)
    e = float(np.abs(cs - cr).max()) / max(float(np.abs(cr).max()), 1e-300)
    frac = (np.abs(Tx - ref) > 1e-5 * np.abs(ref).max()).mean()
    extra = ''
    if not np.isfinite(e):
        extra = ' [nan ours %d ref %d, inf ours %d ref %d]' % (np.isnan(Tx).sum(), np.isnan(ref).sum(),
                                                           np.isinf(Tx).sum(), np.isinf(ref).sum())
    return e <= 100 * tol and frac <= 1e-2, 'colsum %.1e differ %.1e%s' % (e, frac, extra)


def main(n_cases=40, seed=0):
    rng = np.random.default_rng(seed)
    bad = 0
    with emu_backend.emulated() as S:
        for case in range(n_cases):
            dtype = str(rng.choice(['float32', 'float64']))
            tol = 1e-5 if dtype == 'float32' else 1e-11
            kind = rng.choice(['ssq_cwt', 'cwt', 'ssq_stft', 'inverse', 'ridges', 'extras'])
            if os.environ.get('FUZZ_KIND'):           # e.g. FUZZ_KIND=extras/opts2
                kind = os.environ['FUZZ_KIND'].split('/')[0]
            N = int(rng.integers(64, 400))
            x = two_chirps(N, seed=case + 1000 * seed)
            fam = str(rng.choice(['gmw', 'morlet', 'bump', 'cmhat', 'hhhat']))
            nv = int(rng.choice([4, 8, 16]))
            pad = str(rng.choice(['reflect', 'zero', 'symmetric', 'wrap', 'replicate']))
            st = str(rng.choice(['log', 'log-piecewise', 'linear']))
            fs = float(rng.choice([1.0, 2.5, 100.0]))
            desc, ok, info = '', True, ''
            try:
                if kind == 'cwt':
                    l1 = bool(rng.random() < 0.7)
                    kw = dict(scales=st, nv=nv, padtype=pad, fs=fs, l1_norm=l1, derivative=True,
                              rpadded=bool(rng.random() < 0.2))
                    desc = f'cwt {dtype} {fam} {kw}'
                    a = R.cwt(x, R.Wavelet((fam, {'dtype': dtype})), **kw)
                    b = S.cwt(x, S.Wavelet((fam, {'dtype': dtype})), **kw)
                    ok = (np.array_equal(np_(b[1]), a[1]) and relmax(np_(b[0]), a[0]) <= tol
                          and relmax(np_(b[2]), a[2]) <= tol)
                    info = 'eW %.1e eD %.1e' % (relmax(np_(b[0]), a[0]), relmax(np_(b[2]), a[2]))
                elif kind == 'ssq_cwt':
                    kw = dict(scales=st, nv=nv, padtype=pad, fs=fs,
                              squeezing=str(rng.choice(['sum', 'lebesgue', 'abs'])),
                              maprange=str(rng.choice(['peak', 'maximal', 'energy'])),
                              flipud=bool(rng.random() < 0.7),
                              preserve_transform=bool(rng.random() < 0.5))
                    if rng.random() < 0.3:
                        kw['gamma'] = float(rng.choice([1e-3, 1e-1]))
                    desc = f'ssq_cwt {dtype} {fam} N={N} {kw}'
                    a = R.ssq_cwt(x, R.Wavelet((fam, {'dtype': dtype})), **kw)
                    b = S.ssq_cwt(x, S.Wavelet((fam, {'dtype': dtype})), **kw)
                    okT, info = same_Tx(np_(b[0]), a[0], tol)
                    ok = (okT and relmax(np_(b[1]), a[1]) <= tol and np.array_equal(np_(b[2]), a[2])
                          and np.array_equal(np_(b[3]), a[3]))
                elif kind == 'ssq_stft':
                    n_fft = int(rng.choice([32, 50, 64, 128]))
                    n_fft = min(n_fft, N // 2)
                    kw = dict(n_fft=n_fft, hop_len=int(rng.integers(1, n_fft // 2 + 1)), fs=fs,
                              modulated=bool(rng.random() < 0.7), padtype=pad, dtype=dtype,
                              squeezing=str(rng.choice(['sum', 'lebesgue'])),
                              flipud=bool(rng.random() < 0.3))
                    if rng.random() < 0.4:
                        kw['window'] = str(rng.choice(['hann', 'hamming', 'blackman']))
                    desc = f'ssq_stft N={N} {kw}'
                    a = R.ssq_stft(x, **kw)
                    b = S.ssq_stft(x, **kw)
                    okT, info = same_Tx(np_(b[0]), a[0], tol)
                    ok = (okT and relmax(np_(b[1]), a[1]) <= tol and np.array_equal(np_(b[2]), a[2])
                          and np.array_equal(np_(b[3]), a[3]))
                elif kind == 'inverse':
                    wa, wb = R.Wavelet((fam, {'dtype': dtype})), S.Wavelet((fam, {'dtype': dtype}))
                    st2 = str(rng.choice(['log', 'log-piecewise']))
                    Tx, Wx, sf, sc = R.ssq_cwt(x, wa, scales=st2, nv=nv)
                    desc = f'inverses {dtype} {fam} {st2} nv={nv} N={N}'
                    e1 = relmax(np_(S.issq_cwt(Tx, wb)), R.issq_cwt(Tx, wa))
                    e2 = relmax(np_(S.icwt(Wx, wb, scales=sc, nv=nv)), R.icwt(Wx, wa, scales=sc, nv=nv))
                    n_fft = int(min(64, N // 2))
                    Sx = R.stft(x, n_fft=n_fft, hop_len=4, dtype=dtype)
                    e3 = relmax(np_(S.istft(Sx, n_fft=n_fft, hop_len=4, N=N)),
                                R.istft(Sx, n_fft=n_fft, hop_len=4, N=N))
                    ok = max(e1, e2) <= 1e-6 and e3 <= 100 * tol
                    info = 'issq_cwt %.1e icwt %.1e istft %.1e' % (e1, e2, e3)
                elif kind == 'extras':
                    wa, wb = R.Wavelet(('gmw', {'dtype': dtype})), S.Wavelet(('gmw', {'dtype': dtype}))
                    sub = str(rng.choice(['batch_w', 'hiorder', 'trigdiff', 'phase_ssq', 'tvec', 'stft_opts',
                                          'components', 'icwt_opts', 'opts2']))
                    if '/' in os.environ.get('FUZZ_KIND', ''):
                        sub = os.environ['FUZZ_KIND'].split('/')[1]
                    desc = f'extras/{sub} {dtype} N={N} nv={nv}'
                    if sub == 'batch_w':          # batched input; get_w / get_dWx outputs
                        xb = np.stack([x, x[::-1].copy()])
                        a = R.ssq_cwt(xb, wa, nv=nv); b = S.ssq_cwt(xb, wb, nv=nv)
                        ok1, info = same_Tx(np_(b[0]), a[0], tol)
                        a2 = R.ssq_cwt(x, wa, nv=nv, get_w=True, get_dWx=True)
                        b2 = S.ssq_cwt(x, wb, nv=nv, get_w=True, get_dWx=True)
                        wr, wo = a2[4], np_(b2[4])
                        fin = np.isfinite(wr) & np.isfinite(wo)
                        ew = float(np.abs(wo[fin] - wr[fin]).max() / np.abs(wr[fin]).max())
                        ok = (ok1 and relmax(np_(b[1]), a[1]) <= tol and relmax(np_(b2[5]), a2[5]) <= tol
                              and (np.isfinite(wr) == np.isfinite(wo)).mean() > 0.999 and ew < 1e-2)
                        info += ' | w %.1e' % ew
                    elif sub == 'hiorder':
                        order = (0, 1, 2) if rng.random() < 0.5 else 1
                        a = R.ssq_cwt(x, wa, nv=nv, order=order); b = S.ssq_cwt(x, wb, nv=nv, order=order)
                        okT, info = same_Tx(np_(b[0]), a[0], tol)
                        ok = okT and relmax(np_(b[1]), a[1]) <= 10 * tol
                    elif sub == 'trigdiff':
                        Wx, sc = R.cwt(x, wa, nv=nv)
                        a = R.utils.common.trigdiff(Wx, fs=fs, padtype=pad, N=N)
                        b = S.trigdiff(Wx, fs=fs, padtype=pad, N=N)
                        ok = relmax(np_(b), a) <= tol
                        info = 'e %.1e' % relmax(np_(b), a)
                    elif sub == 'phase_ssq':
                        from ssqueezepy.experimental import phase_ssqueeze as rps
                        Wx, sc = R.cwt(x, wa, nv=nv)
                        a = rps(Wx.copy(), None, scales=sc, wavelet=wa, padtype='reflect', difftype='trig',
                                get_w=bool(rng.random() < 0.5), transform='cwt')
                        b = S.phase_ssqueeze(Wx.copy(), None, scales=sc, wavelet=wb, padtype='reflect',
                                             difftype='trig', get_w=(a[5] is not None), transform='cwt')
                        okT, info = same_Tx(np_(b[0]), a[0], tol)
                        ok = okT and np.array_equal(np_(b[2]), a[2])
                    elif sub == 'opts2':          # wavelet parameters, scale arrays, ssq_freqs / maprange forms
                        wcfg = {'gmw': {'gamma': float(rng.choice([2, 3, 4])), 'beta': float(rng.choice([20, 60, 90]))},
                                'morlet': {'mu': float(rng.choice([5, 10, 13.4]))},
                                'bump': {'mu': float(rng.choice([4, 6])), 's': float(rng.choice([0.8, 1.2]))},
                                'cmhat': {'mu': float(rng.choice([1, 2]))}, 'hhhat': {'mu': float(rng.choice([4, 6]))}}[fam]
                        wcfg['dtype'] = dtype
                        wa, wb = R.Wavelet((fam, wcfg)), S.Wavelet((fam, wcfg))
                        kw2 = {}
                        if rng.random() < 0.5:
                            kw2['scales'] = (2 ** np.linspace(1, 6, int(rng.integers(10, 40)))).astype(dtype)
                        else:
                            kw2['scales'] = str(rng.choice(['log:maximal', 'log-piecewise:minimal', 'log']))
                            kw2['nv'] = nv
                        r3 = rng.random()
                        if r3 < 0.3:
                            kw2['ssq_freqs'] = str(rng.choice(['log', 'linear']))
                        elif r3 < 0.5:
                            kw2['maprange'] = (0.02 * fs, 0.4 * fs)
                        elif r3 < 0.65:
                            kw2['ssq_freqs'] = np.linspace(0.01 * fs, 0.45 * fs, 50)
                        kw2['padtype'] = pad if rng.random() < 0.8 else None
                        kw2['fs'] = fs
                        desc += ' %s %s' % (fam, {k: (v if not isinstance(v, np.ndarray) else 'array[%d]' % len(v)) for k, v in kw2.items()})
                        a = R.ssq_cwt(x, wa, **kw2); b = S.ssq_cwt(x, wb, **kw2)
                        okT, info = same_Tx(np_(b[0]), a[0], tol)
                        ok = (okT and relmax(np_(b[1]), a[1]) <= tol and np.array_equal(np_(b[2]), a[2])
                              and np.array_equal(np_(b[3]), a[3]))
                    elif sub == 'stft_opts':      # win_len < n_fft, window arrays, get_w, no padding
                        n_fft = int(rng.choice([32, 64, 96]))
                        n_fft = min(n_fft, N // 2)
                        kw2 = dict(n_fft=n_fft, win_len=int(rng.integers(n_fft // 2, n_fft + 1)),
                                   hop_len=int(rng.integers(1, 9)), dtype=dtype,
                                   modulated=bool(rng.random() < 0.5))
                        if rng.random() < 0.4:
                            kw2['window'] = np.hanning(kw2['win_len']).astype(dtype)
                        a = R.ssq_stft(x, get_w=True, get_dWx=True, **kw2)
                        b = S.ssq_stft(x, get_w=True, get_dWx=True, **kw2)
                        okT, info = same_Tx(np_(b[0]), a[0], tol)
                        wr, wo = a[4], np_(b[4])
                        fin = np.isfinite(wr) & np.isfinite(wo)
                        ok = (okT and relmax(np_(b[1]), a[1]) <= tol and relmax(np_(b[5]), a[5]) <= tol
                              and (np.isfinite(wr) == np.isfinite(wo)).mean() > 0.999
                              and np.array_equal(np_(b[2]), a[2]))
                        desc += ' %s' % {k: v for k, v in kw2.items() if k != 'window'}
                    elif sub == 'components':     # component inversion around curves
                        Tx, Wx, sf, sc = R.ssq_cwt(x, wa, nv=nv)
                        na = len(Tx)
                        K = int(rng.integers(1, 3))
                        cc = np.stack([np.clip((na * (0.3 + 0.4 * k / K) + 3 * np.sin(np.arange(N) / 20)), 0, na - 1)
                                       for k in range(K)], axis=1).astype(int)
                        cw = np.full((N, K), int(rng.integers(1, 6)))
                        a = R.issq_cwt(Tx, wa, cc, cw); b = S.issq_cwt(Tx, wb, cc, cw)
                        e1 = relmax(np_(b), a)
                        Ts, Sx, sfs, Sfs = R.ssq_stft(x, n_fft=64, dtype=dtype)
                        ns = len(Ts)
                        cc2 = np.full((N, 1), ns // 3); cw2 = np.full((N, 1), 4)
                        e2 = relmax(np_(S.issq_stft(Ts, cc=cc2, cw=cw2, n_fft=64)), R.issq_stft(Ts, cc=cc2, cw=cw2, n_fft=64))
                        ok = e1 <= 1e-6 and e2 <= 1e-6
                        info = 'issq_cwt comps %.1e issq_stft comps %.1e' % (e1, e2)
                    elif sub == 'icwt_opts':      # double integral, x_mean, L2 norm, padded input
                        st2 = str(rng.choice(['log', 'log-piecewise', 'linear']))
                        Wx, sc = R.cwt(x, wa, scales=st2, nv=nv)
                        xm = float(x.mean())
                        e1 = relmax(np_(S.icwt(Wx, wb, scales=sc, nv=nv, one_int=False, x_len=N, x_mean=xm)),
                                    R.icwt(Wx, wa, scales=sc, nv=nv, one_int=False, x_len=N, x_mean=xm))
                        Wp, _ = R.cwt(x, wa, scales=st2, nv=nv, rpadded=True)
                        e2 = relmax(np_(S.icwt(Wp, wb, scales=sc, nv=nv, x_len=N, rpadded=True, x_mean=xm)),
                                    R.icwt(Wp, wa, scales=sc, nv=nv, x_len=N, rpadded=True, x_mean=xm))
                        ok = e1 <= 100 * tol and e2 <= 1e-6
                        info = '%s icwt2 %.1e icwt(rpadded) %.1e' % (st2, e1, e2)
                    else:                         # non-uniformly scaled time vector instead of fs
                        tv = np.linspace(0., N / fs, N, endpoint=False)
                        a = R.ssq_cwt(x, wa, nv=nv, t=tv); b = S.ssq_cwt(x, wb, nv=nv, t=tv)
                        okT, info = same_Tx(np_(b[0]), a[0], tol)
                        ok = okT and np.array_equal(np_(b[2]), a[2]) and relmax(np_(b[1]), a[1]) <= tol
                else:
                    wa = R.Wavelet((fam, {'dtype': dtype}))
                    Tx, Wx, sf, sc = R.ssq_cwt(x, wa, nv=nv)
                    Tf, scl = (Tx, sf) if rng.random() < 0.5 else (Wx, sc)
                    kw = dict(penalty=float(rng.choice([0.5, 2.0, 20.0])), n_ridges=int(rng.integers(1, 4)),
                              bw=int(rng.choice([2, 4, 15])))
                    desc = f'ridges {dtype} {fam} N={N} {kw}'
                    a = R.extract_ridges(Tf, scl, parallel=False, **kw)
                    b = S.extract_ridges(Tf, scl, **kw)
                    same = (a == b)
                    ok = same.mean() >= 0.97 and same[:, 0].mean() >= 0.99
                    info = 'identical %.4f (first ridge %.4f)' % (same.mean(), same[:, 0].mean())
            except Exception as e:
                # a failing call must fail on both sides (in every branch the reference runs first)
                import traceback
                ours = any('ssqueezepy_amd' in f.filename for f in traceback.extract_tb(e.__traceback__))
                if not ours:
                    if kind in ('inverse', 'ridges', 'extras'):
                        ok, info = True, 'reference raised while preparing inputs (%s): skipped' % type(e).__name__
                    else:
                        try:
                            fn = {'cwt': S.cwt, 'ssq_cwt': S.ssq_cwt}.get(kind)
                            if fn is not None:
                                fn(x, S.Wavelet((fam, {'dtype': dtype})), **kw)
                            else:
                                S.ssq_stft(x, **kw)
                            ok, info = False, 'ONLY THE REFERENCE RAISED: %r' % (e,)
                        except Exception as e2:
                            ok, info = True, 'both raise (%s / %s)' % (type(e).__name__, type(e2).__name__)
                elif 'frequency range of the transform' in str(e):
                    # documented deviation: the reference returns NaN ssq_freqs
                    try:
                        bad = not np.all(np.isfinite(np_(a[2])))
                    except Exception:
                        bad = False
                    ok, info = bad, 'degenerate range: ours raises, the reference returns NaN ssq_freqs'
                else:
                    ok, info = False, 'ONLY OURS RAISED: %r' % (e,)
            print('%3d %-8s %s | %s | %s' % (case, 'OK' if ok else 'MISMATCH', desc, info, ''), flush=True)
            bad += not ok
    print('%d cases, %d mismatches' % (n_cases, bad))
    return bad


if __name__ == '__main__':
    sys.exit(1 if main(*(int(a) for a in sys.argv[1:3])) else 0)
