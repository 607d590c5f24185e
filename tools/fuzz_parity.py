# -*- coding: utf-8 -*-
"""Randomised parity sweep (design aid): ssq_cwt / ssq_stft on the device against the CPU
oracle pipeline for random lengths, voices, wavelets, pad types, dtypes and batch sizes.
    python tools/fuzz_parity.py [n_cases] [seed]
(SSQ_EMULATE=1: against the CPU emulation of the kernels, tests/emu/, where there is no GPU.)
Prints one line per case; exits non-zero on the first mismatch."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ssqueezepy_amd as S
from oracle import oracle as orc
from pipeline import oracle_ssq_cwt, oracle_ssq_stft, GRIDNAME
from conftest import two_chirps


def tx_ok(Tx, ref):
    """bit for bit, or -- the default kernels' float64 sums in arrival order -- to the data type's rounding
    of the reference's running sums"""
    tol = 1e-6 if Tx.dtype == np.complex64 else 1e-13
    return np.array_equal(Tx, ref) or np.abs(Tx - ref).max() <= tol * np.abs(ref).max()


def relmax(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def main(n_cases=30, seed=0):
    rng = np.random.default_rng(seed)
    orc.lib()
    for case in range(n_cases):
        dtype = rng.choice(['float32', 'float64'])
        tol = 1e-5 if dtype == 'float32' else 1e-11
        if rng.random() < 0.7:
            N = int(rng.integers(200, 30000))
            nv = int(rng.choice([4, 8, 16]))
            fam = rng.choice(['gmw', 'morlet', 'bump', 'cmhat', 'hhhat'])
            pad = rng.choice(['reflect', 'zero', 'symmetric', 'wrap', 'replicate'])
            st = rng.choice(['log', 'log-piecewise', 'linear'])
            if st == 'linear':
                N = int(rng.integers(200, 1500))       # one row per sample spacing: keep small
            x = two_chirps(N, seed=case)
            wav = S.Wavelet((fam, {'dtype': dtype}))
            try:
                Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, wav, scales=st, nv=nv, padtype=pad,
                                                get_dWx=True, astensor=False)
            except ValueError as e:           # degenerate design (the reference fails too)
                print('cwt ', dtype, fam, st, pad, 'N=%d nv=%d' % (N, nv), 'SKIP:', str(e)[:60])
                continue
            r = oracle_ssq_cwt(orc, x, dtype, wavelet=fam, scales=st, nv=nv, padtype=pad)
            eW, eD = relmax(Wx, r['Wx']), relmax(dWx, r['dWx'])
            ref = orc.ssqueeze(Wx, dWx, GRIDNAME[r['grid']], r['params'], r['const'],
                               r['gamma'], True, typing=0)
            ok = eW <= tol and eD <= tol and tx_ok(Tx, ref) and \
                np.array_equal(sf, r['ssq_freqs'])
            if ok and rng.random() < 0.4:             # batched == single, get_w == oracle phase
                xb = np.stack([x, x[::-1].copy(), 2 * x])
                Tb, Wb, *_ = S.ssq_cwt(xb, wav, scales=st, nv=nv, padtype=pad, astensor=False)
                ok = np.array_equal(Wb[0], Wx) and tx_ok(Tb[0], Tx)
                # (the two-step form runs every row on the block kernels: its own Wx / dWx)
                out = S.ssq_cwt(x, wav, scales=st, nv=nv, padtype=pad, get_w=True, get_dWx=True,
                                astensor=False)
                ok = ok and np.array_equal(out[4], orc.phase_cwt(out[1], out[5], r['gamma'], typing=0))
                ok = ok and relmax(out[1], r['Wx']) <= tol
            print('cwt ', dtype, fam, st, pad, 'N=%d nv=%d na=%d' % (N, nv, len(sc)),
                  'eW=%.1e eD=%.1e' % (eW, eD), 'OK' if ok else 'MISMATCH')
        else:
            N = int(rng.integers(300, 20000))
            # (powers of two: the fused kernel; any other length: the mixed-radix fused kernel when its prime factors
            # are <= 31, else framing + rocFFT)
            n_fft = int(rng.choice([64, 100, 128, 256, 512, 1000, 1024])) if rng.random() < 0.4 else int(rng.integers(16, 1500))
            n_fft = min(n_fft, N // 2)
            hop = int(rng.integers(1, max(2, n_fft // 2)))
            mod = bool(rng.random() < 0.7)
            x = two_chirps(N, seed=case)
            Tx, Sx, sf, Sfs, dSx = S.ssq_stft(x, n_fft=n_fft, hop_len=hop, modulated=mod,
                                              dtype=dtype, get_dWx=True, astensor=False)
            ro = oracle_ssq_stft(orc, x, dtype, n_fft=n_fft, hop_len=hop, modulated=mod)
            eS, eD = relmax(Sx, ro['Sx']), relmax(dSx, ro['dSx'])
            from ssqueezepy_amd.ssqueezing import ssq_grid_params
            _, p = ssq_grid_params(Sfs, False)
            ref = orc.ssqueeze(Sx, dSx, 'linear', p, Sfs[1] - Sfs[0], ro['gamma'], False,
                               Sfs=Sfs, typing=0)
            ok = eS <= tol and eD <= tol and np.array_equal(Tx, ref)
            # without dSx the fused kernel hands the reassignment a 2-byte bin map instead
            T2, S2, *_ = S.ssq_stft(x, n_fft=n_fft, hop_len=hop, modulated=mod, dtype=dtype,
                                    astensor=False)
            ok = ok and tx_ok(T2, Tx) and np.array_equal(S2, Sx)
            print('stft', dtype, 'N=%d n_fft=%d hop=%d mod=%d' % (N, n_fft, hop, mod),
                  'eS=%.1e eD=%.1e' % (eS, eD), 'OK' if ok else 'MISMATCH')
        if not ok:
            sys.exit(1)
    print('all %d cases OK' % n_cases)


if __name__ == '__main__':
    if os.environ.get('SSQ_EMULATE') == '1':      # no GPU: the kernels under the CPU emulator
        import emu_backend
        with emu_backend.emulated():
            main(*(int(a) for a in sys.argv[1:3]))
    else:
        main(*(int(a) for a in sys.argv[1:3]))
