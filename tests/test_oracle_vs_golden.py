# -*- coding: utf-8 -*-
"""Pins the CPU oracle (oracle/) against fixtures produced by the reference itself
(oracle/gen_golden.py; see tests/golden/). CPU-only.

Kernel level: bit-for-bit, both dtypes, in the typing mode the fixtures were
generated with (NumPy scalar rules -- numba is not installable, see
oracle/ssq_oracle.c). The numba-typed mode, which the HIP kernels implement, shares
every line of control flow; the tests below also bound how far apart the two modes
can be on the same inputs.
"""
import numpy as np
import pytest
from conftest import (golden, kernel_inputs, make_ssq_freqs, const_of, two_chirps)
from ssqueezepy_amd.ssqueezing import ssq_grid_params

NUMBA, NUMPY = 0, 1
DTYPES = ('float32', 'float64')


@pytest.mark.parametrize('dtype', DTYPES)
def test_phase_transforms_bitexact(orc, dtype):
    g = golden('kernels_' + dtype)
    na, n, gamma = int(g['na']), int(g['n']), float(g['gamma'])
    Wx, dWx, w, winf, Sfs, x = kernel_inputs(dtype, na, n)
    assert np.array_equal(orc.phase_cwt(Wx, dWx, gamma, typing=NUMPY),
                          g['phase_cwt'])
    assert np.array_equal(orc.phase_stft(Wx, dWx, Sfs, gamma, typing=NUMPY),
                          g['phase_stft'])
    # closed form of the reference's own test (tests/fft_test.py:159-161)
    ref = np.abs((dWx / Wx).imag / (2 * np.pi))
    ref[np.abs(Wx) < gamma] = np.inf
    out = orc.phase_cwt(Wx, dWx, gamma, typing=NUMBA)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(out), fin)
    assert np.allclose(out[fin], ref[fin], rtol=1e-5 if dtype == 'float32'
                       else 1e-13)


@pytest.mark.parametrize('dtype', DTYPES)
def test_ssqueeze_and_indexed_sum_bitexact(orc, dtype):
    g = golden('kernels_' + dtype)
    na, n, gamma = int(g['na']), int(g['n']), float(g['gamma'])
    Wx, dWx, w, winf, Sfs, x = kernel_inputs(dtype, na, n)
    checked = 0
    for st in ('log-piecewise', 'log', 'linear'):
        sf = make_ssq_freqs(na, st)
        _, p = ssq_grid_params(sf, st.startswith('log'))
        for flipud in (False, True):
            for ck in ('scalar', 'vec64', 'vecdt'):
                key = f'{st}/{int(flipud)}/{ck}'
                if 'ssq_cwt/' + key not in g:
                    continue
                const = const_of(ck, na, dtype)
                out = orc.ssqueeze(Wx, dWx, st, p, const, gamma, flipud,
                                   typing=NUMPY)
                assert np.array_equal(out, g['ssq_cwt/' + key]), key
                out = orc.indexed_sum(Wx, winf, st, p, const, flipud,
                                      typing=NUMPY)
                assert np.array_equal(out, g['isum/' + key]), key
                checked += 1
    assert checked == 12
    _, p = ssq_grid_params(Sfs, False)
    for flipud in (False, True):
        out = orc.ssqueeze(Wx, dWx, 'linear', p, Sfs[1] - Sfs[0], gamma, flipud,
                           Sfs=Sfs, typing=NUMPY)
        assert np.array_equal(out, g[f'ssq_stft/{int(flipud)}'])


@pytest.mark.parametrize('dtype', DTYPES)
def test_typing_modes_agree_up_to_bin_ties(orc, dtype):
    """numba-typed vs NumPy-typed arithmetic on identical inputs: the bin of a
    point may differ only where `w` sits on a bin edge to within float32
    rounding, and then by one bin; everything else is identical."""
    na, n, gamma = 100, 512, 1e-2
    Wx, dWx, *_ = kernel_inputs(dtype, na, n)
    for st in ('log-piecewise', 'log', 'linear'):
        sf = make_ssq_freqs(na, st)
        _, p = ssq_grid_params(sf, st.startswith('log'))
        _, k0 = orc.ssqueeze(Wx, dWx, st, p, 1., gamma, typing=NUMBA, get_k=True)
        _, k1 = orc.ssqueeze(Wx, dWx, st, p, 1., gamma, typing=NUMPY, get_k=True)
        diff = (k0 != k1)
        assert np.abs(k0 - k1).max() <= 1
        assert diff.mean() <= (2e-3 if dtype == 'float32' else 1e-4), diff.mean()
        assert np.array_equal(k0 < 0, k1 < 0)


@pytest.mark.parametrize('dtype', DTYPES)
def test_degenerate_points(orc, dtype):
    """Exact-zero derivative (w == 0 -> log2 = -inf): bin 0 before the flip on all
    grids (reference 'log' does this via max(., 0), algos.py:907; its un-jitted
    'log-piecewise' nest cannot run the case). Below-threshold points and
    non-finite `w` contribute nothing."""
    na, n, gamma = 16, 8, 1e-2
    cd = np.complex64 if dtype == 'float32' else np.complex128
    Wx = np.ones((na, n), cd) * (1 + 1j)
    dWx = np.zeros((na, n), cd)
    Wx[2, 3] = 0
    for st in ('log-piecewise', 'log', 'linear'):
        sf = make_ssq_freqs(na, st) + (1 if st == 'linear' else 0)
        _, p = ssq_grid_params(sf, st.startswith('log'))
        for flipud in (False, True):
            out, k = orc.ssqueeze(Wx, dWx, st, p, 1., gamma, flipud, get_k=True)
            row = na - 1 if flipud else 0
            assert k[2, 3] == -1
            assert (np.delete(k.ravel(), 2 * n + 3) == row).all()
            assert out[row, 0] == na * (1 + 1j)
            assert out[row, 3] == (na - 1) * (1 + 1j)
            others = np.delete(out, row, axis=0)
            assert (others == 0).all()


@pytest.mark.parametrize('dtype', DTYPES)
def test_replace_and_buffer_bitexact(orc, dtype):
    g = golden('kernels_' + dtype)
    na, n = int(g['na']), int(g['n'])
    Wx, dWx, w, winf, Sfs, x = kernel_inputs(dtype, na, n)
    assert np.array_equal(orc.replace_under_abs(w, Wx, 1.5, np.inf),
                          g['replace_under_abs'])
    for seg, ov in ((128, 96), (127, 100), (64, 0), (33, 32)):
        for mod in (False, True):
            out = orc.buffer(x, seg, ov, mod)
            assert np.array_equal(out, g[f'buffer/{seg}/{ov}/{int(mod)}'])
            if mod:     # modulated framing == ifftshift of each frame
                plain = orc.buffer(x, seg, ov, False)
                assert np.array_equal(out, np.fft.ifftshift(plain, axes=0))


def _ridge_cases(g):
    names = sorted({k.rsplit('/', 1)[0] for k in g.files if '/' in k})
    for k in names:
        if k == 'basic':
            yield k, dict(penalty=2.0, bw=15, transform='cwt', n_ridges=1)
        else:
            a = g[k + '/args']
            yield k, dict(penalty=a[0], bw=int(a[1]), transform=('cwt', 'stft')[int(a[2])],
                          n_ridges=2)


def test_ridge_extraction_bitexact(orc):
    """extract_ridges (ridge_extraction.py:11-232) on the reference's own transforms and
    its 3x3 example (tests/ridge_extraction_test.py:17-26): indices, scales and energies."""
    g = golden('ridges')
    n_cases = 0
    for k, kw in _ridge_cases(g):
        ri, rf, re = orc.extract_ridges(g[k + '/Tf'], g[k + '/scales'], get_params=True, **kw)
        assert np.array_equal(ri, g[k + '/idx']), k
        assert np.array_equal(rf, g[k + '/f']), k
        assert np.array_equal(re, g[k + '/e']), k
        n_cases += 1
    assert n_cases == 9
