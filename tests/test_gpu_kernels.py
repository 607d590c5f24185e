# -*- coding: utf-8 -*-
"""Kernel-level parity on the MI355X: every C-ABI kernel entry point
(include/ssq_hip.h, called through ssqueezepy_amd.algos) against the CPU oracle on
identical inputs. Bin indices (integer work) must match exactly; float sums follow
the reference's summation order and are compared exactly where the arithmetic is
IEEE-reproducible, else within the tolerance stated at the assertion.
Modelled on the reference's tests/fft_test.py:141-415.
"""
import numpy as np
import pytest
from conftest import (golden, kernel_inputs, make_ssq_freqs, const_of)

pytestmark = pytest.mark.gpu
NUMBA, NUMPY = 0, 1
DTYPES = ('float32', 'float64')


@pytest.fixture(scope='module')
def A():
    from conftest import compute_module
    for mod in compute_module():
        from ssqueezepy_amd import algos
        yield algos


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('dtype', DTYPES)
def test_phase_cwt_and_stft(A, orc, dtype):
    na, n, gamma = 100, 1028, 1e-2
    Wx, dWx, w, winf, Sfs, x = kernel_inputs(dtype, na, n)
    out = _np(A.phase_cwt_gpu(Wx, dWx, gamma))
    ref = orc.phase_cwt(Wx, dWx, gamma, typing=NUMBA)
    assert np.array_equal(out, ref)
    out = _np(A.phase_stft_gpu(Wx, dWx, Sfs, gamma))
    ref = orc.phase_stft(Wx, dWx, Sfs, gamma, typing=NUMBA)
    assert np.array_equal(out, ref)
    # closed form, reference tolerance (tests/fft_test.py:159-174: np.allclose)
    cf = np.abs((dWx / Wx).imag / (2 * np.pi))
    cf[np.abs(Wx) < gamma] = np.inf
    assert np.allclose(_np(A.phase_cwt_gpu(Wx, dWx, gamma)), cf)


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(100, 512), (300, 1000), (513, 77), (37, 16),
                                   (1500, 40), (6000, 24), (64, 60000)])
def test_ssqueeze_fast_vs_oracle(A, orc, dtype, shape):
    na, n = shape
    gamma = 1e-2
    Wx, dWx, *_ = kernel_inputs(dtype, max(na, 12), max(n, 12))
    Wx, dWx = Wx[:na, :n].copy(), dWx[:na, :n].copy()
    dWx[5, 3] = 0                      # exact-zero derivative -> bin 0 pre-flip
    kinds = ('scalar', 'vec64', 'vecdt') if na <= 513 else ('scalar',)
    for st in ('log-piecewise', 'log', 'linear'):
        sf = make_ssq_freqs(na, st)
        logscale = st.startswith('log')
        from ssqueezepy_amd.ssqueezing import ssq_grid_params
        _, p = ssq_grid_params(sf, logscale)
        for flipud in (False, True):
            for ck in kinds:
                const = const_of(ck, na, dtype)
                out, k = A.ssqueeze_fast(Wx, dWx, sf, const, logscale, flipud,
                                         gamma, get_k=True)
                ref, kref = orc.ssqueeze(Wx, dWx, st, p, const, gamma, flipud,
                                         typing=NUMBA, get_k=True)
                assert np.array_equal(_np(k), kref), (st, flipud, ck)   # index: exact
                assert np.array_equal(_np(out), ref), (st, flipud, ck)  # sums: same order


@pytest.mark.parametrize('dtype', DTYPES)
def test_ssqueeze_fast_vs_reference_golden(A, dtype):
    """Against the reference's own outputs (NumPy-typed run, see oracle/): identical
    except at float32 bin-edge ties; column sums are assignment-invariant."""
    g = golden('kernels_' + dtype)
    na, n, gamma = int(g['na']), int(g['n']), float(g['gamma'])
    Wx, dWx, w, winf, Sfs, x = kernel_inputs(dtype, na, n)
    for st in ('log-piecewise', 'log', 'linear'):
        sf = make_ssq_freqs(na, st)
        for flipud in (False, True):
            out = _np(A.ssqueeze_fast(Wx, dWx, sf, const_of('scalar', na, dtype),
                                      st.startswith('log'), flipud, gamma))
            ref = g[f'ssq_cwt/{st}/{int(flipud)}/scalar']
            differing = (np.abs(out - ref) > 1e-6 * np.abs(ref).max()).mean()
            assert differing <= (5e-3 if dtype == 'float32' else 1e-3), differing
            assert np.allclose(out.sum(0), ref.sum(0), rtol=0,
                               atol=(2e-6 if dtype == 'float32' else 1e-13)
                               * np.abs(ref.sum(0)).max())
    for flipud in (False, True):
        out = _np(A.ssqueeze_fast(Wx, dWx, Sfs, Sfs[1] - Sfs[0], False, flipud,
                                  gamma, Sfs=Sfs))
        ref = g[f'ssq_stft/{int(flipud)}']
        assert (np.abs(out - ref) > 1e-6 * np.abs(ref).max()).mean() <= 5e-3
        assert np.allclose(out.sum(0), ref.sum(0), rtol=0,
                           atol=(2e-6 if dtype == 'float32' else 1e-13)
                           * np.abs(ref.sum(0)).max())


@pytest.mark.parametrize('dtype', DTYPES)
def test_ssqueeze_stft_form_and_batch(A, orc, dtype):
    na, n, gamma = 129, 300, 1e-2
    Wx, dWx, w, winf, Sfs, x = kernel_inputs(dtype, na, n)
    from ssqueezepy_amd.ssqueezing import ssq_grid_params
    _, p = ssq_grid_params(Sfs, False)
    for flipud in (False, True):
        out, k = A.ssqueeze_fast(Wx, dWx, Sfs, Sfs[1] - Sfs[0], False, flipud,
                                 gamma, Sfs=Sfs, get_k=True)
        ref, kref = orc.ssqueeze(Wx, dWx, 'linear', p, Sfs[1] - Sfs[0], gamma,
                                 flipud, Sfs=Sfs, typing=NUMBA, get_k=True)
        assert np.array_equal(_np(k), kref)
        assert np.array_equal(_np(out), ref)
    # batched == looped (reference: tests/fft_test.py:559-631)
    Wb = np.stack([Wx, Wx[::-1].copy(), 2 * Wx])
    dWb = np.stack([dWx, dWx[::-1].copy(), dWx])
    sf = make_ssq_freqs(na, 'log')
    outb = _np(A.ssqueeze_fast(Wb, dWb, sf, 0.5, True, True, gamma))
    for b in range(3):
        assert np.array_equal(outb[b], _np(A.ssqueeze_fast(Wb[b], dWb[b], sf, 0.5,
                                                           True, True, gamma)))


@pytest.mark.parametrize('dtype', DTYPES)
def test_indexed_sum_onfly(A, orc, dtype):
    na, n = 100, 512
    Wx, dWx, w, winf, Sfs, x = kernel_inputs(dtype, na, n)
    from ssqueezepy_amd.ssqueezing import ssq_grid_params
    for st in ('log-piecewise', 'log', 'linear'):
        sf = make_ssq_freqs(na, st)
        _, p = ssq_grid_params(sf, st.startswith('log'))
        for flipud in (False, True):
            for ck in ('scalar', 'vec64'):
                const = const_of(ck, na, dtype)
                out = _np(A.indexed_sum_onfly(Wx, winf, sf, const,
                                              st.startswith('log'), flipud))
                ref = orc.indexed_sum(Wx, winf, st, p, const, flipud, typing=NUMBA)
                if dtype == 'float64' or st == 'linear':
                    assert np.array_equal(out, ref), (st, flipud, ck)
                else:
                    # float32 log grids: log2f (device libm vs glibc) may differ by
                    # an ulp, which moves a point only when it sits on a bin edge;
                    # reference tolerance (tests/fft_test.py:277-281): mean abs
                    # diff < 1e-8
                    assert np.abs(out - ref).mean() < 1e-8, (st, flipud, ck)
                    assert (out != ref).mean() < 1e-3


@pytest.mark.parametrize('dtype', DTYPES)
def test_replace_buffer_pad(A, orc, dtype):
    import torch
    g = golden('kernels_' + dtype)
    na, n = int(g['na']), int(g['n'])
    Wx, dWx, w, winf, Sfs, x = kernel_inputs(dtype, na, n)
    wt = A.to_device(w.copy())
    A.replace_under_abs(wt, Wx, 1.5, np.inf)
    assert np.array_equal(_np(wt), g['replace_under_abs'])
    for seg, ov in ((128, 96), (127, 100), (64, 0), (33, 32)):
        for mod in (False, True):
            out = _np(A.buffer(x, seg, ov, mod))
            assert np.array_equal(out, g[f'buffer/{seg}/{ov}/{int(mod)}'])
    xb = np.stack([x, x[::-1].copy()])
    outb = _np(A.buffer(xb, 100, 60, True))
    assert np.array_equal(outb[1], orc.buffer(xb[1], 100, 60, True))
    from ssqueezepy_amd.padding import padsignal
    for pt in ('reflect', 'symmetric', 'replicate', 'wrap', 'zero'):
        for n1, n2 in ((12, 12), (5, 4), (0, 7)):
            ref = np.pad(x[:50], (n1, n2), mode={'zero': 'constant', 'reflect': 'reflect',
                                                'symmetric': 'symmetric', 'replicate': 'edge',
                                                'wrap': 'wrap'}[pt])
            assert np.array_equal(_np(A.pad_signal_gpu(x[:50], n1, n2, pt)), ref)
    # reflect pad longer than the signal (numpy repeats the reflection)
    ref = np.pad(x[:10], (23, 17), mode='reflect')
    assert np.array_equal(_np(A.pad_signal_gpu(x[:10], 23, 17, 'reflect')), ref)


def test_error_paths(A):
    import torch
    from ssqueezepy_amd._lib import SsqError
    Wx = np.ones((4, 8), np.complex64)
    with pytest.raises(ValueError):
        A.ssqueeze_fast(Wx, Wx, np.linspace(1, 2, 4), 1., False, False, None)
    with pytest.raises(ValueError):
        A.ssqueeze_fast(Wx, Wx[:, :4], np.linspace(1, 2, 4), 1., False, False, 1e-3)
    with pytest.raises(TypeError):
        A.ssqueeze_fast(Wx.real.copy(), Wx, np.linspace(1, 2, 4), 1., False, False, 1e-3)
    from ssqueezepy_amd import _lib
    lib = _lib.load()
    rc = lib.ssq_pad_signal(7, None, None, 1, 4, 1, 1, 1, None)
    assert rc != 0 and b'dtype' in lib.ssq_last_error()


def test_quad_variant_of_accumulate_is_bit_identical(orc):
    """SSQ_DEBUG_ACC_VARIANT=2 selects the one-wave-per-tile (DPP quad) build of the
    reassignment kernel; it must produce the same bits. Run in a subprocess because the
    variant is latched at first launch."""
    import subprocess, sys, os
    if os.environ.get('SSQ_EMULATE') == '1':
        pytest.skip("runs a subprocess against the real library")
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import kernel_inputs, make_ssq_freqs
from oracle import oracle as orc
from ssqueezepy_amd import algos as A
from ssqueezepy_amd.ssqueezing import ssq_grid_params
for dtype in ('float32', 'float64'):
    Wx, dWx, *_ = kernel_inputs(dtype, 300, 1000)
    for st in ('log-piecewise', 'log', 'linear'):
        sf = make_ssq_freqs(300, st)
        _, p = ssq_grid_params(sf, st.startswith('log'))
        out = A.ssqueeze_fast(Wx, dWx, sf, 0.02, st.startswith('log'), True, 1e-2).cpu().numpy()
        ref = orc.ssqueeze(Wx, dWx, st, p, 0.02, 1e-2, True, typing=0)
        assert np.array_equal(out, ref), (dtype, st)
print("QUAD_OK")
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
       os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SSQ_DEBUG_ACC_VARIANT='2')
    res = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True,
                         timeout=600)
    assert 'QUAD_OK' in res.stdout, res.stdout + res.stderr


@pytest.mark.parametrize('st', ['log', 'log-piecewise', 'linear'])
def test_bin_screening_stress_float64(A, orc, st):
    """Round 6: float64 data goes through the float32 screen too (inputs rounded to float32, an explicit bound on the
    estimate's error; `bin_of_point(double ...)`, csrc/ssq_point_math.inl) -- the exact double map only inside the guard
    band. 3M points that sweep the grid and both ends densely, a tenth of them placed within 1e-3 ... 1e-10 bins of the rounding
    boundaries k + 1/2 on either side, and points whose numerator b c - a d cancels to a
    sliver of its terms: every index must equal the CPU path's exact map."""
    from ssqueezepy_amd.ssqueezing import ssq_grid_params
    na, n = 300, 10000
    rng = np.random.default_rng(321)
    logscale = st.startswith('log')
    if st == 'linear':
        sf = np.linspace(1e-3, 0.5, na)
        wt = rng.uniform(-0.02, 0.55, (na, n))
    else:
        sf = 0.5 * 2.0 ** (-(np.arange(na)[::-1]) / 32.0) if st == 'log' else make_ssq_freqs(na, st)
        lo, hi = np.log2(sf[0]) - 1, np.log2(sf[-1]) + 1
        wt = 2.0 ** rng.uniform(lo, hi, (na, n))
    # boundary points: the grid's own half-way marks (in the map's coordinate) and their close neighbourhoods
    kk = rng.integers(0, na - 1, (na, n // 10)).astype(np.float64)
    # (not ON them: at k + 1/2 exactly the CPU path's own answer hangs on its libm's last bit)
    dl = rng.choice([1e-10, -1e-10, 1e-9, -1e-9, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3],
                    (na, n // 10))
    t = kk + 0.5 + dl
    if st == 'linear':
        wb = sf[0] + t * (sf[1] - sf[0])
    elif st == 'log':
        wb = 2.0 ** (np.log2(sf[0]) + t * (np.log2(sf[1]) - np.log2(sf[0])))
    else:
        i0 = np.clip(np.floor(t).astype(int), 0, na - 2)      # (between neighbours of the piecewise grid, log-linearly)
        fr = t - i0
        wb = 2.0 ** (np.log2(sf[i0]) * (1 - fr) + np.log2(sf[i0 + 1]) * fr)
    wt[:, :n // 10] = wb
    Wx = rng.standard_normal((na, n)) + 1j * rng.standard_normal((na, n))
    g = rng.standard_normal((na, n))
    g[:, n // 10:n // 5] *= 1e6                                 # a large real part: the imaginary part survives a cancellation
    dWx = Wx * (g + 2j * np.pi * wt)
    kind, p = ssq_grid_params(sf, logscale)
    const = np.log(2) / 32
    for flipud in (False, True):
        out, k = A.ssqueeze_fast(Wx, dWx, sf, const, logscale, flipud, 1e-3, get_k=True)
        ref, kref = orc.ssqueeze(Wx, dWx, st, p, const, 1e-3, flipud, typing=NUMBA, get_k=True)
        k = _np(k)
        bad = np.argwhere(k != kref)
        info = [(int(i), int(j), float(dl[i, j]) if j < n // 10 else None, int(k[i, j]), int(kref[i, j])) for i, j in bad[:12]]
        assert np.array_equal(k, kref), (st, flipud, int((k != kref).sum()), info)
        assert np.array_equal(_np(out), ref)


@pytest.mark.parametrize('st', ['log', 'log-piecewise', 'linear'])
def test_bin_screening_stress(A, orc, st):
    """float32 bin map under load: 6M points whose phase transform sweeps the whole
    grid (and beyond both ends) densely, so that many land within the float32
    screen's guard band of a rounding boundary. Every index must equal the exact
    double-precision map of the CPU path (index work: bit-exact)."""
    from ssqueezepy_amd.ssqueezing import ssq_grid_params
    na, n = 300, 20000
    rng = np.random.default_rng(123)
    if st == 'linear':
        sf = np.linspace(1e-3, 0.5, na)
        wt = rng.uniform(-0.02, 0.55, (na, n))
    else:
        if st == 'log':
            sf = 0.5 * 2.0 ** (-(np.arange(na)[::-1]) / 32.0)
        else:                    # two log segments of different density (reference's test grid)
            sf = make_ssq_freqs(na, st)
        lo, hi = np.log2(sf[0]) - 1, np.log2(sf[-1]) + 1
        wt = 2.0 ** rng.uniform(lo, hi, (na, n))
    Wx = (rng.standard_normal((na, n)) + 1j * rng.standard_normal((na, n))).astype('complex64')
    dWx = (Wx.astype('complex128') * (rng.standard_normal((na, n)) + 2j * np.pi * wt)
           ).astype('complex64')
    logscale = st.startswith('log')
    kind, p = ssq_grid_params(sf, logscale)
    assert kind == {'log': 0, 'log-piecewise': 1, 'linear': 2}[st]
    const = np.log(2) / 32
    for flipud in (False, True):
        out, k = A.ssqueeze_fast(Wx, dWx, sf, const, logscale, flipud, 1e-3, get_k=True)
        ref, kref = orc.ssqueeze(Wx, dWx, st, p, const, 1e-3, flipud, typing=NUMBA,
                                 get_k=True)
        k = _np(k)
        assert np.array_equal(k, kref), (st, flipud, int((k != kref).sum()))
        assert len(np.unique(kref)) > 0.9 * na          # the sweep really covers the grid
        assert np.array_equal(_np(out), ref)
