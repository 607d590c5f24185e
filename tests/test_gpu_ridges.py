# -*- coding: utf-8 -*-
"""Ridge extraction on the MI355X (csrc/ssq_ridge.hip, ssqueezepy_amd/ridge_extraction.py)
against the CPU oracle (oracle.ridge_track / extract_ridges, pinned to the reference's
ridge_extraction.py by tests/test_oracle_vs_golden.py) and the reference-generated fixture
tests/golden/ridges.npz.

The forward / backward recurrences are index work on rounded sums: compared bit for bit on
the same negative-log energy (the device's). The energy and its logarithm are floating
point: 1e-6 / 1e-13 relative (float32 / float64). End to end against the reference's
indices: float32 energies and logarithms follow NumPy's operation order (IEEE-exact), the
float64 logarithm is the device's and its last bit can move a weak ridge at isolated columns
(synchrosqueezed transforms are mostly exact zeros, i.e. ties): at least 99 % of the indices
identical, the dominant ridge identical."""
import os
import numpy as np
import pytest
from conftest import golden, two_chirps

DEV = 'cpu' if os.environ.get('SSQ_EMULATE') == '1' else 'cuda'    # see conftest.compute_module

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(180)]


@pytest.fixture(scope='module')
def S():
    from conftest import compute_module
    yield from compute_module()


def _stages(Tf, sc, penalty, eps, penalty_f32):
    """energy, E, pe, ridge of one tracking pass through the C ABI."""
    import torch
    from ssqueezepy_amd import _lib
    from ssqueezepy_amd._lib import check, F32, F64
    lib = _lib.load()
    Tf = torch.as_tensor(Tf, device=DEV).contiguous()
    f64 = Tf.dtype in (torch.complex128, torch.float64)
    rdt, code = (torch.float64, F64) if f64 else (torch.float32, F32)
    na, n = Tf.shape
    en = torch.empty((na, n), dtype=rdt, device=DEV)
    E, pe = torch.empty_like(en), torch.empty_like(en)
    ridge = torch.empty(n, dtype=torch.int64, device=DEV)
    scd = torch.as_tensor(np.ascontiguousarray(sc), device=DEV)
    check(lib.ssq_ridge_energy(code, int(Tf.is_complex()), Tf.data_ptr(), en.data_ptr(), na, n, None))
    check(lib.ssq_ridge_neglog(code, en.data_ptr(), E.data_ptr(), float(eps), na, n, None))
    check(lib.ssq_ridge_track(code, int(penalty_f32), E.data_ptr(), pe.data_ptr(), scd.data_ptr(),
                              float(penalty), float(eps), na, n, ridge.data_ptr(), None))
    if DEV == 'cuda':
        torch.cuda.synchronize()
    return en.cpu().numpy(), E.cpu().numpy(), pe.cpu().numpy(), ridge.cpu().numpy()


def _random_tf(rng, na, n, cdtype):
    """A noisy two-ridge magnitude pattern with random phases."""
    t = np.arange(n) / n
    rows = np.arange(na)[:, None]
    c1 = na * (0.2 + 0.5 * t)[None]
    c2 = na * (0.8 - 0.3 * t**2)[None]
    mag = np.exp(-0.5 * ((rows - c1) / 2.5)**2) + 0.6 * np.exp(-0.5 * ((rows - c2) / 2.0)**2)
    mag = mag + 0.05 * rng.random((na, n))
    return (mag * np.exp(2j * np.pi * rng.random((na, n)))).astype(cdtype)


@pytest.mark.parametrize('cdtype,na,n,transform', [
    ('complex64', 300, 3001, 'cwt'), ('complex64', 65, 384, 'stft'), ('complex64', 7, 50, 'cwt'),
    ('complex64', 513, 700, 'stft'), ('complex64', 1100, 150, 'cwt'),
    ('complex128', 300, 1500, 'cwt'), ('complex128', 552, 260, 'stft'),
    ('float64', 40, 333, 'cwt'), ('float32', 129, 1000, 'stft'), ('complex64', 300, 1, 'cwt'),
    ('complex64', 3, 2, 'stft')])
def test_tracking_stages_vs_oracle(orc, S, cdtype, na, n, transform):
    rng = np.random.default_rng(na * 7 + n)
    Tf = _random_tf(rng, na, n, cdtype)
    if not np.iscomplexobj(np.zeros(1, cdtype)):
        Tf = np.abs(_random_tf(rng, na, n, 'complex128')).astype(cdtype)
    scales = (np.exp(np.linspace(-0.3, 6.2, na)) if transform == 'cwt'
              else np.linspace(0, .5, na))
    penalty = 2.0 if transform == 'cwt' else 40.0
    pdt, eps, P = orc.ridge_design(Tf.dtype, scales, penalty, transform)
    sc = np.asarray(scales, dtype=pdt)
    sc = np.log(sc) if transform == 'cwt' else sc
    en, E, pe, ridge = _stages(Tf, sc, penalty, eps, pdt == np.float32)
    # floating point stages
    en_ref = np.abs(Tf)**2
    E_ref = -np.log(en_ref / en_ref.max(axis=0) + eps)
    tol = 1e-6 if en.dtype == np.float32 else 1e-13
    assert en.dtype == en_ref.dtype and E.dtype == E_ref.dtype
    assert np.abs(en - en_ref).max() <= tol * np.abs(en_ref).max()
    assert np.abs(E - E_ref).max() <= 4 * tol * max(1., np.abs(E_ref).max())
    # index work: bit for bit on the device's E
    ridge_ref, pe_ref = orc.ridge_track(E, P.reshape(na, na), eps)
    assert np.array_equal(pe, pe_ref)
    assert np.array_equal(ridge, ridge_ref)


def _golden_cases(g):
    names = sorted({k.rsplit('/', 1)[0] for k in g.files if '/' in k})
    for k in names:
        if k == 'basic':
            yield k, dict(penalty=2.0, bw=15, transform='cwt', n_ridges=1)
        else:
            a = g[k + '/args']
            yield k, dict(penalty=a[0], bw=int(a[1]), transform=('cwt', 'stft')[int(a[2])],
                          n_ridges=2)


def test_extract_ridges_vs_reference(S, orc):
    """The reference's outputs on its own transforms (tests/golden/ridges.npz)."""
    import torch
    g = golden('ridges')
    for k, kw in _golden_cases(g):
        Tf, sc = g[k + '/Tf'], g[k + '/scales']
        ri, rf, re = S.extract_ridges(Tf, sc, get_params=True, **kw)
        ref_i, ref_f, ref_e = g[k + '/idx'], g[k + '/f'], g[k + '/e']
        assert isinstance(ri, np.ndarray) and ri.shape == ref_i.shape, k
        assert rf.dtype == ref_f.dtype and re.dtype == ref_e.dtype, k
        same = (ri == ref_i)
        assert same.mean() >= 0.99, (k, same.mean())
        assert same[:, 0].all(), k                    # the dominant ridge: identical
        assert np.array_equal(rf[same], ref_f[same]), k
        tol = 1e-6 if re.dtype == np.float32 else 1e-13
        assert np.abs(re[same] - ref_e[same]).max() <= tol * np.abs(ref_e).max(), k
        # tensors in -> tensors out, same values
        ti = S.extract_ridges(torch.as_tensor(np.asarray(Tf, dtype=Tf.dtype if Tf.dtype.kind in 'fc'
                                                         else np.float64), device=DEV),
                              sc, **kw)
        assert isinstance(ti, torch.Tensor) and np.array_equal(ti.cpu().numpy(), ri), k


def test_basic_example(S):
    """tests/ridge_extraction_test.py:17-26."""
    tm = np.array([[1, 4, 4], [2, 2, 2], [5, 5, 4]])
    ridge_idxs, *_ = S.extract_ridges(tm, np.exp([1, 2, 3]), penalty=2.0, get_params=True)
    assert np.allclose(ridge_idxs, np.array([[2, 2, 2]]))


def test_on_device_transforms(S):
    """ssq_cwt / ssq_stft output -> ridges without leaving the device
    (tests/ridge_extraction_test.py:66-88 runs the same pipeline for coverage): tensors in,
    tensors out; `ridge_f` / `ridge_e` are the scales / energies at the returned indices."""
    import torch
    N = 4096
    x = torch.as_tensor(two_chirps(N, 3, noise=0.01), device=DEV)
    Tx, Wx, ssq_freqs, scales = S.ssq_cwt(x, 'gmw')
    Ts, Sx, sf, Sfs = S.ssq_stft(x, n_fft=256)
    for Tf, fr, kw in ((Tx, ssq_freqs, dict(penalty=2.0, bw=4, transform='cwt')),
                       (Ts, sf, dict(penalty=20.0, bw=4, transform='stft'))):
        ri, rf, re = S.extract_ridges(Tf, fr, n_ridges=2, get_params=True, **kw)
        assert all(isinstance(a, torch.Tensor) and a.is_cuda for a in (ri, rf, re))
        na = Tf.shape[0]
        assert tuple(ri.shape) == (N, 2) and ri.dtype == torch.int64
        assert int(ri.min()) >= 0 and int(ri.max()) < na
        frn = np.asarray(fr.cpu() if hasattr(fr, 'cpu') else fr, dtype=np.float32).reshape(-1)
        assert np.array_equal(rf.cpu().numpy(), frn[ri.cpu().numpy()])
        en = (torch.abs(Tf)**2).cpu().numpy()
        e0 = en[ri[:, 0].cpu().numpy(), np.arange(N)]          # first ridge: nothing cleared yet
        assert np.abs(re[:, 0].cpu().numpy() - e0).max() <= 1e-6 * en.max()
        # the first ridge follows energy: well above the mean energy of its column
        assert (e0 > en.mean(axis=0)).mean() > 0.95
        assert (ri[:, 0] != ri[:, 1]).float().mean() > 0.99   # the +-bw band was cleared


def test_batch_of_transforms(S):
    """A 3D input (batch of transforms of one scale vector; not in the reference) == the 2D call per
    transform: indices, frequencies, energies; tensors and arrays, CWT and STFT forms."""
    import torch
    N = 3000
    xb = np.stack([two_chirps(N, s, noise=0.02) for s in (1, 2, 3)])
    Tb, _, ssq_freqs, _ = S.ssq_cwt(xb, 'gmw')
    assert Tb.ndim == 3
    out = S.extract_ridges(Tb, ssq_freqs, penalty=2.0, n_ridges=2, bw=4, get_params=True)
    assert tuple(out[0].shape) == (3, N, 2) and all(isinstance(a, torch.Tensor) for a in out)
    for b in range(3):
        one = S.extract_ridges(Tb[b], ssq_freqs, penalty=2.0, n_ridges=2, bw=4, get_params=True)
        for A, a in zip(out, one):
            assert torch.equal(A[b], a), b
    Sb = S.ssq_stft(xb, n_fft=128, astensor=False)
    ri = S.extract_ridges(Sb[0], Sb[2], penalty=20.0, transform='stft')
    assert isinstance(ri, np.ndarray) and ri.shape == (3, Sb[0].shape[-1], 1)
    for b in range(3):
        assert np.array_equal(ri[b], S.extract_ridges(Sb[0][b], Sb[2], penalty=20.0, transform='stft'))


def test_argument_checks(S):
    with pytest.raises(ValueError):
        S.extract_ridges(np.zeros((4, 8, 2)), np.arange(1, 5.))
    with pytest.raises(ValueError):
        S.extract_ridges(np.ones((4, 8), dtype='complex64'), np.arange(1, 4.))
    with pytest.raises(ValueError):
        S.extract_ridges(np.ones((4, 8), dtype='complex64'), np.arange(1, 5.), transform='dwt')
