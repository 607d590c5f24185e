# -*- coding: utf-8 -*-
"""The ridge-tracking kernels (csrc/ssq_ridge.hip) compiled for the host and run with one
OS thread per work-item (tests/emu/, tests/emu_backend.py): checks the workgroup geometry, the LDS layout, the
barriers and the index arithmetic against the CPU oracle where no GPU is available.
The numbers a GPU produces are checked by tests/test_gpu_ridges.py; this is the same source
through the same C entry points on host memory. CPU-only."""
import ctypes
import numpy as np
import pytest
import emu_backend


@pytest.fixture(scope='module')
def emu():
    if not emu_backend.available():                    # ext_vector_type needs clang
        pytest.skip("no clang++ under $ROCM_PATH/lib/llvm/bin")
    return ctypes.CDLL(emu_backend.build())


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.mark.parametrize('cdtype,na,n,transform', [
    ('complex64', 30, 40, 'cwt'), ('complex128', 30, 40, 'stft'), ('float64', 20, 35, 'cwt'),
    ('complex64', 300, 12, 'cwt'), ('complex128', 300, 9, 'cwt'), ('complex64', 1100, 3, 'stft'),
    ('complex64', 5, 1, 'cwt'),
    # register-resident penalty variants (F, C) = (4, 32), (4, 16), (4, 16) with float64 data
    ('complex64', 256, 6, 'cwt'), ('complex64', 129, 8, 'stft'), ('float64', 150, 5, 'cwt'),
    ('complex64', 320, 4, 'stft'), ('complex64', 321, 4, 'cwt')])
def test_emulated_kernels_vs_oracle(emu, orc, cdtype, na, n, transform):
    rng = np.random.default_rng(na + n)
    ridge_row = na * (0.3 + 0.4 * np.arange(n)[None] / n)
    mag = rng.random((na, n)) + 3 * np.exp(-0.5 * ((np.arange(na)[:, None] - ridge_row) / 2)**2)
    Tf = ((mag * np.exp(2j * np.pi * rng.random((na, n)))).astype(cdtype)
          if cdtype.startswith('c') else mag.astype(cdtype))
    scales = np.exp(np.linspace(-0.3, 6.2, na)) if transform == 'cwt' else np.linspace(0, .5, na)
    penalty = 2.0 if transform == 'cwt' else 40.0
    pdt, eps, P = orc.ridge_design(Tf.dtype, scales, penalty, transform)
    sc = np.asarray(scales, dtype=pdt)
    sc = np.ascontiguousarray(np.log(sc) if transform == 'cwt' else sc)
    f64 = Tf.dtype in (np.complex128, np.float64)
    rdt, code = (np.float64, 1) if f64 else (np.float32, 0)
    en = np.empty((na, n), rdt)
    E, pe = np.empty_like(en), np.empty_like(en)
    ridge = np.empty(n, np.int64)
    i64, dbl = ctypes.c_int64, ctypes.c_double
    assert emu.ssq_ridge_energy(code, int(np.iscomplexobj(Tf)), _p(Tf), _p(en), i64(na), i64(n),
                                None) == 0
    assert emu.ssq_ridge_neglog(code, _p(en), _p(E), dbl(float(eps)), i64(na), i64(n), None) == 0
    assert emu.ssq_ridge_track(code, int(pdt == np.float32), _p(E), _p(pe), _p(sc),
                               dbl(penalty), dbl(float(eps)), i64(na), i64(n), _p(ridge), None) == 0
    en_ref = np.abs(Tf)**2
    tol = 1e-6 if rdt == np.float32 else 1e-13
    assert np.abs(en - en_ref).max() <= tol * np.abs(en_ref).max()
    assert np.array_equal(en, en_ref)      # NumPy's |z| formula: m * sqrt(fma(r, r, 1)), then squared
    if rdt == np.float32:                  # NumPy's float32 log, operation for operation
        assert np.array_equal(E, -np.log(en_ref / en_ref.max(axis=0) + eps))
    ridge_ref, pe_ref = orc.ridge_track(E, P.reshape(na, na), eps)
    assert np.array_equal(pe, pe_ref)
    assert np.array_equal(ridge, ridge_ref)
    # zeroing of the band around the ridge, Python slice rules (ridge_extraction.py:147-150)
    for bw in (4, 2.5, 40):
        en2, ref = en.copy(), en.copy()
        r_e = np.empty(n, rdt)
        assert emu.ssq_ridge_clear(code, _p(en2), _p(ridge), dbl(bw), _p(r_e), i64(na), i64(n),
                                   None) == 0
        for t in range(n):
            ref[int(ridge[t] - bw):int(ridge[t] + bw), t] = 0
        assert np.array_equal(en2, ref)
        assert np.array_equal(r_e, en[ridge, np.arange(n)])


@pytest.mark.parametrize('cdtype,na,n', [('complex64', 130, 37), ('complex128', 40, 70)])
def test_emulated_batch_entry_points(emu, cdtype, na, n):
    """ssq_ridge_{neglog,track,clear}_batch over three transforms == the single-transform entry points on each
    (one workgroup per transform in the tracking passes, blockIdx.y elsewhere)."""
    B = 3
    rng = np.random.default_rng(na)
    mag = rng.random((B, na, n)) + 3 * np.exp(-0.5 * ((np.arange(na)[None, :, None] - na * 0.5
                                                         - 5 * np.arange(B)[:, None, None]) / 2)**2)
    Tf = (mag * np.exp(2j * np.pi * rng.random((B, na, n)))).astype(cdtype)
    f64 = cdtype == 'complex128'
    rdt, code, pdt = (np.float64, 1, np.float64) if f64 else (np.float32, 0, np.float32)
    eps = np.finfo(pdt).eps
    sc = np.ascontiguousarray(np.log(np.exp(np.linspace(-0.3, 6.2, na))).astype(pdt))
    i64, dbl = ctypes.c_int64, ctypes.c_double
    en = np.empty((B, na, n), rdt)
    assert emu.ssq_ridge_energy(code, 1, _p(Tf), _p(en), i64(B * na), i64(n), None) == 0
    Eb, peb, rb = np.empty_like(en), np.empty_like(en), np.empty((B, n), np.int64)
    assert emu.ssq_ridge_neglog_batch(code, _p(en), _p(Eb), dbl(float(eps)), i64(na), i64(n), i64(B), None) == 0
    assert emu.ssq_ridge_track_batch(code, int(not f64), _p(Eb), _p(peb), _p(sc), dbl(2.0), dbl(float(eps)),
                                     i64(na), i64(n), _p(rb), i64(B), None) == 0
    enb, reb = en.copy(), np.empty((B, n), rdt)
    assert emu.ssq_ridge_clear_batch(code, _p(enb), _p(rb), dbl(4), _p(reb), i64(na), i64(n), i64(B), None) == 0
    for b in range(B):
        e1 = np.ascontiguousarray(en[b])
        E1, pe1, r1 = np.empty_like(e1), np.empty_like(e1), np.empty(n, np.int64)
        assert emu.ssq_ridge_neglog(code, _p(e1), _p(E1), dbl(float(eps)), i64(na), i64(n), None) == 0
        assert emu.ssq_ridge_track(code, int(not f64), _p(E1), _p(pe1), _p(sc), dbl(2.0), dbl(float(eps)),
                                   i64(na), i64(n), _p(r1), None) == 0
        re1 = np.empty(n, rdt)
        assert emu.ssq_ridge_clear(code, _p(e1), _p(r1), dbl(4), _p(re1), i64(na), i64(n), None) == 0
        assert np.array_equal(Eb[b], E1) and np.array_equal(peb[b], pe1) and np.array_equal(rb[b], r1)
        assert np.array_equal(enb[b], e1) and np.array_equal(reb[b], re1)


@pytest.mark.parametrize('as_numpy', [True, False])
def test_batches_beyond_one_launch_are_walked_in_chunks(as_numpy, monkeypatch):
    """`extract_ridges` on more transforms than one launch takes (MAX_BATCH_PER_LAUNCH, patched to 2 here) walks the
    batch in chunks: the same results as transform by transform, NumPy in -> NumPy out (the chunks are device tensors
    inside: joining them with np.concatenate raised for NumPy input -- round-5 advisor finding), tensor in -> tensor out,
    with and without `get_params`."""
    import emu_backend
    rng = np.random.default_rng(5)
    B, na, n = 5, 24, 40
    mag = rng.random((B, na, n)) + 3 * np.exp(-0.5 * ((np.arange(na)[None, :, None] - 8 - np.arange(B)[:, None, None]) / 2)**2)
    Tf = (mag * np.exp(2j * np.pi * rng.random((B, na, n)))).astype('complex64')
    scales = np.exp(np.linspace(0.1, 3.0, na)).astype('float32')
    with emu_backend.emulated() as S:
        import torch
        from ssqueezepy_amd import ridge_extraction as R
        monkeypatch.setattr(R, 'MAX_BATCH_PER_LAUNCH', 2)
        arg = Tf if as_numpy else torch.as_tensor(Tf)
        idx = R.extract_ridges(arg, scales, penalty=2., n_ridges=2, bw=3)
        idx_p, f_p, e_p = R.extract_ridges(arg, scales, penalty=2., n_ridges=2, bw=3, get_params=True)
        kind = np.ndarray if as_numpy else torch.Tensor
        assert isinstance(idx, kind) and isinstance(idx_p, kind) and isinstance(f_p, kind) and isinstance(e_p, kind)
        to_np = (lambda a: a) if as_numpy else (lambda a: a.cpu().numpy())
        assert to_np(idx).shape == (B, n, 2) and np.array_equal(to_np(idx), to_np(idx_p))
        for b in range(B):
            i1, f1, e1 = R.extract_ridges(Tf[b], scales, penalty=2., n_ridges=2, bw=3, get_params=True)
            assert np.array_equal(to_np(idx)[b], i1) and np.array_equal(to_np(f_p)[b], f1) and np.array_equal(to_np(e_p)[b], e1)
