# -*- coding: utf-8 -*-
"""The kernel-level seam (csrc/ssq_kernels.hip: phase transforms, fused reassignment,
indexed sum, thresholding, framing, padding) compiled for the host and run with one OS thread
per work-item (tests/emu/: pthread barriers for __syncthreads, wavefront rendezvous for the
DPP moves / ballots), driven through the product's own host layer (ssqueezepy_amd/algos.py)
and compared bit for bit with the CPU oracle. It checks tile geometry, LDS layout, the
in-order fold and the index arithmetic where no GPU is available; what a GPU computes is
checked by tests/test_gpu_kernels.py. CPU-only."""
import numpy as np
import pytest
import emu_backend
from conftest import kernel_inputs, make_ssq_freqs, const_of

NUMBA = 0


@pytest.fixture(scope='module')
def A():
    """ssqueezepy_amd.algos bound to the emulated library, tensors on the host."""
    if not emu_backend.available():
        pytest.skip("no clang++ under $ROCM_PATH/lib/llvm/bin")
    with emu_backend.emulated():
        from ssqueezepy_amd import algos
        yield algos


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_phase_transforms(A, orc, dtype):
    na, n, gamma = 20, 70, 1e-2
    Wx, dWx, w, winf, Sfs, x = kernel_inputs(dtype, na, n)
    assert np.array_equal(_np(A.phase_cwt_gpu(Wx, dWx, gamma)), orc.phase_cwt(Wx, dWx, gamma, typing=NUMBA))
    assert np.array_equal(_np(A.phase_stft_gpu(Wx, dWx, Sfs, gamma)),
                          orc.phase_stft(Wx, dWx, Sfs, gamma, typing=NUMBA))


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
@pytest.mark.parametrize('shape', [(40, 70), (70, 20), (37, 16)])
def test_fused_reassignment_vs_oracle(A, orc, dtype, shape):
    from ssqueezepy_amd.ssqueezing import ssq_grid_params
    na, n = shape
    gamma = 1e-2
    Wx, dWx, *_ = kernel_inputs(dtype, na, n)
    dWx[5, 3] = 0                      # exact-zero derivative -> bin 0 pre-flip
    for st, flipud, ck in (('log-piecewise', True, 'vec64'), ('log', True, 'scalar'),
                           ('log', False, 'vecdt'), ('linear', False, 'scalar')):
        sf = make_ssq_freqs(na, st)
        logscale = st.startswith('log')
        _, p = ssq_grid_params(sf, logscale)
        const = const_of(ck, na, dtype)
        out, k = A.ssqueeze_fast(Wx, dWx, sf, const, logscale, flipud, gamma, get_k=True)
        ref, kref = orc.ssqueeze(Wx, dWx, st, p, const, gamma, flipud, typing=NUMBA, get_k=True)
        assert np.array_equal(_np(k), kref), (st, flipud, ck)   # index: exact
        assert np.array_equal(_np(out), ref), (st, flipud, ck)  # sums: same order


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_indexed_sum_and_helpers(A, orc, dtype):
    from ssqueezepy_amd.ssqueezing import ssq_grid_params
    na, n, gamma = 40, 50, 1e-2
    Wx, dWx, w, winf, Sfs, x = kernel_inputs(dtype, na, n)
    for st in ('log', 'linear'):
        sf = make_ssq_freqs(na, st)
        logscale = st.startswith('log')
        _, p = ssq_grid_params(sf, logscale)
        const = const_of('scalar', na, dtype)
        out = A.indexed_sum_onfly(Wx, winf, sf, const, logscale, True)
        ref = orc.indexed_sum(Wx, winf, st, p, const, True, typing=NUMBA)
        assert np.array_equal(_np(out), ref), st
    xb = x[:200].copy()
    assert np.array_equal(_np(A.buffer(xb, 32, 24, modulated=True)), orc.buffer(xb, 32, 24, True))
