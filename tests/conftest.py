# -*- coding: utf-8 -*-
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line(
        "markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def compute_module():
    """Generator behind the `S` / `A` fixtures of the GPU test modules: the package bound to
    the in-tree HIP library on a real GPU -- or, with SSQ_EMULATE=1 (a development aid, run
    e.g. `SSQ_EMULATE=1 pytest tests -m gpu -k "not full_size"` in a container without a GPU),
    bound to the CPU emulation of the same kernels (tests/emu_backend.py)."""
    if os.environ.get('SSQ_EMULATE') == '1':
        import emu_backend
        with emu_backend.emulated() as mod:
            yield mod
        return
    import torch
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    import ssqueezepy_amd
    from ssqueezepy_amd import _lib
    _lib.load(build_if_missing=False)          # the in-tree HIP library must exist
    yield ssqueezepy_amd


def golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


def two_chirps(N, seed=0, noise=0.1):
    """Signal family of SURVEY.md section 8(d) (same formula as
    oracle/gen_golden.py:two_chirps)."""
    rng = np.random.default_rng(seed)
    f0 = rng.uniform(0.01, 0.05)
    f1 = rng.uniform(0.30, 0.45)
    t = np.arange(N) / N
    ph = f0 * N * t + 0.5 * (f1 - f0) * N * t**2
    return (np.cos(2 * np.pi * ph) + np.cos(2 * np.pi * (ph + 0.04 * N * t))
            + noise * rng.standard_normal(N))


def kernel_inputs(dtype, na, n):
    """Seeded inputs of the kernel-level fixtures (oracle/gen_golden.py:gen_kernels,
    modelled on the reference's tests/fft_test.py:284-315)."""
    np.random.seed(0)
    Wx = np.random.randn(na, n).astype(dtype) * (1 + 2j)
    dWx = np.random.randn(na, n).astype(dtype) * (2 - 1j)
    w = np.abs(np.random.randn(na, n).astype(dtype))
    w *= (2 * na / w.max())
    Sfs = np.linspace(0, .5, na).astype(dtype)
    Wx[3, 5] = 1e-4 * (1 + 1j)
    Wx[7, 9] = 0
    winf = w.copy()
    winf[2, 3] = np.inf
    x = np.random.randn(1000).astype(dtype)
    return Wx, dWx, w, winf, Sfs, x


def make_ssq_freqs(M, scaletype):
    # grids of the reference's kernel tests (tests/fft_test.py:236-246)
    if scaletype == 'log-piecewise':
        sf = np.logspace(0, np.log10(M), 2 * M)
        return np.hstack([sf[:M // 2], sf[M // 2 + 3 - 1::3]])
    elif scaletype == 'log':
        return np.logspace(0, np.log10(M), M)
    return np.linspace(0, M, M)


def const_of(kind, na, dtype):
    if kind == 'scalar':
        return np.log(2) / 32
    v = np.log(2) / np.linspace(8, 32, na)
    return v if kind == 'vec64' else v.astype(dtype)


@pytest.fixture(scope='session')
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle


# ---- Tx of the fused ssq_cwt on the column-tile path -----------------------------------------
# The default tile kernel (csrc/ssq_cwt_tiles.hip: tile2_kernel) adds a cell's terms in float64, in
# the order the wavefronts get to them, and rounds once: the BINS are the integers of the CPU loop,
# the sums differ from the reference's running float32 sums (algos.py:912-924) by those sums' own
# rounding -- ~1e-7 of the largest cell (measured 2.4e-7 at config 2), bounded here at 1e-6.
# SSQ_TILE_ORDER=ordered selects the ticketed kernel, whose sums are the CPU loop's bit for bit; the
# tests that check Tx against the oracle run in both modes.
TX_SUM_TOL = 1e-6


def tile_order():
    return 'ordered' if os.environ.get('SSQ_TILE_ORDER') == 'ordered' else 'f64'


def needs_tile_path():
    """Tests of the column-tile path itself: skipped when a run switches it off (SSQ_CWT_TILES=0,
    the whole suite on the block kernels + the two-step reassignment)."""
    if os.environ.get('SSQ_CWT_TILES', '') == '0':
        pytest.skip('the column-tile path is switched off (SSQ_CWT_TILES=0)')


@pytest.fixture(params=['f64', 'ordered'])
def tile_mode(request, monkeypatch):
    monkeypatch.setenv('SSQ_TILE_ORDER', request.param)
    return request.param


TX_SUM_TOL64 = 1e-13


def _tx_tol(dtype):
    return TX_SUM_TOL if np.dtype(dtype) == np.complex64 else TX_SUM_TOL64


def assert_tx_vs_oracle(Tx, ref, tiles=None, what=''):
    """`Tx` against the oracle's reassignment (reference order, sums in the data type) of the device's
    own (Wx, dWx). Bit for bit in the ordered mode, and wherever the caller knows an ordered kernel
    produced it (`tiles=False`: bins from a dWx or a stored w -- the two-step API, ssq_stft with
    dSx kept). Otherwise the fused transforms sum each cell in float64 in arrival order (tile2_kernel,
    accumulate_f64_kernel) and round once: a point in a wrong bin would move |Wx| * const -- 1e-3 .. 1
    of the largest cell -- so 1e-6 of the largest cell (float32; 1e-13 for float64 data) still pins
    every bin that matters, and the sums to the data type's rounding."""
    if tiles is None:
        tiles = True
    if not tiles or tile_order() == 'ordered':
        assert np.array_equal(Tx, ref), what
        return 0.0
    err = float(np.abs(Tx - ref).max() / np.abs(ref).max())
    assert err <= _tx_tol(Tx.dtype), (what, err)
    return err


def assert_tx_repeat(T1, T2, tiles=None, what=''):
    """Two runs on the same data (batched / single, lean / full build, ...). float32 data: the float64
    sums round the same way in any arrival order except when the exact sum sits within ~1e-16 of a
    float32 rounding boundary -- identical but for a stray last bit. float64 data: the sums themselves
    depend on the order, to 1e-13 of the largest cell."""
    if tiles is None:
        tiles = True
    if not tiles or tile_order() == 'ordered':
        assert np.array_equal(T1, T2), what
        return
    ne = T1 != T2
    if T1.dtype == np.complex64:
        assert ne.mean() <= 1e-6, (what, float(ne.mean()))
    if ne.any():
        assert np.abs(T1 - T2).max() <= _tx_tol(T1.dtype) * np.abs(T2).max(), what


def report_measured(test, **values):
    """Print the measured parity figures of a test and append them to
    ``gpurun_out/parity_measured.jsonl`` (when that directory exists): the bounds asserted in the
    tests are set from these (about 2x the measured value), see profiles/r3*_parity_measured.jsonl."""
    import json
    rec = {"test": test}
    rec.update({k: (float(v) if not isinstance(v, (str, int)) else v) for k, v in values.items()})
    print("measured:", json.dumps(rec))
    d = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(d):
        with open(os.path.join(d, 'parity_measured.jsonl'), 'a') as fh:
            fh.write(json.dumps(rec) + "\n")
