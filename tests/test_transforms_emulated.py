# -*- coding: utf-8 -*-
"""cwt / ssq_cwt / stft / ssq_stft / inverses through the public API with the product's own
kernels and host code running under the CPU emulator (tests/emu/, tests/emu_backend.py: every
translation unit of csrc/ compiled for the host, one OS thread per work-item, a CPU DFT in
rocFFT's place). Same assertions as the GPU suite -- several of its test functions are run
as they are -- at sizes the emulator finishes in seconds: the reference's fixtures for the
generic plans and the fused STFT kernel, the CPU oracle for the block ("overlap-save zoom")
kernels in both precisions. It checks everything but what only a GPU can show (speed, the
hardware's arithmetic in `log2`/`rcp`/rocFFT). CPU-only."""
import numpy as np
import pytest
import emu_backend
from conftest import golden, two_chirps
from pipeline import oracle_ssq_cwt

RTOL = {'float32': 1e-5, 'float64': 1e-12}


@pytest.fixture(scope='module')
def S():
    if not emu_backend.available():
        pytest.skip("no clang++ under $ROCM_PATH/lib/llvm/bin")
    with emu_backend.emulated() as mod:
        yield mod


def relmax(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, 'detach') else t


@pytest.mark.parametrize('dtype,N,st,nv', [('float32', 256, 'log-piecewise', 16),
                                          ('float64', 256, 'log', 16),
                                          ('float32', 256, 'linear', None)])
def test_generic_plan_vs_reference(S, orc, dtype, N, st, nv):
    """ssq_cwt below the block path's minimum length: banded multiply -> inverse FFT ->
    epilogue -> reassignment, against the reference's outputs (tests/golden/cwt_*.npz)."""
    from test_gpu_transforms import check_Tx
    g = golden('cwt_' + dtype)
    wav = S.Wavelet(('gmw', {'dtype': dtype}))
    x = g[f'x/{N}']
    Tx, Wx, ssq_freqs, scales, dWx = S.ssq_cwt(x, wav, scales=st, nv=nv, get_dWx=True)
    Tx, Wx, dWx = _np(Tx), _np(Wx), _np(dWx)
    pre = f'{N}/{st}'
    assert np.array_equal(scales, g[f'scales/{pre}'])
    assert np.array_equal(ssq_freqs, g[f'ssq_freqs/{pre}'])
    assert relmax(Wx, g[f'Wx/{pre}']) <= RTOL[dtype]
    r = oracle_ssq_cwt(orc, x, dtype, scales=st, nv=nv)
    check_Tx(orc, Tx, Wx, dWx, r, dtype)          # exact given the emulated Wx, dWx


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_block_kernels_vs_oracle(S, orc, dtype):
    """The block ("overlap-save zoom") kernels -- LDS FFT, fused epilogue with the 2-byte bin
    map, reassignment from the bin map -- against the oracle of the reference's full-length
    algorithm."""
    from test_gpu_transforms import check_Tx
    from ssqueezepy_amd import _cwt
    N, nv = 1500, 8
    x = two_chirps(N, seed=N)
    wav = S.Wavelet(('gmw', {'dtype': dtype}))
    _cwt.clear_plan_cache()
    Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, wav, scales='log', nv=nv, get_dWx=True, astensor=False)
    plan = next(iter(_cwt._PLAN_CACHE.values()))
    assert plan.algo.startswith('blockzoom') and plan.block_rows > 0.5 * len(sc)
    r = oracle_ssq_cwt(orc, x, dtype, scales='log', nv=nv, typing=1)
    assert np.array_equal(sf, r['ssq_freqs']) and np.array_equal(sc, r['scales'])
    assert relmax(Wx, r['Wx']) <= RTOL[dtype] and relmax(dWx, r['dWx']) <= RTOL[dtype]
    check_Tx(orc, Tx, Wx, dWx, r, dtype)
    # without dWx the kernels write the bin map only (the bench configuration)
    Tx2, Wx2, *_ = S.ssq_cwt(x, wav, scales='log', nv=nv, astensor=False)
    from conftest import assert_tx_repeat
    assert np.array_equal(Wx2, Wx)
    assert_tx_repeat(Tx2, Tx)
    # a plain cwt runs the kernels' Wx-only instantiation (no derivative transform): the same Wx, bit for bit
    Wc = S.cwt(x, wav, scales='log', nv=nv, astensor=False)[0]
    Wd = S.cwt(x, wav, scales='log', nv=nv, derivative=True, astensor=False)[0]
    assert np.array_equal(Wc, Wd) and relmax(Wc, r['Wx']) <= RTOL[dtype]
    _cwt.clear_plan_cache()


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_nyquist_rows_as_block_rows_vs_oracle(S, orc, dtype, monkeypatch):
    """Rows cut by the Nyquist bin, continued past it and run by the block kernels over the
    analytic signal (_blocks.extend_past_nyquist; classes with analytic = 1) against the oracle
    of the reference's full-length algorithm, and next to the exact path they replace."""
    from test_gpu_transforms import check_Tx
    from ssqueezepy_amd import _cwt
    N, nv = 6000, 8
    x = two_chirps(N, seed=N)
    wav = S.Wavelet(('gmw', {'dtype': dtype}))
    r = oracle_ssq_cwt(orc, x, dtype, scales='log', nv=nv, typing=1)
    out = {}
    for ext in ('1', '0'):
        monkeypatch.setenv('SSQ_DEBUG_CWT_NYQ_EXT', ext)
        _cwt.clear_plan_cache()
        Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, wav, scales='log', nv=nv, get_dWx=True, astensor=False)
        plan = next(iter(_cwt._PLAN_CACHE.values()))
        if ext == '1':
            assert plan.extended_rows >= 5 and plan.block_rows == len(sc), (plan.algo, plan.extended_rows)
            assert 'fourstep' not in plan.algo and 'rocfft' not in plan.algo, plan.algo
            assert plan.block_plan['classes'][:, 4].sum() >= 1
        else:
            assert plan.extended_rows == 0 and plan.block_rows < len(sc)
        assert relmax(Wx, r['Wx']) <= RTOL[dtype] and relmax(dWx, r['dWx']) <= RTOL[dtype]
        check_Tx(orc, Tx, Wx, dWx, r, dtype)
        out[ext] = (Wx, plan.extended_rows)
    # the rows the two runs evaluate differently agree to rounding; the others are identical
    n_ext = out['1'][1]
    assert np.array_equal(out['1'][0][n_ext:], out['0'][0][n_ext:])
    assert relmax(out['1'][0][:n_ext], out['0'][0][:n_ext]) <= RTOL[dtype]
    _cwt.clear_plan_cache()


def test_stft_paths_vs_reference(S, orc):
    """The GPU suite's own ssq_stft test (fused STFT kernel with the bin map, generic
    rocFFT path, reference fixtures) under the emulator."""
    from test_gpu_transforms import test_ssq_stft_vs_reference
    test_ssq_stft_vs_reference(S, orc, 'float32')


def test_block_classes_in_one_launch_emulated(S, monkeypatch):
    """Small transforms run every block class in ONE launch (blockzoom_multi_kernel: the class's body
    picked per workgroup); the same bodies as the per-class kernels, so the results are identical --
    `cwt` with derivative and the fused `ssq_cwt` (lean bodies), float32."""
    from conftest import two_chirps
    from ssqueezepy_amd import _cwt
    x = two_chirps(3000, seed=5)
    wav = S.Wavelet()
    out = {}
    for multi in ('0', '1'):
        monkeypatch.setenv('SSQ_DEBUG_CWT_BLOCKS_MULTI', multi)
        _cwt.clear_plan_cache()
        Wx, sc, dWx = S.cwt(x, wav, scales='log', nv=16, derivative=True, astensor=False)
        plan = next(iter(_cwt._PLAN_CACHE.values()))
        assert plan.algo.startswith('blockzoom') and plan.block_rows > 0.5 * len(sc)
        Tx, W2, *_ = S.ssq_cwt(x, wav, scales='log', nv=16, astensor=False)
        out[multi] = (Wx, dWx, Tx, W2)
    for a, b in zip(out['0'], out['1']):
        assert np.array_equal(a, b)
    _cwt.clear_plan_cache()


def test_tile_walk_through_signal_boundaries_emulated(S, monkeypatch):
    """tile2_kernel's workgroups walk tiles b, b + G, ... of the signals laid end to end (no short last round per
    signal) when the lanes' resident weights survive the boundary: same results as the walk that restarts at every
    signal (SSQ_DEBUG_TILE2_CARRY=0); 6 and 4 workgroups over 80 / 126 tiles x 3 signals, so that the boundary shifts.
    (Round 5: with 16 workgroups and more the workgroups' first tiles are permuted per XCD -- workgroup b starts at tile
    (b mod 8) G / 8 + b / 8 -- so that the workgroups of one XCD walk adjacent tiles; SSQ_DEBUG_TILE2_XCD=0 is the identity:
    same results, both walks.)"""
    from conftest import two_chirps
    from ssqueezepy_amd import _cwt
    for grid, N in (('6', 2560), ('4', 4003), ('16', 2560), ('24', 4003)):
        monkeypatch.setenv('SSQ_DEBUG_TILE_GRID', grid)
        xb = np.stack([two_chirps(N, seed=s) for s in range(3)])
        out = {}
        for carry, xcd in (('0', '1'), ('1', '1'), ('0', '0'), ('1', '0')):
            monkeypatch.setenv('SSQ_DEBUG_TILE2_CARRY', carry)
            monkeypatch.setenv('SSQ_DEBUG_TILE2_XCD', xcd)
            _cwt.clear_plan_cache()
            Tx, Wx, *_ = S.ssq_cwt(xb, S.Wavelet(), scales='log', nv=16, astensor=False)
            plan = next(iter(_cwt._PLAN_CACHE.values()))
            assert plan.tile_rows > 0 and plan.tiles_done() == 3 * plan.tiles_per_signal(N)
            out[carry + xcd] = (Tx, Wx)
        for k in ('11', '00', '10'):
            assert np.array_equal(out['01'][1], out[k][1]) and np.array_equal(out['01'][0], out[k][0]), (grid, k)
    _cwt.clear_plan_cache()


def test_mixed_radix_stft_emulated(S, orc):
    """Round 5: window lengths that are not powers of two run the mixed-radix fused kernel (csrc/ssq_stft_generic.hip)
    when their prime factors are <= 31 and the transform fits the LDS -- the reference's published benchmark length 598 =
    2 x 13 x 23 among them -- and framing + rocFFT otherwise; what each plan chose is asserted on, the results against
    the oracle (reference tolerance), the block spectra of the CWT through the own 4096-point kernel and through rocFFT
    against each other."""
    from conftest import two_chirps
    from pipeline import oracle_ssq_stft
    from ssqueezepy_amd import _stft, _lib
    relmax = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    for n_fft, want in ((598, 'fused-mixed-radix'), (1001, 'fused-mixed-radix'), (97, 'rocfft'), (2 * 37, 'rocfft'),
                        (256, 'fused'), (5000, 'rocfft')):
        N = 3 * n_fft + 700
        x = two_chirps(N, seed=n_fft)
        _stft._PLAN_CACHE.clear()
        Tx, Sx, sf, Sfs, dSx = S.ssq_stft(x, n_fft=n_fft, hop_len=max(1, n_fft // 5), dtype='float32', get_dWx=True,
                                          astensor=False)
        plan = next(iter(_stft._PLAN_CACHE.values()))
        assert plan.algo == want, (n_fft, plan.algo)
        ro = oracle_ssq_stft(orc, x, 'float32', n_fft=n_fft, hop_len=max(1, n_fft // 5))
        assert relmax(Sx, ro['Sx']) <= 1e-5 and relmax(dSx, ro['dSx']) <= 1e-5, n_fft
    _stft._PLAN_CACHE.clear()
    sha = _lib.load().ssq_build_sha().decode()
    assert sha == 'unknown' or len(sha.split('-')[0]) == 40, sha


def test_fused_stft_reassignment_emulated(S, monkeypatch):
    """ssq_stft without dSx: the fused STFT kernel sums Tx of its frames in LDS (float64, unordered) --
    against the ordered two-kernel path on the same input (which the GPU suite checks against the
    oracle), every FFT configuration that is cheap under the emulator, odd hops, flipud, batched;
    SSQ_TILE_ORDER=ordered gives the ordered sums bit for bit."""
    from conftest import two_chirps, assert_tx_vs_oracle
    from ssqueezepy_amd import _stft
    # (hop 1: the frames' samples staged through LDS at an arbitrary alignment; hop 300 at n_fft 128: frames too far apart
    # for the staging buffer, read directly)
    for n_fft, hop, N, fl in ((128, 32, 1500, False), (1024, 256, 6000, False), (256, 37, 3000, True),
                              (128, 1, 2200, False), (128, 300, 12000, True),
                              # (the mixed-radix kernel sums Tx itself too: 598 = 2 x 13 x 23 at G = 8, 60 = 4 x 3 x 5 at G = 16)
                              (598, 119, 2500, False), (60, 7, 900, True)):
        x = two_chirps(N, seed=n_fft)
        _stft._PLAN_CACHE.clear()
        Tx, Sx, *_ = S.ssq_stft(x, n_fft=n_fft, hop_len=hop, dtype='float32', get_dWx=True, flipud=fl,
                                astensor=False)
        T2, S2, *_ = S.ssq_stft(x, n_fft=n_fft, hop_len=hop, dtype='float32', flipud=fl, astensor=False)
        assert np.array_equal(S2, Sx)
        assert_tx_vs_oracle(T2, Tx, what=n_fft)
        Tb, _, *_ = S.ssq_stft(np.stack([x, x[::-1].copy()]), n_fft=n_fft, hop_len=hop, dtype='float32',
                               flipud=fl, astensor=False)
        assert_tx_vs_oracle(Tb[0], Tx, what=n_fft)
        monkeypatch.setenv('SSQ_TILE_ORDER', 'ordered')
        T3, *_ = S.ssq_stft(x, n_fft=n_fft, hop_len=hop, dtype='float32', flipud=fl, astensor=False)
        monkeypatch.delenv('SSQ_TILE_ORDER')
        assert np.array_equal(T3, Tx)
    _stft._PLAN_CACHE.clear()


def test_inverses_vs_reference(S, orc):
    """icwt / issq_cwt / istft / issq_stft / trigdiff (tests/test_gpu_inverse.py) under the
    emulator."""
    import test_gpu_inverse as TI
    TI.test_icwt_and_issq_cwt(S, 'float32')
    TI.test_istft_and_issq_stft(S, 'float64')
    TI.test_trigdiff_vs_reference(S, 'float32')
    TI.test_trigdiff_vs_reference(S, 'float64')
    TI.test_phase_ssqueeze_vs_reference(S, orc, 'float32')
    TI.test_phase_ssqueeze_vs_reference(S, orc, 'float64')


def test_cwt_autograd(S, monkeypatch):
    """`cwt` is differentiable (tests/test_gpu_transforms.py::test_cwt_is_differentiable): the
    adjoint of the plan against torch.autograd through a torch.fft statement of the transform."""
    from test_gpu_transforms import test_cwt_is_differentiable
    monkeypatch.setenv('SSQ_EMULATE', '1')        # the test places its tensors accordingly
    test_cwt_is_differentiable(S, 'float64', 'reflect', True)
    test_cwt_is_differentiable(S, 'float32', 'zero', False)


def test_custom_wavelet_functions_do_not_share_cached_plans():
    """Two different user-supplied wavelet functions created one after the other (CPython
    reuses the id() of a freed function object) must not hit each other's cached plan or
    design: the cache key holds the function itself (wavelets.py `key`)."""
    from ssqueezepy_amd.wavelets import Wavelet
    keys = []
    for mu in (5., 6., 7.):
        w = Wavelet(lambda om, mu=mu: np.exp(-(om - mu) ** 2))
        keys.append(w.key())
        del w
    assert len(set(keys)) == 3 and keys[0] != keys[1] != keys[2]
    f = lambda om: np.exp(-(om - 5.) ** 2)
    assert Wavelet(f).key() == Wavelet(f).key()          # the same function: the same plan
    import emu_backend
    x = np.cos(2 * np.pi * 0.05 * np.arange(600)) + 0.3 * np.cos(2 * np.pi * 0.21 * np.arange(600))
    outs = []
    with emu_backend.emulated() as S:
        for mu in (5., 9.):
            def make(mu):
                return lambda om: np.exp(-(om - mu) ** 2) * (om > 0)
            Wx, sc = S.cwt(x, S.Wavelet(make(mu)), scales=2 ** np.arange(1, 6, 0.25), astensor=False)
            outs.append(Wx)
    assert outs[0].shape == outs[1].shape and np.abs(outs[0] - outs[1]).max() > 1e-3


def test_tile_path_emulated_vs_oracle(tile_mode):
    """The column-tile path of the fused ssq_cwt (persistent workgroups, several tiles per
    workgroup; both tile kernels: float64 tile with unordered adds, ticketed float32 tile) under
    the emulator: Wx / dWx against the oracle, Tx against the oracle's reassignment of the device's
    own Wx / dWx (bit for bit in the ordered mode, to float32 rounding otherwise), lean == full
    instantiation, batched == single; odd length (last tile partial, column-by-column write-out) and
    both exponential grids."""
    import emu_backend
    from oracle import oracle as orc
    from pipeline import oracle_ssq_cwt, GRIDNAME
    from conftest import two_chirps, assert_tx_vs_oracle, assert_tx_repeat
    with emu_backend.emulated() as S:
        from ssqueezepy_amd import _cwt
        for N, st in ((5003, 'log-piecewise'), (2500, 'log')):
            x = two_chirps(N, seed=N)
            wav = S.Wavelet()
            _cwt.clear_plan_cache()
            Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, wav, scales=st, nv=16, get_dWx=True, astensor=False)
            plan = next(iter(_cwt._PLAN_CACHE.values()))
            assert plan.tile_cols == (64 if tile_mode == 'ordered' else 32)
            assert plan.tile_rows > 0.5 * plan.na and plan.tiles_done() == plan.tiles_per_signal(N)
            r = oracle_ssq_cwt(orc, x, 'float32', scales=st, nv=16)
            assert np.abs(Wx - r['Wx']).max() <= 1e-5 * np.abs(r['Wx']).max()
            assert np.abs(dWx - r['dWx']).max() <= 1e-5 * np.abs(r['dWx']).max()
            ref = orc.ssqueeze(Wx, dWx, GRIDNAME[r['grid']], r['params'], r['const'], r['gamma'],
                               True, typing=0)
            assert_tx_vs_oracle(Tx, ref)
            T2, W2, *_ = S.ssq_cwt(x, wav, scales=st, nv=16, astensor=False)
            assert_tx_repeat(T2, Tx)
            assert np.array_equal(W2, Wx)
        xb = np.stack([x, x[::-1].copy()])
        Tb, Wb, *_ = S.ssq_cwt(xb, wav, scales=st, nv=16, astensor=False)
        assert_tx_repeat(Tb[0], Tx)
        assert np.array_equal(Wb[0], Wx)
        T1, W1, *_ = S.ssq_cwt(xb[1], wav, scales=st, nv=16, astensor=False)
        assert_tx_repeat(Tb[1], T1)
        assert np.array_equal(Wb[1], W1)


def test_tile_intermediates_four_step_emulated(monkeypatch):
    """The long classes of the tile path's intermediates (L >= 2^14) through the four-step
    kernels of ssq_cwt_tiles.hip (pruned first pass from the signal's spectrum, hardware sin / cos
    twiddles, blocked intermediate, second pass) under the emulator: L = 65 536 (256 x 256),
    32 768 (128 x 256) and 16 384 (128 x 128) in one transform, against the same transform with
    every class on the DFT that stands in for rocFFT, against the oracle, and -- two signals --
    batched == single."""
    import emu_backend
    from oracle import oracle as orc
    from pipeline import oracle_ssq_cwt
    from conftest import two_chirps
    N, nv = 100000, 1
    x = two_chirps(N, seed=N)
    with emu_backend.emulated() as S:
        from ssqueezepy_amd import _cwt
        wav = S.Wavelet()
        res = {}
        for mode in ('own', 'rocfft'):
            monkeypatch.setenv('SSQ_DEBUG_TILE_FFT', mode)
            _cwt.clear_plan_cache()
            Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, wav, scales='log', nv=nv, get_dWx=True, astensor=False)
            plan = next(iter(_cwt._PLAN_CACHE.values()))
            assert plan.tiles_done() == plan.tiles_per_signal(N)
            assert {65536, 32768, 16384} <= set(int(v) for v in plan.tile_plan['classes'][:, 0])
            res[mode] = (Wx, dWx, Tx)
        r = oracle_ssq_cwt(orc, x, 'float32', scales='log', nv=nv)
        for k in range(2):
            assert relmax(res['own'][k], res['rocfft'][k]) <= 1e-6
        assert relmax(res['own'][0], r['Wx']) <= 1e-5 and relmax(res['own'][1], r['dWx']) <= 1e-5
        monkeypatch.setenv('SSQ_DEBUG_TILE_FFT', 'own')
        _cwt.clear_plan_cache()
        xb = np.stack([x, x[::-1].copy()])
        Tb, Wb, *_ = S.ssq_cwt(xb, wav, scales='log', nv=nv, astensor=False)
        from conftest import assert_tx_repeat
        assert np.array_equal(Wb[0], res['own'][0])
        assert_tx_repeat(Tb[0], res['own'][2])
        T1, W1, *_ = S.ssq_cwt(xb[1], wav, scales='log', nv=nv, astensor=False)
        assert np.array_equal(Wb[1], W1)
        assert_tx_repeat(Tb[1], T1)
        _cwt.clear_plan_cache()


def test_tile_path_emulated_partial_launch_group():
    """More signals than a launch group holds, under the emulator: the partial last group and
    the workspaces reused between groups (see tests/test_gpu_edge_cases.py)."""
    import emu_backend
    from conftest import two_chirps, assert_tx_repeat
    N, B = 2200, 18
    xb = np.stack([two_chirps(N, seed=300 + s) for s in range(B)])
    with emu_backend.emulated() as S:
        wav = S.Wavelet()
        Tb, Wb, *_ = S.ssq_cwt(xb, wav, scales='log', nv=8, astensor=False)
        for s in (0, 15, 16, 17):
            T1, W1, *_ = S.ssq_cwt(xb[s], wav, scales='log', nv=8, astensor=False)
            assert np.array_equal(Wb[s], W1), s
            assert_tx_repeat(Tb[s], T1, what=s)


def test_tile_path_emulated_fewer_steps_than_wavefronts(tile_mode):
    """Very few scales: some wavefronts of the tile kernels have no step / item at all and only take
    part in the write-out of each tile."""
    import emu_backend
    from oracle import oracle as orc
    from pipeline import oracle_ssq_cwt, GRIDNAME
    from conftest import two_chirps, assert_tx_vs_oracle
    with emu_backend.emulated() as S:
        from ssqueezepy_amd import _cwt
        for N, nv in ((4500, 2), (8000, 1)):
            x = two_chirps(N, seed=N)
            _cwt.clear_plan_cache()
            Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, S.Wavelet(), scales='log', nv=nv, get_dWx=True,
                                            astensor=False)
            plan = next(iter(_cwt._PLAN_CACHE.values()))
            assert plan.tiles_done() == plan.tiles_per_signal(N) and plan.na < 32
            r = oracle_ssq_cwt(orc, x, 'float32', scales='log', nv=nv)
            assert np.abs(Wx - r['Wx']).max() <= 1e-5 * np.abs(r['Wx']).max()
            ref = orc.ssqueeze(Wx, dWx, GRIDNAME[r['grid']], r['params'], r['const'], r['gamma'],
                               True, typing=0)
            assert_tx_vs_oracle(Tx, ref)


def test_pair_kernel_equals_single_column_kernel_emulated(monkeypatch):
    """Round 6, under the emulator: the tile kernel with a column PAIR per lane (csrc/ssq_tile_pair.hip, the default) against
    the one-column-per-lane kernel (`SSQ_DEBUG_TILE_PAIR=0`): the same arithmetic per point -- `Wx`, `dWx` bit for bit, `Tx` as
    two runs of one kernel -- for even / odd lengths, odd left paddings (tiles start one column early), partial last
    tiles, a two-signal batch and both default grids. See tests/test_gpu_edge_cases.py for the device's run."""
    import emu_backend
    from conftest import two_chirps, assert_tx_repeat
    monkeypatch.delenv('SSQ_TILE_ORDER', raising=False)
    with emu_backend.emulated() as S:
        from ssqueezepy_amd import _cwt
        for N, st in ((2500, 'log'), (2502, 'log-piecewise'), (2499, 'log'), (2558, 'log'), (2049, 'log-piecewise')):
            xb = np.stack([two_chirps(N, seed=N + s) for s in range(2)])
            out = {}
            for pair in ('1', '0'):
                monkeypatch.setenv('SSQ_DEBUG_TILE_PAIR', pair)
                _cwt.clear_plan_cache()
                Tx, Wx, sf, sc, dWx = S.ssq_cwt(xb, S.Wavelet(), scales=st, nv=16, get_dWx=True, astensor=False)
                plan = next(iter(_cwt._PLAN_CACHE.values()))
                assert plan.tile_kernel == (3 if pair == '1' else 2)
                assert plan.tiles_done() == 2 * plan.tiles_per_signal(N)
                out[pair] = (Tx, Wx, dWx)
            assert np.array_equal(out['1'][1], out['0'][1]) and np.array_equal(out['1'][2], out['0'][2]), N
            assert_tx_repeat(out['1'][0], out['0'][0], what=N)
        _cwt.clear_plan_cache()


@pytest.mark.parametrize('N,padtype', [(3000, 'reflect'), (10000, 'reflect'), (10000, 'zero'), (6000, 'symmetric'), (20000, 'reflect')])
def test_short_signal_prestage_and_spectra_in_one_launch_emulated(S, orc, N, padtype, monkeypatch):
    """Short float32 signals (M = 8192 / 16384): `small_prestage_kernel` (pad + forward transform + analytic signal in
    one workgroup per signal) and `block_spectra_multi_kernel` (the P = 4096 / 8192 / 16384 classes' spectra in one
    launch) against the routes they replace (pad kernel, rocFFT, four-step analytic signal, gather + rocFFT:
    SSQ_DEBUG_BLOCK_SPECTRA=rocfft) and against the oracle of the reference's full-length algorithm."""
    from ssqueezepy_amd import _cwt
    nv = 8
    x = two_chirps(N, seed=N).astype(np.float32)
    wav = S.Wavelet(('gmw', {'dtype': 'float32'}))
    out = {}
    for mode in ('', 'rocfft'):
        if mode:
            monkeypatch.setenv('SSQ_DEBUG_BLOCK_SPECTRA', mode)
        else:
            monkeypatch.delenv('SSQ_DEBUG_BLOCK_SPECTRA', raising=False)
        _cwt.clear_plan_cache()
        Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, wav, scales='log', nv=nv, padtype=padtype, get_dWx=True, astensor=False)
        plan = next(iter(_cwt._PLAN_CACHE.values()))
        assert plan.algo.startswith('blockzoom')
        P = plan.block_plan['classes'][:, 0]
        assert P.max() >= 8192, P                     # (a class the P = 4096-only kernel does not serve)
        out[mode] = (Wx, dWx)
    _cwt.clear_plan_cache()
    assert relmax(out[''][0], out['rocfft'][0]) <= 2e-6 and relmax(out[''][1], out['rocfft'][1]) <= 2e-6
    # a plain cwt (Wx alone: the block kernels' instantiation without the derivative) gives the Wx of a cwt with its
    # derivative, bit for bit
    Wc = S.cwt(x, wav, scales='log', nv=nv, padtype=padtype, astensor=False)[0]
    Wd, _, dWd = S.cwt(x, wav, scales='log', nv=nv, padtype=padtype, derivative=True, astensor=False)
    assert np.array_equal(Wc, Wd)
    assert relmax(Wd, out[''][0]) <= 5e-6 and relmax(dWd, out[''][1]) <= 5e-6      # (ssq_cwt's rows come from the tile kernel)
    if padtype == 'reflect':
        r = oracle_ssq_cwt(orc, x, 'float32', scales='log', nv=nv, typing=1)
        assert relmax(out[''][0], r['Wx']) <= 1e-5 and relmax(out[''][1], r['dWx']) <= 1e-5
