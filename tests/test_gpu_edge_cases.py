# -*- coding: utf-8 -*-
"""Edge cases of the public API on the MI355X against the oracle pipeline (itself
pinned to the reference): odd / tiny lengths, unpadded transforms, custom scale arrays,
time vectors, odd n_fft, short windows, single-scale banks, plan-cache reuse."""
import os
import numpy as np
import pytest
from conftest import (two_chirps, assert_tx_vs_oracle, assert_tx_repeat, tile_mode, needs_tile_path,  # noqa: F401
                      tile_order)
from pipeline import oracle_ssq_cwt, oracle_ssq_stft, GRIDNAME

pytestmark = pytest.mark.gpu
NUMBA = 0


@pytest.fixture(scope='module')
def S():
    from conftest import compute_module
    yield from compute_module()


def relmax(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize('N', [129, 1001, 4097, 9000])
def test_cwt_odd_and_boundary_lengths(S, orc, N):
    x = two_chirps(N, seed=N)
    for dtype, tol in (('float32', 1e-5), ('float64', 1e-12)):
        wav = S.Wavelet(('gmw', {'dtype': dtype}))
        Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, wav, scales='log', nv=8, get_dWx=True,
                                        astensor=False)
        r = oracle_ssq_cwt(orc, x, dtype, scales='log', nv=8, typing=1)
        assert Wx.shape == r['Wx'].shape
        assert relmax(Wx, r['Wx']) <= tol and relmax(dWx, r['dWx']) <= tol
        ref = orc.ssqueeze(Wx, dWx, GRIDNAME[r['grid']], r['params'], r['const'], r['gamma'],
                           True, typing=NUMBA)
        assert_tx_vs_oracle(Tx, ref)


def test_cwt_unpadded_custom_scales_and_time_vector(S, orc):
    x = two_chirps(777, seed=1)
    wav = S.Wavelet()
    # padtype=None: transform length = signal length (odd, not a power of two)
    Wx, sc = S.cwt(x, wav, scales='log', nv=8, padtype=None, astensor=False)
    r = oracle_ssq_cwt(orc, x, 'float32', scales='log', nv=8, padtype=None, ssq=False)
    assert relmax(Wx, r['Wx']) <= 1e-5
    # explicit scale array (log) and a time vector instead of fs
    scales = np.power(2., np.arange(8, 60) / 8.)
    t = np.arange(len(x)) / 250.
    Tx, Wx, sf, sc2, dWx = S.ssq_cwt(x, wav, scales=scales, t=t, get_dWx=True, astensor=False)
    r = oracle_ssq_cwt(orc, x, 'float32', scales=scales, fs=250., typing=1)
    assert np.array_equal(sc2, scales.astype('float32'))
    assert np.allclose(sf, r['ssq_freqs'], rtol=1e-12)
    assert relmax(Wx, r['Wx']) <= 1e-5 and relmax(dWx, r['dWx']) <= 1e-5
    # a three-scale bank (the fewest the scale-type inference accepts), and maprange='maximal' with a linear frequency axis
    W1, s1 = S.cwt(x, wav, scales=np.array([4., 8., 16.]), astensor=False)
    assert W1.shape == (3, len(x))
    Tx, Wx, sf, sc = S.ssq_cwt(x, wav, scales='log', nv=8, ssq_freqs='linear',
                               maprange='maximal', astensor=False)
    assert Tx.shape == Wx.shape and np.all(np.diff(sf) < 0)
    with pytest.raises(ValueError):
        S.ssq_cwt(x, wav, scales='log-piecewise', maprange='maximal')


def test_plan_cache_and_repeated_calls(S):
    from ssqueezepy_amd import _cwt
    _cwt.clear_plan_cache()
    x = two_chirps(5000, seed=4)
    wav = S.Wavelet()
    a = S.ssq_cwt(x, wav, scales='log', nv=8, astensor=False)
    assert len(_cwt._PLAN_CACHE) == 1
    b = S.ssq_cwt(x, wav, scales='log', nv=8, astensor=False)
    assert len(_cwt._PLAN_CACHE) == 1                      # same plan reused
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])   # deterministic
    c = S.ssq_cwt(x, wav, scales='log', nv=8, flipud=False, astensor=False)
    assert np.array_equal(c[0], a[0][::-1])                # new ssq params on the same plan
    S.ssq_cwt(np.stack([x, x]), wav, scales='log', nv=8)   # larger batch -> plan rebuilt
    d = S.ssq_cwt(x, wav, scales='log', nv=8, astensor=False)
    assert np.array_equal(d[0], a[0])
    for n in (300, 400, 500, 600, 700, 800, 900, 1000, 1100):   # cache eviction
        S.cwt(two_chirps(n, 0), wav, scales='log', nv=4)
    assert len(_cwt._PLAN_CACHE) <= 8


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_stft_odd_sizes_and_windows(S, orc, dtype):
    tol = 1e-5 if dtype == 'float32' else 1e-12
    x = two_chirps(1237, seed=8)
    for kw in (dict(n_fft=127, hop_len=5), dict(n_fft=64, hop_len=64),
               dict(n_fft=200, win_len=150, hop_len=13, window='hamming'),
               dict(n_fft=128, hop_len=3, modulated=False, padtype='zero')):
        Tx, Sx, sf, Sfs, dSx = S.ssq_stft(x, dtype=dtype, get_dWx=True, astensor=False, **kw)
        okw = {k: v for k, v in kw.items()}
        r = oracle_ssq_stft(orc, x, dtype, **okw)
        assert Sx.shape == r['Sx'].shape, kw
        assert relmax(Sx, r['Sx']) <= tol and relmax(dSx, r['dSx']) <= tol, kw
        from ssqueezepy_amd.ssqueezing import ssq_grid_params
        _, p = ssq_grid_params(Sfs, False)
        ref = orc.ssqueeze(Sx, dSx, 'linear', p, Sfs[1] - Sfs[0], r['gamma'], False, Sfs=Sfs,
                           typing=NUMBA)
        assert np.array_equal(Tx, ref), kw        # (dSx kept: the ordered kernel takes its bins from it)
    Tf = S.ssq_stft(x, n_fft=64, hop_len=8, dtype=dtype, flipud=True, astensor=False)
    Tn = S.ssq_stft(x, n_fft=64, hop_len=8, dtype=dtype, astensor=False)
    assert_tx_repeat(Tf[0], Tn[0][::-1])
    assert np.array_equal(Tf[2], Tn[2][::-1])


@pytest.mark.parametrize('n_fft', [128, 256, 512, 1024, 2048, 598, 1001, 899, 323, 210, 1600, 24, 729])
def test_fused_stft_every_size(S, orc, n_fft):
    """The fused framing + window + packed-pair FFT kernels (float32): the power-of-two one for each of its five
    FFT configurations, and the mixed-radix one (round 5) for window lengths that cover every radix it has --
    598 = 2 x 13 x 23 (the reference's published benchmark shape, examples/benchmarks.py:78-79), 1001 = 7 x 11 x 13
    (odd), 899 = 29 x 31, 323 = 17 x 19, 210 = 2 x 3 x 5 x 7, 1600 = 16 x 4 x 5 x 5, 24 = 8 x 3, 729 = 3^6 (six passes). Against the oracle (reference tolerance 1e-5), against this engine's rocFFT
    path (env switch), modulated and not, batched, hops that leave a partial last workgroup; Tx exact against the
    oracle reassignment of the device's own Sx, dSx."""
    import os, subprocess, sys, json
    from ssqueezepy_amd import _stft
    N = 6000 + n_fft
    pow2 = n_fft & (n_fft - 1) == 0
    for hop, mod in ((n_fft // 4, True), (37, False)):
        x = two_chirps(N, seed=n_fft + hop)
        _stft._PLAN_CACHE.clear()
        Tx, Sx, sf, Sfs, dSx = S.ssq_stft(x, n_fft=n_fft, hop_len=hop, modulated=mod,
                                          dtype='float32', get_dWx=True, astensor=False)
        plan = next(iter(_stft._PLAN_CACHE.values()))
        assert plan.algo == ('fused' if pow2 else 'fused-mixed-radix'), plan.algo       # what was planned to run
        ro = oracle_ssq_stft(orc, x, 'float32', n_fft=n_fft, hop_len=hop, modulated=mod)
        assert Sx.shape == ro['Sx'].shape
        assert relmax(Sx, ro['Sx']) <= 1e-5 and relmax(dSx, ro['dSx']) <= 1e-5, (hop, mod)
        from ssqueezepy_amd.ssqueezing import ssq_grid_params
        _, p = ssq_grid_params(Sfs, False)
        ref = orc.ssqueeze(Sx, dSx, 'linear', p, Sfs[1] - Sfs[0], ro['gamma'], False,
                           Sfs=Sfs, typing=0)
        assert np.array_equal(Tx, ref), (hop, mod)
        xb = np.stack([x, x[::-1].copy(), 0.5 * x])
        Txb, Sxb, *_ = S.ssq_stft(xb, n_fft=n_fft, hop_len=hop, modulated=mod,
                                  dtype='float32', astensor=False)
        assert np.array_equal(Sxb[0], Sx)
        assert_tx_vs_oracle(Txb[0], Tx)        # (bin map + float64 sums vs the ordered sums above)
        # the generic (rocFFT) path of this engine on the same input
        os.environ['SSQ_DEBUG_STFT_GENERIC'] = '1'
        try:
            _stft._PLAN_CACHE.clear()
            Sx2, dSx2 = S.stft(x, n_fft=n_fft, hop_len=hop, modulated=mod, derivative=True,
                               dtype='float32', astensor=False)
        finally:
            del os.environ['SSQ_DEBUG_STFT_GENERIC']
            _stft._PLAN_CACHE.clear()
        assert relmax(Sx2, ro['Sx']) <= 1e-5 and relmax(Sx, Sx2) <= 1e-5
        assert relmax(dSx, dSx2) <= 1e-5


@pytest.mark.parametrize('n_fft,hop', [(1024, 256), (256, 1), (598, 37), (2048, 300), (128, 77)])
def test_fused_stft_sums_equal_the_separate_pass(S, n_fft, hop, monkeypatch):
    """Round 6: the fused STFT kernels sum Tx themselves at EVERY batch size (their float64 planes take the transform
    buffer's place in LDS). Batches beyond the old 4096-frame rule, the mixed-radix kernel included, against the
    separate pass over the 2-byte bin map (`SSQ_DEBUG_STFT_FUSED_TX=0`): the same Sx bit for bit, the same float64 sums
    of the same terms -- equal up to the order of the adds."""
    from ssqueezepy_amd import _stft
    N, B = (20000 if hop < 64 else 60000), (6 if hop == 1 else 24)
    x = np.stack([two_chirps(N, seed=100 + s) for s in range(B)])
    out = {}
    for ft in ('1', '0'):
        monkeypatch.setenv('SSQ_DEBUG_STFT_FUSED_TX', ft)
        _stft._PLAN_CACHE.clear()
        Tx, Sx, *_ = S.ssq_stft(x, n_fft=n_fft, hop_len=hop, dtype='float32', astensor=False)
        out[ft] = (Tx, Sx)
    monkeypatch.delenv('SSQ_DEBUG_STFT_FUSED_TX')
    _stft._PLAN_CACHE.clear()
    assert B * out['1'][1].shape[-1] > 4096
    assert np.array_equal(out['1'][1], out['0'][1])
    for b in range(B):
        assert_tx_vs_oracle(out['1'][0][b], out['0'][0][b], what=(n_fft, hop, b))


def test_plan_reuse_across_streams_and_parameter_changes(S):
    """One cached plan, two non-blocking streams, reassignment parameters that alternate
    between the calls: an execute must neither see the other's weights nor share its
    workspace unordered (plan state was updated on the null stream)."""
    import torch
    x = two_chirps(30000, seed=7)
    wav = S.Wavelet()
    ref = {}
    for fl in (True, False):
        ref[fl] = S.ssq_cwt(x, wav, scales='log', nv=16, flipud=fl)[0].clone()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(6):
        for st, fl in ((s1, True), (s2, False)):
            with torch.cuda.stream(st):
                outs.append((fl, S.ssq_cwt(x, wav, scales='log', nv=16, flipud=fl)[0]))
    torch.cuda.synchronize()
    for fl, T in outs:
        assert_tx_repeat(T.cpu().numpy(), ref[fl].cpu().numpy())
    y = two_chirps(8192, seed=8)
    r2 = {h: S.ssq_stft(y, n_fft=256, hop_len=h and 32 or 64)[0].clone() for h in (True, False)}
    torch.cuda.synchronize()
    outs = []
    for _ in range(6):
        for st, h in ((s1, True), (s2, False)):
            with torch.cuda.stream(st):
                outs.append((h, S.ssq_stft(y, n_fft=256, hop_len=h and 32 or 64)[0]))
    torch.cuda.synchronize()
    for h, T in outs:
        assert_tx_repeat(T.cpu().numpy(), r2[h].cpu().numpy())


def test_small_transforms_repeat_and_follow_their_inputs(S):
    """Launch-bound sizes called repeatedly with the buffers torch hands out in turn: results
    equal the first call's, follow the contents of the input buffer, and equal a fresh plan's."""
    import torch
    from ssqueezepy_amd import _cwt
    N = 10000
    wav = S.Wavelet()
    scales = S.process_scales('log', N, wav, nv=32)[:300]
    x = torch.as_tensor(two_chirps(N, seed=2), dtype=torch.float32, device='cuda')
    ref = S.cwt(x, wav, scales=scales)[0].clone()
    for _ in range(8):                       # torch hands out the same two output buffers in turn
        out = S.cwt(x, wav, scales=scales)[0]
        assert torch.equal(out, ref)
    x.copy_(torch.as_tensor(two_chirps(N, seed=5), dtype=torch.float32))
    for _ in range(3):
        out = S.cwt(x, wav, scales=scales)[0]
    _cwt.clear_plan_cache()                  # a fresh plan: eager launches
    assert torch.equal(out, S.cwt(x, wav, scales=scales)[0])
    y = torch.as_tensor(two_chirps(4096, seed=9), dtype=torch.float32, device='cuda')
    refs = {fl: S.ssq_cwt(y, wav, scales='log', nv=8, flipud=fl)[0].clone() for fl in (True, False)}
    for _ in range(6):
        for fl in (True, False):
            assert torch.equal(S.ssq_cwt(y, wav, scales='log', nv=8, flipud=fl)[0], refs[fl])
    r3 = S.ssq_stft(y, n_fft=256, hop_len=64)[0].clone()
    for _ in range(6):
        assert torch.equal(S.ssq_stft(y, n_fft=256, hop_len=64)[0], r3)


def test_batches_larger_than_a_launch_group(S):
    """A batch is walked in launch groups (16 signals by default): the last, partial group and
    the reuse of the group's workspaces (bin map, decimated samples of the tile path) between
    groups must not leak from one signal into another."""
    import torch
    N, B = 6000, 21
    wav = S.Wavelet()
    xb = np.stack([two_chirps(N, seed=100 + s) for s in range(B)])
    Tb, Wb, *_ = S.ssq_cwt(xb, wav, scales='log', nv=16)
    for s in (0, 7, 15, 16, 20):
        T1, W1, *_ = S.ssq_cwt(xb[s], wav, scales='log', nv=16)
        assert torch.equal(Wb[s], W1), s
        assert_tx_repeat(Tb[s].cpu().numpy(), T1.cpu().numpy(), what=s)


@pytest.mark.parametrize('N', [100000, 1 << 20])
def test_tile_intermediates_four_step_vs_rocfft(S, N, monkeypatch):
    """The long classes of the tile path's intermediates on the four-step kernels (default) against
    the same transform with every class on rocFFT (SSQ_DEBUG_TILE_FFT=rocfft): N = 100 000 covers
    L = 2^14 .. 2^16, N = 2^20 (padded to 2^21) the factors up to 512 x 1024; batched == single."""
    needs_tile_path()
    from ssqueezepy_amd import _cwt
    if os.environ.get('SSQ_EMULATE') == '1' and N > 200000:
        pytest.skip("2^20 points under the CPU emulation of the kernels take minutes")
    nv = 2
    x = two_chirps(N, seed=N)
    wav = S.Wavelet()
    res = {}
    for mode in ('own', 'rocfft'):
        monkeypatch.setenv('SSQ_DEBUG_TILE_FFT', mode)
        _cwt.clear_plan_cache()
        Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, wav, scales='log', nv=nv, get_dWx=True, astensor=False)
        plan = next(iter(_cwt._PLAN_CACHE.values()))
        assert plan.tiles_done() == plan.tiles_per_signal(N)
        res[mode] = (Wx, dWx)
        lmax = int(plan.tile_plan['classes'][:, 0].max())
    assert lmax >= (1 << 16)
    for k in range(2):
        d = np.abs(res['own'][k] - res['rocfft'][k]).max() / np.abs(res['rocfft'][k]).max()
        assert d <= 2e-6, (k, d)
    monkeypatch.setenv('SSQ_DEBUG_TILE_FFT', 'own')
    _cwt.clear_plan_cache()
    xb = np.stack([x, x[::-1].copy()])
    Tb, Wb, *_ = S.ssq_cwt(xb, wav, scales='log', nv=nv, astensor=False)
    T1, W1, *_ = S.ssq_cwt(xb[1], wav, scales='log', nv=nv, astensor=False)
    assert np.array_equal(Wb[1], W1)
    assert_tx_repeat(Tb[1], T1)
    _cwt.clear_plan_cache()


def test_tile_path_with_few_scales_and_many_tiles(S, orc, tile_mode):
    """Few scales (wavefronts of the tile kernels without a step / item of their own) and a length
    that gives every persistent workgroup several tiles, the last one partial (odd lengths: the
    float64 kernel's column-by-column write-out)."""
    from pipeline import oracle_ssq_cwt, GRIDNAME
    for N, nv in ((70001, 2), (33333, 1)):
        x = two_chirps(N, seed=N)
        Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, S.Wavelet(), scales='log', nv=nv, get_dWx=True,
                                        astensor=False)
        r = oracle_ssq_cwt(orc, x, 'float32', scales='log', nv=nv)
        assert np.abs(Wx - r['Wx']).max() <= 1e-5 * np.abs(r['Wx']).max()
        ref = orc.ssqueeze(Wx, dWx, GRIDNAME[r['grid']], r['params'], r['const'], r['gamma'],
                           True, typing=0)
        assert_tx_vs_oracle(Tx, ref, tiles=True)


@pytest.mark.parametrize('case', ['log/linear', 'log/log-piecewise', 'log-piecewise', 'linear',
                                  'log/f32-weights', 'log/f64-weights'])
def test_tile_kernel_every_instantiation(S, orc, case, tile_mode):
    """Every build of both column-tile kernels (`tile_mode`: the float64 tile with unordered adds --
    the default -- and the ticketed float32 tile), at a length where the block path (M >= 4096) and
    the tile (na <= 316) are both active: the three frequency-grid kinds x the three kinds of
    reassignment weights (one float32; a float32 per row; a float64 per row -- the reference's
    default 'log-piecewise' scales and its 'linear' scales, ssqueezing.py:126-128 ->
    algos.py:66-79), with and without `dWx` stored. What is asserted on is what EXECUTED
    (`plan.tiles_done()` counts the tiles the kernel finished), not the plan's label."""
    needs_tile_path()
    import os
    from ssqueezepy_amd import _cwt, _ssq_cwt
    emulated = os.environ.get('SSQ_EMULATE') == '1'
    N = 6100 if emulated else 20011
    x = two_chirps(N, seed=7)
    wav = S.Wavelet()
    kw = dict(nv=16)
    weights = None
    if case == 'log/linear':
        kw.update(scales='log', ssq_freqs='linear', maprange='maximal')
    elif case == 'log/log-piecewise':
        kw.update(scales='log', ssq_freqs='log-piecewise')
    elif case == 'log-piecewise':
        kw.update(scales='log-piecewise')
    elif case == 'linear':
        kw.update(scales='linear', nv=None)
    else:
        kw.update(scales='log')
        weights = case.split('/')[1]
    _cwt.clear_plan_cache()
    _ssq_cwt._DESIGN_CACHE.clear()
    if weights is None:
        Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, wav, get_dWx=True, astensor=False, **kw)
        plan = next(iter(_cwt._PLAN_CACHE.values()))
        design = _ssq_cwt._ssq_design(wav, kw['scales'], kw['nv'] if kw['scales'] != 'linear' else 32,
                                      N, 1., kw.get('ssq_freqs'), kw.get('maprange', 'peak'), True)
        _, ssq_freqs, const, grid, params = design
        assert np.array_equal(sf, ssq_freqs[::-1])
        gamma = 10 * np.finfo(np.float32).eps
        T2, W2, *_ = S.ssq_cwt(x, wav, astensor=False, **kw)
        n_calls = 2
    else:
        # weights per row through the plan interface (the C ABI's `ssq_cwt_plan_set_ssq` with a
        # vector): a float32 vector stays float32, a float64 one makes the sums go through double
        import torch
        design = _ssq_cwt._ssq_design(wav, 'log', 16, N, 1., None, 'peak', True)
        scales_dt, ssq_freqs, _, grid, params = design
        na = len(scales_dt.reshape(-1))
        const = (np.log(2) / np.linspace(8, 32, na)).astype('float32' if weights.startswith('f32') else 'float64')
        gamma = 10 * np.finfo(np.float32).eps
        plan = _cwt.get_cwt_plan(wav, scales_dt, N, 'reflect', 1., True, 1)
        plan.set_ssq(grid, params, const, True, gamma)
        from ssqueezepy_amd import algos
        xd = algos.to_device(x, torch.float32)
        out = plan.execute(xd, want_dWx=True, want_Tx=True)
        Tx, Wx, dWx = (out[k].cpu().numpy() for k in ('Tx', 'Wx', 'dWx'))
        out = plan.execute(xd, want_Tx=True)
        T2, W2 = out['Tx'].cpu().numpy(), out['Wx'].cpu().numpy()
        n_calls = 2
    ref = orc.ssqueeze(Wx, dWx, GRIDNAME[grid], params, const, gamma, True, typing=0)
    if case == 'linear':
        # 'linear' SCALES leave no row decimated enough for the tile kernels (GRID_LIN on the tile
        # path is what 'log/linear' covers): this case is the block rows + the separate reassignment
        # kernel with float64 per-row weights (bit for bit in the ordered mode)
        assert plan.tile_rows == 0 and plan.tiles_done() == 0, (plan.tile_rows, plan.algo)
        assert_tx_vs_oracle(Tx, ref, what=case)
        assert_tx_repeat(T2, Tx, what=case)
        assert np.array_equal(W2, Wx)
        return
    assert plan.tile_rows > 0.5 * plan.na, (plan.tile_rows, plan.na)
    assert plan.tile_cols == (64 if tile_mode == 'ordered' else 32)
    assert plan.tiles_done() == n_calls * plan.tiles_per_signal(N), (plan.tiles_done(), plan.algo)
    assert_tx_vs_oracle(Tx, ref, tiles=True, what=case)
    # the lean build (no dWx stored) against the full one
    assert np.array_equal(W2, Wx)
    assert_tx_repeat(T2, Tx, tiles=True, what=case)


def test_more_rows_than_the_32_column_tile_holds(S, orc, tile_mode):
    """456 rows -- `process_scales('log', N, nv=32)` without the `[:300]` of the benchmark --
    exceed the 318 rows a 32-column float64 tile (and the ticketed kernel's 64-column float32 tile)
    can keep in a CU's LDS: the default tile kernel then runs 16-column tiles, four rows per
    wavefront instruction. The ordered kernel has no such form: with `SSQ_TILE_ORDER=ordered` the plan
    reports no usable tile kernel (`tile_cols == 0`) and the call takes the block kernels + the ordered
    reassignment for every row -- decided before any row is routed (round-4 advisor: it used to fail
    inside `TilePlan::run`, after the block work had been launched). What executed is asserted on,
    and the result against the oracle, in both modes."""
    needs_tile_path()
    from ssqueezepy_amd import _cwt
    from pipeline import oracle_ssq_cwt, GRIDNAME
    N = 20000 if os.environ.get('SSQ_EMULATE') == '1' else 160000
    wav = S.Wavelet()
    scales = S.process_scales('log', N, wav, nv=32 if N == 160000 else 40)
    x = two_chirps(N, seed=11)
    _cwt.clear_plan_cache()
    Tx, Wx, sf, sc, dWx = S.ssq_cwt(x, wav, scales=scales, get_dWx=True, astensor=False)
    plan = next(iter(_cwt._PLAN_CACHE.values()))
    assert plan.na == len(scales) > 318
    if tile_mode == 'ordered':
        assert plan.tile_cols == 0 and plan.tiles_done() == 0, (plan.tile_cols, plan.tiles_done(), plan.algo)
    else:
        assert plan.tile_cols == 16 and plan.tile_rows > 0.5 * plan.na
        assert plan.tiles_done() == plan.tiles_per_signal(N), (plan.tiles_done(), plan.algo)
    r = oracle_ssq_cwt(orc, x, 'float32', scales=scales)
    assert np.abs(Wx - r['Wx']).max() <= 1e-5 * np.abs(r['Wx']).max()
    assert np.abs(dWx - r['dWx']).max() <= 1e-5 * np.abs(r['dWx']).max()
    ref = orc.ssqueeze(Wx, dWx, GRIDNAME[r['grid']], r['params'], r['const'], r['gamma'], True, typing=0,
                       parallel=True)
    assert_tx_vs_oracle(Tx, ref, tiles=True)
    _cwt.clear_plan_cache()


@pytest.mark.parametrize('N', [4096, 4098, 4099, 4097, 8192 - 2 * 33, 8192 - 2 * 32 - 1, 20010])
def test_pair_kernel_equals_single_column_kernel(S, N, monkeypatch):
    """Round 6: the default tile kernel gives a lane a PAIR of neighbouring columns (csrc/ssq_tile_pair.hip,
    `plan.tile_kernel == 3`); `SSQ_DEBUG_TILE_PAIR=0` keeps the one-column-per-lane kernel of rounds 4-5 (2). Same arithmetic
    per point, so `Wx` and `dWx` must agree bit for bit and `Tx` as two runs of one kernel do -- for even and odd
    lengths, odd left paddings (the pair kernel's tiles then start one column early: n1 = 2047, 2046, 33, 33 here), a
    partial last tile, a first tile with a dead column, and a batch that the walk carries through."""
    needs_tile_path()
    if tile_order() == 'ordered':
        pytest.skip('the ordered mode runs the ticketed kernel')
    from ssqueezepy_amd import _cwt
    xb = np.stack([two_chirps(N, seed=40 + s) for s in range(3)])
    wav = S.Wavelet()
    out = {}
    for pair in ('1', '0'):
        monkeypatch.setenv('SSQ_DEBUG_TILE_PAIR', pair)
        _cwt.clear_plan_cache()
        Tx, Wx, sf, sc, dWx = S.ssq_cwt(xb, wav, scales='log', nv=16, get_dWx=True, astensor=False)
        plan = next(iter(_cwt._PLAN_CACHE.values()))
        assert plan.tile_kernel == (3 if pair == '1' else 2), plan.tile_kernel
        assert plan.tiles_done() == 3 * plan.tiles_per_signal(N)
        T1, W1, *_ = S.ssq_cwt(xb[1], wav, scales='log-piecewise', nv=16, astensor=False)
        out[pair] = (Tx, Wx, dWx, T1, W1)
    for k in (1, 2, 4):
        assert np.array_equal(out['1'][k], out['0'][k]), k
    assert_tx_repeat(out['1'][0], out['0'][0])
    assert_tx_repeat(out['1'][3], out['0'][3])
    _cwt.clear_plan_cache()
