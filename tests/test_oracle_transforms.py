# -*- coding: utf-8 -*-
"""Pins the NumPy restatement of the cwt / stft *transforms* (oracle/oracle.py) and
the chaining design -> transform -> reassignment against end-to-end outputs of the
reference (tests/golden/cwt_*.npz, stft_*.npz from oracle/gen_golden.py). CPU-only.
The oracle pipeline assembled here (`oracle_ssq_cwt`, `oracle_ssq_stft`) is what
the GPU parity tests compare the HIP path with at sizes without stored fixtures.
"""
import numpy as np
import pytest
from conftest import golden
from pipeline import oracle_ssq_cwt, oracle_ssq_stft

NUMPY = 1


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_ssq_cwt_pipeline_matches_reference(orc, dtype):
    g = golden('cwt_' + dtype)
    cases = [(256, 'log', 16), (256, 'log-piecewise', 16), (1000, 'log', 8)]
    if dtype == 'float32':
        cases.append((256, 'linear', None))
    for N, st, nv in cases:
        x = g[f'x/{N}']
        r = oracle_ssq_cwt(orc, x, dtype, scales=st, nv=nv, typing=NUMPY)
        pre = f'{N}/{st}'
        assert np.array_equal(r['scales'], g[f'scales/{pre}'])
        assert np.array_equal(r['ssq_freqs'], g[f'ssq_freqs/{pre}'])
        assert np.array_equal(r['Wx'], g[f'Wx/{pre}']), pre
        if f'dWx/{pre}' in g:
            assert np.array_equal(r['dWx'], g[f'dWx/{pre}']), pre
        assert np.array_equal(r['Tx'], g[f'Tx/{pre}']), pre
    # get_w (two-step) and flipud=False variants
    x = g['x/256']
    r = oracle_ssq_cwt(orc, x, dtype, scales='log', nv=16, typing=NUMPY, get_w=True)
    assert np.array_equal(r['w'], g['w/256/log'])
    assert np.array_equal(r['Tx'], g['Tx_getw/256/log'])
    r = oracle_ssq_cwt(orc, x, dtype, scales='log', nv=16, typing=NUMPY,
                       flipud=False)
    assert np.array_equal(r['Tx'], g['Tx_noflip/256/log'])
    # fs != 1
    r = oracle_ssq_cwt(orc, g['x/300'], dtype, scales='log', nv=8, fs=400.,
                       typing=NUMPY)
    assert np.array_equal(r['Wx'], g['Wx/300/fs400'])
    assert np.array_equal(r['dWx'], g['dWx/300/fs400'])
    assert np.array_equal(r['Tx'], g['Tx/300/fs400'])
    assert np.array_equal(r['ssq_freqs'], g['ssq_freqs/300/fs400'])
    # batched == looped
    xb = g['x/batch200']
    for b in range(len(xb)):
        r = oracle_ssq_cwt(orc, xb[b], dtype, scales='log', nv=8, typing=NUMPY)
        assert np.array_equal(r['Wx'], g['Wx/batch200'][b])
        assert np.array_equal(r['Tx'], g['Tx/batch200'][b])


def test_cwt_paddings_and_families(orc):
    g = golden('cwt_float32')
    x = g['x/300']
    for pt in ('zero', 'symmetric', 'replicate', 'wrap', None):
        r = oracle_ssq_cwt(orc, x, 'float32', scales='log', nv=8, padtype=pt,
                           typing=NUMPY, ssq=False)
        key = 'Wx/300/pad_' + (pt or 'none')
        assert np.array_equal(r['Wx'], g[key]), pt
    g = golden('cwt_families')
    for name in ('morlet', 'bump', 'cmhat', 'hhhat'):
        r = oracle_ssq_cwt(orc, g['x'], 'float32', wavelet=name, scales='log',
                           nv=8, typing=NUMPY)
        assert np.array_equal(r['Wx'], g[f'Wx/{name}']), name
        assert np.array_equal(r['Tx'], g[f'Tx/{name}']), name
        assert np.array_equal(r['ssq_freqs'], g[f'ssq_freqs/{name}'])
    r = oracle_ssq_cwt(orc, g['x'], 'float32', wavelet='morlet', scales='log', nv=8,
                       typing=NUMPY, ssq=False, l1_norm=False)
    assert np.array_equal(r['Wx'], g['Wx/morlet_l2'])


@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_ssq_stft_pipeline_matches_reference(orc, dtype):
    g = golden('stft_' + dtype)
    for N, n_fft, hop in ((256, 64, 1), (1000, 128, 32), (2000, 256, 64),
                          (777, 100, 7)):
        pre = f'{N}/{n_fft}/{hop}'
        r = oracle_ssq_stft(orc, g['x/' + pre], dtype, n_fft=n_fft, hop_len=hop,
                            typing=NUMPY)
        assert np.array_equal(r['Sx'], g['Sx/' + pre]), pre
        assert np.array_equal(r['dSx'], g['dSx/' + pre]), pre
        assert np.array_equal(r['Tx'], g['Tx/' + pre]), pre
        assert np.array_equal(r['Sfs'], g['Sfs/' + pre])
    x = g['x/600']
    r = oracle_ssq_stft(orc, x, dtype, n_fft=128, hop_len=16, modulated=False,
                        fs=10., typing=NUMPY, ssq=False)
    assert np.array_equal(r['Sx'], g['Sx/600/nomod'])
    assert np.array_equal(r['dSx'], g['dSx/600/nomod'])
    r = oracle_ssq_stft(orc, x, dtype, n_fft=128, hop_len=16, fs=10., typing=NUMPY,
                        get_w=True)
    assert np.array_equal(r['w'], g['w/600/getw'])
    assert np.array_equal(r['Tx'], g['Tx/600/getw'])
    r = oracle_ssq_stft(orc, x, dtype, window='hann', win_len=100, n_fft=128,
                        hop_len=16, typing=NUMPY, ssq=False)
    assert np.array_equal(r['Sx'], g['Sx/600/hann100'])
