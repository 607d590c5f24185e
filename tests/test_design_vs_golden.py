# -*- coding: utf-8 -*-
"""Host-side design code (scales, filter bank, centre frequencies, synchrosqueezing
frequency grids, STFT windows) must equal the reference's values exactly: these
numbers define the transform. Fixtures come from the
reference itself (oracle/gen_golden.py:gen_design). CPU-only."""
import numpy as np
import pytest
from conftest import golden
from ssqueezepy_amd.wavelets import Wavelet, center_frequency
from ssqueezepy_amd.scales import process_scales, cwt_scalebounds
from ssqueezepy_amd.ssqueezing import _compute_associated_frequencies
from ssqueezepy_amd._bank import banded_bank
from ssqueezepy_amd.padding import p2up, padsignal

WAVS = {'gmw32': 'gmw', 'gmw64': ('gmw', {'dtype': 'float64'}),
        'morlet32': 'morlet',
        'gmw64_b8': ('gmw', {'beta': 8, 'dtype': 'float64'})}


@pytest.fixture(scope='module')
def g():
    return golden('design')


@pytest.mark.parametrize('tag', list(WAVS))
def test_scales_and_bounds(g, tag):
    wav = Wavelet(WAVS[tag])
    for N in (1000, 10000, 160000):
        for st in ('log', 'log-piecewise'):
            sc = process_scales(st, N, wav, nv=32)
            assert np.array_equal(sc, g[f'scales/{tag}/{N}/{st}']), (N, st)
        for preset in ('maximal', 'minimal'):
            b = np.array(cwt_scalebounds(wav, N, preset=preset))
            assert np.array_equal(b, g[f'bounds/{tag}/{N}/{preset}'])
    assert np.array_equal(process_scales('linear', 1000, wav),
                          g[f'scales/{tag}/1000/linear'])


@pytest.mark.parametrize('tag', list(WAVS))
def test_bank_xi_and_center_frequency(g, tag):
    wav = Wavelet(WAVS[tag])
    sc = g[f'bank/{tag}/scales']
    ref = g[f'bank/{tag}/Psih']
    assert np.array_equal(wav(scale=sc, N=4096, nohalf=False), ref)
    assert np.array_equal(wav.xifn(1., 4096), g[f'bank/{tag}/xi'])
    for kind in ('peak', 'energy', 'peak-ct'):
        got = [center_frequency(wav, scale=float(s), N=4096, kind=kind)
               for s in (sc[3, 0], sc[20, 0])]
        assert np.array_equal(np.array(got), g[f'wc/{tag}/{kind}']), kind
    # banded form: identical inside the band, negligible outside
    vals, off, lo = banded_bank(wav, sc.astype(wav.dtype), 4096)
    peak = np.abs(ref).max()
    eps = np.finfo(ref.dtype).eps
    for i in range(len(sc)):
        band = vals[off[i]:off[i + 1]]
        assert np.array_equal(band, ref[i, lo[i]:lo[i] + len(band)]), i
        outside = np.delete(ref[i], np.arange(lo[i], lo[i] + len(band)))
        assert np.abs(outside).max(initial=0) <= 1e-3 * eps * peak * 1.01
    assert off[-1] < 0.5 * ref.size


@pytest.mark.parametrize('tag', list(WAVS))
def test_ssq_frequency_grids(g, tag):
    wav = Wavelet(WAVS[tag])
    n = 0
    for N in (1000, 10000):
        for st in ('log', 'log-piecewise', 'linear'):
            scd = process_scales(st, N, wav, nv=32).astype(wav.dtype)
            for mr in ('peak', 'maximal', 'energy'):
                key = f'ssqf/{tag}/{N}/{st}/{mr}'
                if key not in g:
                    continue
                got = _compute_associated_frequencies(scd, N, wav, st, mr,
                                                      was_padded=True, dt=1.)
                assert np.array_equal(got, g[key]), key
                n += 1
    assert n == 13


def test_windows(g):
    from ssqueezepy_amd._stft import get_window
    for n_fft, win_len in ((128, 128), (1024, 1024), (256, 200), (127, 127)):
        for dtype in ('float32', 'float64'):
            w, dw = get_window(None, win_len, n_fft, derivative=True, dtype=dtype)
            assert np.array_equal(w, g[f'window/dpss/{n_fft}/{win_len}/{dtype}'])
            assert np.array_equal(dw, g[f'dwindow/dpss/{n_fft}/{win_len}/{dtype}'])
        w, dw = get_window('hann', win_len, n_fft, derivative=True, dtype='float64')
        assert np.array_equal(w, g[f'window/hann/{n_fft}/{win_len}/float64'])
        assert np.array_equal(dw, g[f'dwindow/hann/{n_fft}/{win_len}/float64'])


def test_pad_geometry_and_modes():
    # sizes SURVEY.md section 8 quotes for the BASELINE configs
    assert p2up(160000) == (262144, 51072, 51072)
    assert p2up(10000) == (16384, 3192, 3192)
    assert p2up(1048576) == (2097152, 524288, 524288)
    x = np.arange(1., 8.)
    xp, n_up, n1, n2 = padsignal(x, 'reflect', get_params=True)
    assert (n_up, n1, n2) == (16, 5, 4)
    assert np.array_equal(xp[:n1], [6, 5, 4, 3, 2]) and np.array_equal(xp[-n2:], [6, 5, 4, 3])
    assert np.array_equal(padsignal(x, 'symmetric')[:n1], [5, 4, 3, 2, 1])
    assert np.array_equal(padsignal(x, 'replicate')[:n1], [1] * 5)
    assert np.array_equal(padsignal(x, 'wrap')[-n2:], [1, 2, 3, 4])
    assert np.array_equal(padsignal(x, 'zero')[-n2:], [0] * 4)
    assert len(padsignal(x, 'reflect', padlength=12)) == 12


def test_admissibility_constants():
    """adm_ssq / adm_cwt (utils/cwt_utils.py:28-64): same quadrature, same digits."""
    from conftest import golden
    from ssqueezepy_amd.scales import adm_ssq, adm_cwt
    g = golden('inverse')
    for name, spec in (('gmw', 'gmw'), ('gmw_l2', ('gmw', {'norm': 'energy'})),
                       ('morlet', 'morlet'), ('bump', 'bump')):
        assert adm_ssq(spec) == float(g['adm_ssq/' + name]), name
        assert adm_cwt(spec) == float(g['adm_cwt/' + name]), name


def test_higher_order_gmw_samples():
    """Order-k generalized Morse wavelets (_gmw.py:268-394): value-exact samples."""
    from conftest import golden
    from ssqueezepy_amd.wavelets import Wavelet
    g = golden('hiorder')
    w = np.linspace(-1, 12, 527)
    for dtype in ('float32', 'float64'):
        for k in (1, 2, 3):
            wav = Wavelet(('gmw', {'order': k, 'dtype': dtype}))
            assert np.array_equal(wav.fn(w.astype(dtype)), g[f'psih_l1/{dtype}/{k}'],
                                  equal_nan=True), (dtype, k)
    for k in (1, 2):
        wav = Wavelet(('gmw', {'order': k, 'norm': 'energy', 'dtype': 'float64'}))
        assert np.array_equal(wav.fn(w.copy()), g[f'psih_l2/float64/{k}'], equal_nan=True)


def test_degenerate_frequency_range_raises():
    """A compactly supported wavelet whose support misses the grid at the extreme scales
    has no peak frequency to map (the reference crashes with "cannot convert float NaN to
    integer"): a clear ValueError instead of a NaN frequency grid."""
    import pytest
    from ssqueezepy_amd.wavelets import Wavelet
    from ssqueezepy_amd.scales import process_scales
    from ssqueezepy_amd.ssqueezing import _compute_associated_frequencies
    wav = Wavelet(('bump', {'dtype': 'float32'}))
    N = 9492
    scales, st, _, nv = process_scales('log', N, wav, nv=4, get_params=True)
    with pytest.raises(ValueError):
        _compute_associated_frequencies(scales, N, wav, st, 'peak', True, 1., 'cwt')


def test_scale_frequency_conversions():
    """freq_to_scale / scale_to_freq (experimental.py:15-143) against the reference."""
    import warnings
    from ssqueezepy_amd import Wavelet
    from ssqueezepy_amd.experimental import freq_to_scale, scale_to_freq
    g = golden('experimental')
    for name in ('gmw', 'morlet', 'bump'):
        wav = Wavelet(name)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            for N in (512, 2000):
                sc = g[f's2f/{name}/{N}/scales']
                assert np.array_equal(scale_to_freq(sc, wav, N, fs=2.0), g[f's2f/{name}/{N}/reflect'])
                assert np.array_equal(scale_to_freq(sc, wav, N, padtype=None), g[f's2f/{name}/{N}/none'])
        fr = g[f'f2s/{name}/freqs']
        assert np.array_equal(freq_to_scale(fr, wav, 1024), g[f'f2s/{name}/peak'])
        assert np.array_equal(freq_to_scale(fr * 4, wav, 1024, fs=4, kind='energy',
                                            n_search_scales=100, base=3), g[f'f2s/{name}/energy'])


def test_admissibility_and_bounds_sweeps():
    """The reference's own stability sweeps of the host design code: admissibility constants
    stay away from zero over mu in [4, 30] for every family (tests/adm_coef_test.py, 1e-3), and
    `cwt_scalebounds` runs for N = 64 ... 4096 (tests/misc_test.py:13-20)."""
    from ssqueezepy_amd import Wavelet, adm_cwt, adm_ssq, cwt_scalebounds
    for fam in ('morlet', 'bump', 'cmhat', 'hhhat'):
        for mu in np.linspace(4, 30, 40):
            wav = (fam, {'mu': mu})
            assert adm_cwt(wav) > 1e-3 and adm_ssq(wav) > 1e-3, (fam, mu)
    wavelet = Wavelet(('morlet', {'mu': 6}))
    for N in (4096, 2048, 1024, 512, 256, 128, 64):
        smin, smax = cwt_scalebounds(wavelet, N=N)
        assert 0 < smin < smax


@pytest.mark.parametrize('family', ['gmw', 'morlet', 'bump', 'cmhat', 'hhhat'])
@pytest.mark.parametrize('dtype', ['float32', 'float64'])
def test_bank_evaluation_in_pieces(family, dtype, monkeypatch):
    """Large banks are evaluated in contiguous pieces on a thread pool (`_bank._evaluate`): the
    values must be the ones a single call gives, bit for bit (they define the transform)."""
    from ssqueezepy_amd import _bank
    from ssqueezepy_amd.wavelets import Wavelet
    from ssqueezepy_amd.scales import process_scales
    N, M = 6000, 16384
    wav = Wavelet((family, {'dtype': dtype}))
    sc = np.asarray(process_scales('log', N, wav, nv=16), dtype=dtype).reshape(-1)
    whole = _bank.banded_bank(wav, sc, M)
    monkeypatch.setattr(_bank, '_PAR_MIN', 1 << 10)        # pieces of a few hundred values, odd edges
    pieces = _bank.banded_bank(wav, sc, M)
    assert len(whole[0]) > (1 << 12)
    assert np.array_equal(whole[0].view(np.uint8), pieces[0].view(np.uint8))
    assert np.array_equal(whole[1], pieces[1]) and np.array_equal(whole[2], pieces[2])
