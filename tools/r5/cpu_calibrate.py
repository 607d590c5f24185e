# -*- coding: utf-8 -*-
"""Calibrates bench.py's CPU baseline (kind "port") against THE REFERENCE ITSELF, once, in the build
container (the GPU box has no /root/reference; SURVEY 8(d), round-4 verdict item 8).

    PYTHONPATH=oracle/refshim:/root/reference SSQ_PARALLEL=1 python tools/r5/cpu_calibrate.py [out.json]

Left column: ssqueezepy's own `ssq_cwt` on BASELINE config 2's input (N = 160 000, 300 'log' scales, float32,
`cache_wavelet=True`; 3 warm-ups, mean of 10 -- examples/benchmarks.py:18-19,30-37), imported with the identity
`numba` stand-in (numba is not installable here), its FFT / bank stages unchanged (scipy.fft, all cores), and its
`@jit(parallel=True)` loop nests -- which would run as Python loops under the stand-in -- replaced IN ITS OWN
FUNCTION TABLE (`algos._cpu_fns`, algos.py:986-1005) by the oracle's OpenMP restatement of the same loops
(oracle/ssq_oracle.c, pinned bit for bit to the reference's outputs by tests/test_oracle_vs_golden.py): "numba
unavailable; loop nests compiled from an equivalent OpenMP restatement".
Right column: `bench.cpu_baseline()` -- the port the GPU box times -- on the same cores, in the same process.
Test / measurement infrastructure; nothing of the product imports it."""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('SSQ_PARALLEL', '1')
os.environ.setdefault('MPLBACKEND', 'Agg')

import numpy as np


def bind_loop_nests(ref_algos, orc):
    """functions with the signatures of algos._cpu_fns entries, backed by the OpenMP restatement"""
    def make(grid, par):
        def fn(Wx, dWx, out, const, gamma, omax, flipud=False, **p):
            if grid == 'log':
                gp = (p['vlmin'], p['dvl'])
            elif grid == 'log-piecewise':
                gp = (p['vlmin0'], p['vlmin1'], p['dvl0'], p['dvl1'], p['idx1'])
            else:
                gp = (p['vmin'], p['dv'])
            assert omax == len(out) - 1
            # typing=1: the NumPy typing the identity-numba reference computes in (what the fixtures pin)
            orc.ssqueeze(Wx, dWx, grid, gp, const, gamma, flipud, typing=orc.TYPING_NUMPY, out=out, parallel=par)
        return fn
    for key, grid in (('log', 'log'), ('log_piecewise', 'log-piecewise'), ('lin', 'linear')):
        ref_algos._cpu_fns['ssq_cwt_' + key] = make(grid, False)
        ref_algos._cpu_fns['ssq_cwt_' + key + '_par'] = make(grid, True)


def main(out_path=None):
    import ssqueezepy                       # the reference (PYTHONPATH), numba = identity stand-in
    assert os.path.realpath(ssqueezepy.__file__).startswith('/root/reference'), ssqueezepy.__file__
    from ssqueezepy import ssq_cwt, Wavelet, algos as ref_algos
    from ssqueezepy.utils import process_scales
    from ssqueezepy.configs import IS_PARALLEL
    from oracle import oracle as orc
    import bench
    orc.lib()
    assert IS_PARALLEL(), "SSQ_PARALLEL=1 expected"
    bind_loop_nests(ref_algos, orc)

    N, na = 160000, 300
    x = bench.two_chirps(N, 0).astype('float32')
    wavelet = Wavelet(('gmw', {'dtype': 'float32'}))
    scales = process_scales('log', N, wavelet, nv=32)[:na]
    kw = dict(wavelet=wavelet, scales=scales)
    for _ in range(3):
        o = ssq_cwt(x, cache_wavelet=True, **kw)
        del o; gc.collect()
    t = []
    for _ in range(10):
        t0 = time.perf_counter()
        o = ssq_cwt(x, cache_wavelet=True, **kw)
        t.append(time.perf_counter() - t0)
        Tx, Wx = o[0], o[1]
        del o
    ref_s = float(np.mean(t))
    assert Tx.shape == (na, N) and Tx.dtype == np.complex64
    port = bench.cpu_baseline(N, na, seconds_budget=60.0)
    # same numbers? (the port's transform against the reference's, same input)
    rec = {"cores": os.cpu_count(), "workload": "ssq_cwt N=160000, 300 log scales (nv=32), float32, cache_wavelet",
           "reference": {"s_per_transform": ref_s, "transforms_per_s": 1.0 / ref_s, "min_s": float(np.min(t)),
                         "what": "ssqueezepy v%s ssq_cwt (numba stand-in; scipy.fft stages unchanged; _cpu_fns loop "
                                 "nests = OpenMP restatement), SSQ_PARALLEL=1, 3 warm-ups + mean of 10"
                                 % ssqueezepy.__version__},
           "port": {"transforms_per_s": port["value"], "s_per_transform": 1.0 / port["value"], "sample": port["sample"]},
           "port_over_reference": port["value"] * ref_s}
    print(json.dumps(rec, indent=1))
    if out_path:
        with open(out_path, 'w') as fh:
            json.dump(rec, fh, indent=1)


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else None)
