# -*- coding: utf-8 -*-
"""Run the REFERENCE's own kernel tests (/root/reference/tests/fft_test.py) with its parallel CPU
entries bound to libssq_hip's C ABI (tests/refbinding/ssq_hip_binding.py).

    python tests/refbinding/run_reference_tests.py <path to libssq_hip[_emu].so>

Build container only (it needs /root/reference; numba is replaced by the identity decorators
of oracle/refshim, as the reference's own tests/z_all_test.py:8-20 does). Prints one JSON line:
{"installed": [...], "results": {test name: "ok" | error text}}."""
import json
import os
import sys
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('SSQ_REFERENCE', '/root/reference')

TESTS = ['test_phase_cwt', 'test_phase_stft', 'test_replace_under_abs', 'test_indexed_sum_onfly',
         'test_ssqueeze_cwt', 'test_ssqueeze_stft', 'test_ssqueeze_vs_indexed_sum']


def main(lib_path):
    os.environ['SSQ_GPU'] = '0'
    # `parallel=False` means "serial" only when the environment default is serial too
    # (algos.py:48: `parallel or IS_PARALLEL()`): the reference's loops on one side, the library
    # (bound to the parallel entries) on the other
    os.environ['SSQ_PARALLEL'] = '0'
    os.environ.setdefault('MPLBACKEND', 'Agg')
    sys.path[:0] = [os.path.join(ROOT, 'oracle', 'refshim'), REF, os.path.join(REF, 'tests'), HERE]
    from ssqueezepy import algos
    import ssq_hip_binding
    ref_ssq_stft = algos._cpu_fns['ssq_stft']        # the reference's own loop nest
    installed = ssq_hip_binding.install(algos, lib_path)
    # count the calls that reach the library (the evidence that the tests went through it)
    calls = {}
    for name in installed:
        holder = algos._cpu_fns if name in algos._cpu_fns else None
        fn = holder[name] if holder is not None else getattr(algos, name)

        def wrap(fn=fn, name=name):
            def w(*a, **k):
                calls[name] = calls.get(name, 0) + 1
                return fn(*a, **k)
            return w
        if holder is not None:
            holder[name] = wrap()
        else:
            setattr(algos, name, wrap())
    import fft_test
    results = {}
    for t in TESTS:
        try:
            getattr(fft_test, t)()
            results[t] = 'ok'
        except Exception:
            results[t] = traceback.format_exc(limit=3)
    # the reference's tests never pass `Sfs` to ssqueeze_fast (the STFT form of the fused loop,
    # algos.py:957-984): the same comparison, with it
    try:
        import numpy as np
        np.random.seed(0)
        for dtype in ('float32', 'float64'):
            Sx = np.random.randn(100, 512).astype(dtype) * (1 + 2j)
            dSx = np.random.randn(100, 512).astype(dtype) * (2 - 1j)
            Sfs = np.linspace(0, .5, len(Sx)).astype(dtype)
            for flipud in (False, True):
                a = (Sfs, float(Sfs[1] - Sfs[0]), False)
                kw = dict(flipud=flipud, gamma=1e-2, Sfs=Sfs)
                out1 = algos.ssqueeze_fast(Sx, dSx, *a, **kw)                 # -> the library
                bound = algos._cpu_fns['ssq_stft']
                algos._cpu_fns['ssq_stft'] = ref_ssq_stft
                try:
                    out0 = algos.ssqueeze_fast(Sx, dSx, *a, **kw)             # -> the reference's loops
                finally:
                    algos._cpu_fns['ssq_stft'] = bound
                # (the reference's own thresholds for a linear grid, tests/fft_test.py:272-275)
                assert np.abs(out0 - out1).mean() < (1e-13 if dtype == 'float64' else 1e-5), (dtype, flipud)
        results['ssqueeze_fast with Sfs (added)'] = 'ok'
    except Exception:
        results['ssqueeze_fast with Sfs (added)'] = traceback.format_exc(limit=3)
    print(json.dumps({"installed": installed, "calls": calls, "results": results}))


if __name__ == '__main__':
    main(sys.argv[1])
