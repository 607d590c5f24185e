# -*- coding: utf-8 -*-
"""The drop-in boundary, executed from the REFERENCE's side (build container only).

ssqueezepy's own tests of its loop nests (tests/fft_test.py:141-377: phase_cwt, phase_stft,
replace_under_abs, indexed_sum_onfly, ssqueeze_fast for every grid kind / flip / dtype, fused ==
two-step) compare its serial CPU functions with its parallel ones at its own thresholds
(1e-8 float32 / 1e-16 float64). Here the parallel entries of the reference's function table are
bound to libssq_hip's C ABI by tests/refbinding/ssq_hip_binding.py -- the binding
INTEGRATION.md describes -- and the reference's tests are run unmodified, in a process of their
own, against the CPU emulation of the kernels (tests/emu). Skipped where /root/reference does
not exist (the GPU box)."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('SSQ_REFERENCE', '/root/reference')


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'ssqueezepy')), reason="needs the reference checkout")
def test_reference_kernel_tests_through_the_c_abi():
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import emu_backend
    if not emu_backend.available():
        pytest.skip("needs ROCm's clang++ for the emulated library")
    lib = emu_backend.build()
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'refbinding', 'run_reference_tests.py'), lib],
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    bad = {k: v for k, v in rec['results'].items() if v != 'ok'}
    assert not bad, bad
    assert len(rec['results']) == 8
    # every bound entry was reached by the reference's tests
    for name in rec['installed']:
        assert rec['calls'].get(name, 0) > 0, (name, rec['calls'])


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, 'ssqueezepy')), reason="needs the reference checkout")
def test_nyquist_rows_against_the_reference_itself():
    """`cwt` of the reference's own CPU path against this package (kernels under the emulator) where
    the block path is active and the rows cut by the Nyquist bin run over the analytic signal
    (_blocks.extend_past_nyquist): 1e-5 / 1e-12 of the reference's maximum, the continued rows too."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import emu_backend
    if not emu_backend.available():
        pytest.skip("needs ROCm's clang++ for the emulated library")
    emu_backend.build()
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'oracle', 'refshim'), REF]),
               SSQ_GPU='0', MPLBACKEND='Agg')
    env.pop('SSQ_DEBUG_CWT_NYQ_EXT', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'refbinding', 'cwt_vs_reference.py')],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert len(rec) == 4
    for r in rec:
        tol = 1e-5 if r['dtype'] == 'float32' else 1e-12
        assert r['scales_equal'] and r['extended_rows'] >= 3, r
        # every cut row continued -- except the first row of the coarse Morlet bank (nv = 4), whose
        # peak lies far past Nyquist: the gain limit leaves it on the exact (four-step) path
        refused = r['cut_rows'] - r['extended_rows']
        assert refused == (1 if (r['family'], r['nv']) == ('morlet', 4) else 0), r
        assert ('fourstep' in r['algo']) == (refused > 0), r
        assert r['eW'] <= tol and r['eD'] <= tol and r['eW_extended'] <= tol, r
