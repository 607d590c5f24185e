# -*- coding: utf-8 -*-
"""The drop-in boundary exercised from plain C: tests/cabi/cabi_smoke.c includes
include/ssq_hip.h, links libssq_hip.so and runs kernels on buffers it allocates through
the ABI's own runtime helpers -- no Python, no torch in the process."""
import os
import shutil
import subprocess
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_and_run(tmp_path, libdir, libfile):
    cc = shutil.which('gcc') or shutil.which('cc')
    assert cc, "no C compiler"
    assert os.path.isfile(os.path.join(libdir, libfile))
    exe = str(tmp_path / 'cabi_smoke')
    subprocess.check_call([cc, '-std=c99', '-O1', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'cabi', 'cabi_smoke.c'), '-o', exe,
                           '-L', libdir, '-l:' + libfile, '-lm',
                           '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'PASS' in out.stdout


def test_c_program_links_and_passes(tmp_path):
    if os.environ.get('SSQ_EMULATE') == '1':
        pytest.skip("see tests/test_cabi_symbols.py::test_c_client_against_emulated_library")
    build_and_run(tmp_path, os.path.join(ROOT, 'ssqueezepy_amd'), 'libssq_hip.so')
