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


def test_c_program_links_and_passes(tmp_path):
    cc = shutil.which('gcc') or shutil.which('cc')
    assert cc, "no C compiler"
    libdir = os.path.join(ROOT, 'ssqueezepy_amd')
    assert os.path.isfile(os.path.join(libdir, 'libssq_hip.so'))
    exe = str(tmp_path / 'cabi_smoke')
    subprocess.check_call([cc, '-std=c99', '-O1', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'tests', 'cabi', 'cabi_smoke.c'), '-o', exe,
                           '-L', libdir, '-lssq_hip', '-lm',
                           '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'PASS' in out.stdout
