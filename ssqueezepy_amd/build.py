# -*- coding: utf-8 -*-
"""Build libssq_hip.so (in-tree) with hipcc for gfx950.

    python -m ssqueezepy_amd.build [--force]

hipcc cross-compiles without a GPU. The shared library links rocFFT (the forward FFT
of the padded signal and the generic-length inverse; ROCm 7.x) and the HIP runtime;
everything else is hand-written kernels. Sources that compute bin indices are
compiled with -ffp-contract=off (bit-reproducible index arithmetic); the FFT
kernels are compiled with contraction on.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libssq_hip.so')
ROCM = os.environ.get('ROCM_PATH', '/opt/rocm')
HIPCC = os.path.join(ROCM, 'bin', 'hipcc')

# (source, extra flags)
SOURCES = [
    ('ssq_kernels.hip', ['-ffp-contract=off']),
    ('ssq_cwt.hip', ['-ffp-contract=off']),
    ('ssq_cwt_blocks.hip', ['-ffp-contract=off']),
    ('ssq_cwt_tiles.hip', ['-ffp-contract=off']),
    ('ssq_stft.hip', ['-ffp-contract=off']),
    ('ssq_inverse.hip', ['-ffp-contract=off']),
    ('ssq_ridge.hip', ['-ffp-contract=off']),
    ('ssq_fft.hip', []),
]
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
          '-I' + os.path.join(ROCM, 'include'), '-Wno-unused-result']


def _newer(src, dst):
    return (not os.path.isfile(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=True):
    objdir = os.path.join(CSRC, '_obj')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
               if f.endswith(('.h', '.inl'))]
    headers.append(os.path.join(HERE, '..', 'include', 'ssq_hip.h'))
    objs, rebuilt = [], False
    for src, extra in SOURCES:
        spath = os.path.join(CSRC, src)
        if not os.path.isfile(spath):
            continue
        opath = os.path.join(objdir, src.replace('.hip', '.o'))
        stale = force or _newer(spath, opath) or any(_newer(h, opath)
                                                     for h in headers)
        if stale:
            cmd = [HIPCC] + COMMON + extra + ['-c', spath, '-o', opath]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
            rebuilt = True
        objs.append(opath)
    if rebuilt or not os.path.isfile(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + [
            '-L' + os.path.join(ROCM, 'lib'), '-lrocfft',
            '-Wl,-rpath,' + os.path.join(ROCM, 'lib')]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
