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
    ('ssq_tile_fft.hip', ['-ffp-contract=off']),
    ('ssq_tile_f64.hip', ['-ffp-contract=off']),
    ('ssq_tile_pair.hip', ['-ffp-contract=off']),
    ('ssq_tile_ordered.hip', ['-ffp-contract=off']),
    ('ssq_stft.hip', ['-ffp-contract=off']),
    ('ssq_stft_generic.hip', ['-ffp-contract=off']),
    ('ssq_inverse.hip', ['-ffp-contract=off']),
    ('ssq_ridge.hip', ['-ffp-contract=off']),
    ('ssq_fft.hip', []),
]
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
          '-I' + os.path.join(ROCM, 'include'), '-Wno-unused-result']


def device_code_sha():
    """The last commit that touched the device code (csrc/, include/), "-dirty" appended when the
    working tree differs from it there; None outside a git checkout (the GPU box: the library built
    here travels with its stamp)."""
    root = os.path.join(HERE, '..')
    paths = ['ssqueezepy_amd/csrc', 'include']
    try:
        sha = subprocess.run(['git', 'log', '-1', '--format=%H', '--'] + paths, cwd=root, capture_output=True,
                             text=True, timeout=20)
        if sha.returncode or not sha.stdout.strip():
            return None
        dirty = subprocess.run(['git', 'status', '--porcelain', '--'] + paths, cwd=root, capture_output=True,
                               text=True, timeout=20)
        return sha.stdout.strip() + ('-dirty' if dirty.stdout.strip() else '')
    except Exception:
        return None


def _stamp(objdir, verbose):
    """_obj/build_info.o: the strong definition of `ssq_build_sha_value` (ssq_kernels.hip holds a weak
    "unknown"). Returns (object or None, changed)."""
    src, obj = os.path.join(objdir, 'build_info.c'), os.path.join(objdir, 'build_info.o')
    sha = device_code_sha()
    if sha is None:                       # no git here: keep whatever stamp the objects came with
        return (obj if os.path.isfile(obj) else None), False
    text = 'const char ssq_build_sha_value[] = "%s";\n' % sha
    if os.path.isfile(src) and os.path.isfile(obj) and open(src).read() == text:
        return obj, False
    with open(src, 'w') as fh:
        fh.write(text)
    # (any C compiler will do; without one the weak "unknown" stamp of ssq_kernels.hip stands)
    for cc in (os.environ.get('CC'), 'cc', 'gcc', os.path.join(ROCM, 'lib', 'llvm', 'bin', 'clang')):
        if not cc:
            continue
        cmd = [cc, '-fPIC', '-c', src, '-o', obj]
        try:
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
            return obj, True
        except (OSError, subprocess.CalledProcessError):
            continue
    os.remove(src)
    return None, False


def _newer(src, dst):
    return (not os.path.isfile(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=True):
    objdir = os.path.join(CSRC, '_obj')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
               if f.endswith(('.h', '.inl'))]
    headers.append(os.path.join(HERE, '..', 'include', 'ssq_hip.h'))
    objs, rebuilt, jobs = [], False, []
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
            jobs.append(cmd)
            rebuilt = True
        objs.append(opath)
    if jobs:
        # the translation units are independent: compile the stale ones side by side
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1)) as ex:
            list(ex.map(subprocess.check_call, jobs))
    stamp, restamped = _stamp(objdir, verbose)
    if stamp:
        objs.append(stamp)
    if rebuilt or restamped or not os.path.isfile(LIB):
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
