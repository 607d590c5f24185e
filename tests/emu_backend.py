# -*- coding: utf-8 -*-
"""Binds ssqueezepy_amd's host layer to the emulated library (tests/emu/: every kernel of
csrc/ compiled for the host, one OS thread per work-item) with tensors on the host, for the
duration of a `with` block. TEST INFRASTRUCTURE ONLY: the product has no CPU path; this runs
the product's own kernels and host code under an emulator so that their logic can be checked
where there is no GPU."""
import contextlib
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, 'emu')
LIB = os.path.join(EMU, '_build', 'libssq_hip_emu.so')
CLANG = os.path.join(os.environ.get('ROCM_PATH', '/opt/rocm'), 'lib', 'llvm', 'bin', 'clang++')


def available():
    return os.path.isfile(CLANG)


def build():
    # SSQ_EMU_LIB: a build of the same sources made elsewhere (e.g. with -fsanitize=address)
    if os.environ.get('SSQ_EMU_LIB'):
        return os.environ['SSQ_EMU_LIB']
    subprocess.check_call(['make', '-C', EMU, '-s', '-j8'])
    return LIB


@contextlib.contextmanager
def emulated():
    import torch
    import ssqueezepy_amd
    from ssqueezepy_amd import _lib, algos
    for m in ('_cwt', '_stft', '_ssq_cwt', '_ssq_stft', '_inverse', 'ssqueezing', 'ridge_extraction'):
        __import__('ssqueezepy_amd.' + m)
    lib = ctypes.CDLL(build())
    for name, (res, args) in _lib._PROTOS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    assert lib.ssq_version() >= _lib.ABI_VERSION, "stale emulated library: make -C tests/emu clean"
    cpu = torch.device('cpu')
    patches = []

    def patch(obj, name, value):
        patches.append((obj, name, getattr(obj, name)))
        setattr(obj, name, value)

    orig = {'device': algos.device, 'stream': algos.stream, '_require_gpu': algos._require_gpu}
    new = {'device': lambda: cpu, 'stream': lambda: None, '_require_gpu': lambda: None}
    for mod in list(sys.modules.values()):
        if getattr(mod, '__name__', '').startswith('ssqueezepy_amd'):
            for k in orig:
                if getattr(mod, k, None) is orig[k]:
                    patch(mod, k, new[k])
    patch(_lib, 'load', lambda *a, **k: lib)
    patch(_lib, '_lib', lib)
    patch(torch.cuda, 'current_device', lambda: 0)
    patch(torch.Tensor, 'is_cuda', property(lambda self: True))
    patch(torch.Tensor, 'cuda', lambda self, *a, **k: self)
    # On the GPU a device tensor cannot be handed to NumPy (np.asarray(t), t.numpy()) without
    # .cpu(); here tensors live on the host, so that mistake would pass unnoticed (it cost round 1
    # its GPU suite). Emulated tensors therefore refuse the conversion unless they went through
    # .cpu() -- which hands out a flagged alias.
    orig_numpy, orig_cpu = torch.Tensor.numpy, torch.Tensor.cpu

    def guarded_numpy(self, *a, **k):
        if not getattr(self, '_ssq_on_host', False):
            raise TypeError("can't convert cuda:0 device type tensor to numpy. Use Tensor.cpu() "
                            "to copy the tensor to host memory first. (emulated device tensor)")
        return orig_numpy(self, *a, **k)

    def to_host(self, *a, **k):
        t = self.detach() if not self.requires_grad else self.view_as(self)
        t._ssq_on_host = True
        return t

    host_inputs = (torch.from_numpy, torch.as_tensor, torch.tensor)

    def flagged(fn):
        import functools

        @functools.wraps(fn)
        def wrap(*a, **k):
            t = fn(*a, **k)
            if isinstance(t, torch.Tensor) and k.get('device') is None:
                t._ssq_on_host = True       # a tensor the caller built on the host stays a host tensor
            return t
        return wrap
    patch(torch.Tensor, 'numpy', guarded_numpy)
    patch(torch.Tensor, 'cpu', to_host)
    for fn in host_inputs:
        patch(torch, fn.__name__, flagged(fn))
    try:
        yield ssqueezepy_amd
    finally:
        for obj, name, old in reversed(patches):
            setattr(obj, name, old)
        for m in ('_cwt', '_stft'):
            getattr(sys.modules['ssqueezepy_amd.' + m], '_PLAN_CACHE', {}).clear()
