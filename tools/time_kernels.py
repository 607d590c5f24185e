# -*- coding: utf-8 -*-
"""Micro-timings of the kernel-level entry points at BASELINE config 2 size
(300 x 160000 complex64), with a plain device copy as the bandwidth yardstick."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssqueezepy_amd import algos as A


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3    # us


def main():
    na, n = int(os.environ.get('NA', 300)), int(os.environ.get('NN', 160000))
    g = torch.Generator(device='cuda').manual_seed(0)
    Wx = torch.view_as_complex(torch.randn(na, n, 2, device='cuda', generator=g))
    dWx = torch.view_as_complex(torch.randn(na, n, 2, device='cuda', generator=g))
    w = torch.rand(na, n, device='cuda', generator=g) * 0.5 + 1e-3
    out = torch.empty_like(Wx)
    sf = np.logspace(-3, np.log10(0.5), na)
    t = timeit(lambda: out.copy_(Wx))
    print("copy 384MB->384MB          %8.1f us  %6.2f TB/s" % (t, 2 * Wx.numel() * 8 / t / 1e6))
    t = timeit(lambda: out.zero_())
    print("memset 384MB               %8.1f us  %6.2f TB/s" % (t, Wx.numel() * 8 / t / 1e6))
    t = timeit(lambda: A.indexed_sum_onfly(Wx, w, sf, 0.02, True, True, out=out))
    print("indexed_sum (Wx+w -> Tx)   %8.1f us  %6.2f TB/s" % (t, Wx.numel() * 20 / t / 1e6))
    t = timeit(lambda: A.ssqueeze_fast(Wx, dWx, sf, 0.02, True, True, 1e-6, out=out))
    print("ssqueeze_fast (Wx+dWx->Tx) %8.1f us  %6.2f TB/s" % (t, Wx.numel() * 24 / t / 1e6))
    t = timeit(lambda: A.phase_cwt_gpu(Wx, dWx, 1e-6))
    print("phase_cwt (Wx+dWx -> w)    %8.1f us  %6.2f TB/s" % (t, Wx.numel() * 20 / t / 1e6))


if __name__ == '__main__':
    main()
