# -*- coding: utf-8 -*-
"""Config 3 at batch 512, piece by piece (HIP events): `stft` alone (framing + FFT + Sx store), with the derivative
(dSx stored too), and `ssq_stft` (bins + reassignment) -- what the fused STFT kernel's time is made of."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ssqueezepy_amd as S
from conftest import two_chirps


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


N, B = 160000, int(os.environ.get('C3_B', 512))
xb = np.tile(np.stack([two_chirps(N, s) for s in range(8)]), (B // 8, 1))
x = torch.as_tensor(xb, dtype=torch.float32, device='cuda')
kw = dict(n_fft=1024, hop_len=256, dtype='float32')
print(json.dumps({"batch": B,
                  "stft_ms": timeit(lambda: S.stft(x, **kw)),
                  "stft_derivative_ms": timeit(lambda: S.stft(x, derivative=True, **kw)),
                  "ssq_stft_ms": timeit(lambda: S.ssq_stft(x, **kw)),
                  "ssq_stft_get_dWx_ms": timeit(lambda: S.ssq_stft(x, get_dWx=True, **kw))}))
