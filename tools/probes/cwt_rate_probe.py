# -*- coding: utf-8 -*-
"""`cwt` alone at config 2's size (every row on the block kernels, no reassignment): transforms/s with and without the
derivative, 4 signals per call (HIP events)."""
import json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ssqueezepy_amd as S
from conftest import two_chirps


def rate(fn, n=8, per=4):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return per * n / (e0.elapsed_time(e1) * 1e-3)


N, na = 160000, 300
wav = S.Wavelet()
scales = S.process_scales('log', N, wav, nv=32)[:na]
x = torch.as_tensor(np.stack([two_chirps(N, s) for s in range(4)]), dtype=torch.float32, device='cuda')
print(json.dumps({"cwt_per_s": rate(lambda: S.cwt(x, wav, scales=scales)),
                  "cwt_derivative_per_s": rate(lambda: S.cwt(x, wav, scales=scales, derivative=True))}))
