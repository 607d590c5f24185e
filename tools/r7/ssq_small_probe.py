# -*- coding: utf-8 -*-
"""ssq_cwt of ONE short signal (N = 10 000, 300 log scales, float32: config 1's shape through the whole transform), HIP-event
timed, one call at a time -- the interactive use of the reference. `python tools/r7/ssq_small_probe.py [N]`"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ssqueezepy_amd as S
from conftest import two_chirps

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
na = 300
wav = S.Wavelet()
scales = S.process_scales('log', N, wav, nv=32)[:na]
x = torch.as_tensor(two_chirps(N, 0), dtype=torch.float32, device='cuda')
for name, fn in (('ssq_cwt', lambda: S.ssq_cwt(x, wav, scales=scales)), ('cwt', lambda: S.cwt(x, wav, scales=scales))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    print(json.dumps({"call": name, "N": N, "na": na, "ms": e0.elapsed_time(e1) / 50}))
from ssqueezepy_amd._cwt import _PLAN_CACHE
print([p.algo for p in _PLAN_CACHE.values()])
