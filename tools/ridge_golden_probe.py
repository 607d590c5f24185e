# -*- coding: utf-8 -*-
"""extract_ridges on the device vs the reference's outputs (tests/golden/ridges.npz):
per-case agreement figures (diagnostic twin of tests/test_gpu_ridges.py)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ssqueezepy_amd as S
from conftest import golden

g = golden('ridges')
for k in sorted({k.rsplit('/', 1)[0] for k in g.files if '/' in k}):
    if k == 'basic':
        kw = dict(penalty=2.0, bw=15, transform='cwt', n_ridges=1)
    else:
        a = g[k + '/args']
        kw = dict(penalty=a[0], bw=int(a[1]), transform=('cwt', 'stft')[int(a[2])], n_ridges=2)
    ri, rf, re = S.extract_ridges(g[k + '/Tf'], g[k + '/scales'], get_params=True, **kw)
    ref_i, ref_f, ref_e = g[k + '/idx'], g[k + '/f'], g[k + '/e']
    same = ri == ref_i
    print(k, 'same %.4f per ridge %s maxdiff %d' % (same.mean(), same.mean(axis=0), np.abs(ri - ref_i).max()),
          'f ok', np.array_equal(rf[same], ref_f[same]),
          'e err %.2e' % (np.abs(re[same] - ref_e[same]).max() / np.abs(ref_e).max()),
          'where', np.nonzero(~same.all(axis=1))[0][:12], flush=True)
