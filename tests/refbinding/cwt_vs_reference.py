# -*- coding: utf-8 -*-
"""`cwt` of the REFERENCE ITSELF (its CPU path, build container only) against this package with
its kernels under the CPU emulator, at lengths where the block path and the rows continued past
the Nyquist bin (ssqueezepy_amd/_blocks.py: extend_past_nyquist) are active. Prints one JSON line.
    PYTHONPATH=oracle/refshim:<reference> SSQ_GPU=0 python tests/refbinding/cwt_vs_reference.py"""
import json, os, sys, warnings, logging
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('SSQ_GPU', '0'); os.environ['SSQ_PARALLEL'] = '0'
logging.disable(logging.WARNING); warnings.simplefilter('ignore')
import ssqueezepy as R                      # the reference
import emu_backend
from conftest import two_chirps


def relmax(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())


out = []
with emu_backend.emulated() as S:
    from ssqueezepy_amd import _cwt
    for fam, dtype, N, nv in (('gmw', 'float32', 6000, 8), ('gmw', 'float64', 6000, 8),
                              ('morlet', 'float32', 9000, 8), ('morlet', 'float32', 9000, 4)):
        x = two_chirps(N, seed=N)
        rw = R.Wavelet((fam, {'dtype': dtype}))
        Wr, sr, dWr = R.cwt(x, rw, scales='log', nv=nv, derivative=True)
        _cwt.clear_plan_cache()
        Wa, sa, dWa = S.cwt(x, S.Wavelet((fam, {'dtype': dtype})), scales='log', nv=nv, derivative=True,
                            astensor=False)
        plan = next(iter(_cwt._PLAN_CACHE.values()))
        ne = plan.extended_rows
        cut = int(((plan._bank[2] + np.diff(plan._bank[1])) == plan.M // 2 + 1).sum())
        out.append(dict(family=fam, dtype=dtype, N=N, nv=nv, na=len(sa), extended_rows=ne, cut_rows=cut, algo=plan.algo,
                        scales_equal=bool(np.array_equal(np.asarray(sr).squeeze(), np.asarray(sa).squeeze())),
                        eW=relmax(Wa, Wr), eD=relmax(dWa, dWr),
                        eW_extended=relmax(Wa[cut - ne:cut], np.asarray(Wr)[cut - ne:cut]) if ne else None))
print(json.dumps(out))
