# round 6: phase stamps of tile3_kernel (profiling build, SSQ_HIP_LIB=.../libssq_hip_prof.so): shader clocks per wavefront
# of workgroup 0, summed over the launch, by phase. Usage: python tools/r6/tile_prof.py [batch]
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import ssqueezepy_amd as S
from ssqueezepy_amd import _cwt, algos
from conftest import two_chirps
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N, na = 160000, 300
x = torch.as_tensor(np.stack([two_chirps(N, seed=s) for s in range(B)]).astype('float32'), device='cuda')
wav = S.Wavelet()
scales = S.process_scales('log', N, wav, nv=32)[:na]
for rep in range(2):
    out = S.ssq_cwt(x, wav, scales=scales)
torch.cuda.synchronize()
plan = next(iter(_cwt._PLAN_CACHE.values()))
buf = (ctypes.c_ulonglong * 512)()
def read():
    plan.lib.ssq_cwt_plan_tile_counters(plan._h, buf, 512, algos.stream())
    return np.array(buf[:], dtype=np.float64)
c0 = read()
out = S.ssq_cwt(x, wav, scales=scales)
c1 = read()
d = (c1 - c0)[64:64 + 16 * 12].reshape(16, 12)
names = ['loop/prio', 'issue loads', 'wait data', 'gather+taps', 'modul+store', 'bins+adds', 'to tile end', 'barrier1', 'write-out', 'barrier2', 'records', 'tail']
tot = d.sum(1)
print('kernel', plan.tile_kernel, 'tiles', int(c1[0] - c0[0]))
print('%-12s' % 'phase', ' '.join('%6d' % w for w in range(16)), '   mean%')
for i, nm in enumerate(names):
    print('%-12s' % nm, ' '.join('%6.1f' % (100 * d[w, i] / tot[w]) for w in range(16)), '  %6.1f' % (100 * d[:, i].sum() / tot.sum()))
print('%-12s' % 'Mcycles', ' '.join('%6.2f' % (tot[w] / 1e6) for w in range(16)))
