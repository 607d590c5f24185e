# round 6: phase stamps of stft_fused_kernel's Tx-summing form (profiling build -DSTFT_STAMPS, SSQ_HIP_LIB=.../libssq_hip_stprof.so):
# shader clocks of each workgroup's first wavefront, summed over a launch, by phase.
#   python tools/r6/stft_prof.py [hop] [batch]
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import ssqueezepy_amd as S
from ssqueezepy_amd import _lib
from conftest import two_chirps
hop = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else (512 if hop > 1 else 1)
N = 160000
xb = np.stack([two_chirps(N, s) for s in range(min(B, 8))]); xb = np.tile(xb, (max(1, B // 8), 1))[:B]
x = torch.as_tensor(xb, dtype=torch.float32, device='cuda')
for _ in range(2): out = S.ssq_stft(x, n_fft=1024, hop_len=hop, dtype='float32')
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 16)()
def read():
    torch.cuda.synchronize(); lib.ssq_debug_stft_prof(buf); return np.array(buf[:], dtype=np.float64)
c0 = read(); out = S.ssq_stft(x, n_fft=1024, hop_len=hop, dtype='float32'); c1 = read()
d = (c1 - c0)[:9]
names = ['stage samples', 'to registers', 'transform', 'natural order', 'split+Sx+bins', 'zero planes', 'adds', 'Tx stores', 'drain']
print('hop', hop, 'batch', B, 'workgroups x clocks', d.sum())
for nm, v in zip(names, d): print('  %-14s %5.1f %%' % (nm, 100 * v / d.sum()))
