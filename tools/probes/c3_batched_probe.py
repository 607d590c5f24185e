import sys, os, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import ssqueezepy_amd as S
from conftest import two_chirps
N=160000; B=int(sys.argv[1]) if len(sys.argv)>1 else 512
NFFT=int(sys.argv[2]) if len(sys.argv)>2 else 1024; HOP=int(sys.argv[3]) if len(sys.argv)>3 else 256
xb=np.stack([two_chirps(N,s) for s in range(8)]); xb=np.tile(xb,(max(1,B//8),1))[:B]
NIT=10 if B>=64 else 100
x=torch.as_tensor(xb,dtype=torch.float32,device='cuda')
for _ in range(3): out=S.ssq_stft(x,n_fft=NFFT,hop_len=HOP,dtype='float32')
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(NIT): out=S.ssq_stft(x,n_fft=NFFT,hop_len=HOP,dtype='float32')
e1.record(); torch.cuda.synchronize()
print('ms per step',e0.elapsed_time(e1)/NIT,'per transform us',e0.elapsed_time(e1)/NIT/B*1e3, [tuple(o.shape) for o in out[:2]])
