import sys, os, numpy as np, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']); sys.path.insert(0, os.environ['GRAFT_REPO_ROOT']+'/tests')
import ssqueezepy_amd as S
from conftest import two_chirps
N, na = 160000, 300
wav = S.Wavelet(); scales = S.process_scales('log', N, wav, nv=32)[:na]
x = torch.as_tensor(np.stack([two_chirps(N, s) for s in range(4)]), dtype=torch.float32, device='cuda')
for _ in range(6):
    out = S.cwt(x, wav, scales=scales, derivative=True)
torch.cuda.synchronize()
