import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import ssqueezepy_amd as S
from conftest import two_chirps
from ssqueezepy_amd import _cwt
N=1<<20
x=torch.as_tensor(two_chirps(N,0),dtype=torch.float32,device='cuda')
for i in range(3):
    torch.cuda.synchronize(); t0=time.time()
    Tx,Wx,sf,sc=S.ssq_cwt(x)
    torch.cuda.synchronize(); print("call",i,"%.1f ms"%((time.time()-t0)*1e3))
plan=next(iter(_cwt._PLAN_CACHE.values()))
print(plan.algo, plan.na, plan.tile_rows, 'tiles done (what executed):', plan.tiles_done(), 'expected', 3 * ((N + 63) // 64))
const=np.log(2)/32
lhs,rhs=Tx.sum(0),None
# log-piecewise: const varies per row: use plan's const? skip identity; check finite
print(bool(torch.isfinite(Tx.real).all()), float(Tx.abs().max()), float(Wx.abs().max()))
plan.timing(1)
for _ in range(3):
    S.ssq_cwt(x)
torch.cuda.synchronize()
ms, n = plan.timing(0)
print("stages us/transform:", [round(v / n * 1e3, 1) for v in ms], "(pad+fft+spectra+intermediates, block rows, exact rows, tile kernel)")
