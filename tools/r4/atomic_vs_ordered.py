# -*- coding: utf-8 -*-
"""Config 2 on the device, the tile kernel's two forms side by side: the ticketed form (float sums
in the reference's order) and the default form (LDS float atomics, sums in arrival order). Same
Wx bit for bit; Tx differs by rounding only -- prints the distance and the run-to-run spread.
    python tools/r4/atomic_vs_ordered.py [N] [na] [seeds]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import ssqueezepy_amd as S
from bench import two_chirps

N = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
na = int(sys.argv[2]) if len(sys.argv) > 2 else 300
seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
wav = S.Wavelet()
scales = S.process_scales('log', N, wav, nv=32)[:na]
x = torch.as_tensor(np.stack([two_chirps(N, s) for s in range(seeds)]), dtype=torch.float32, device='cuda')
out = {}
for mode in ('ordered', 'atomic', 'atomic2'):  # atomic = the default kernel (float64 tile, ds_add_f64)
    os.environ['SSQ_TILE_ORDER'] = 'ordered' if mode == 'ordered' else 'f64'
    Tx, Wx, *_ = S.ssq_cwt(x, wav, scales=scales)
    torch.cuda.synchronize()
    out[mode] = (Tx.clone(), Wx.clone())
To, Wo = out['ordered']
res = {"N": N, "na": na, "signals": seeds}
for mode in ('atomic', 'atomic2'):
    T, W = out[mode]
    res[mode] = {"Wx_identical": bool(torch.equal(W, Wo)),
                 "Tx_maxdiff_over_max": float((T - To).abs().max() / To.abs().max()),
                 "Tx_cells_differing": float(((T != To).sum() / T.numel()))}
res["atomic_run_to_run"] = {"Tx_maxdiff_over_max": float((out['atomic'][0] - out['atomic2'][0]).abs().max() / To.abs().max()),
                            "identical": bool(torch.equal(out['atomic'][0], out['atomic2'][0]))}
print(json.dumps(res))
