# -*- coding: utf-8 -*-
"""ssq_stft at hop 1, N=160 000: time and the assignment-invariant checksum.
    python tools/stft_hop1_probe.py [n_fft ...]     default: 1024 (the fused kernel: 513 x 160 000
outputs, 657 MB each) and 598 (the reference's published shape, examples/benchmarks.py:81 -- not a
power of two: framing kernel + rocFFT route)"""
import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ssqueezepy_amd as S
from conftest import two_chirps

N = 160000
x = torch.as_tensor(two_chirps(N, 0), dtype=torch.float32, device='cuda')
for n_fft in ([int(a) for a in sys.argv[1:]] or [1024, 598]):
  for _ in range(2):
      Tx, Sx, sf, Sfs = S.ssq_stft(x, n_fft=n_fft, hop_len=1, dtype='float32')
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(5):
      Tx, Sx, sf, Sfs = S.ssq_stft(x, n_fft=n_fft, hop_len=1, dtype='float32')
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / 5
  const = float(Sfs[1] - Sfs[0])
  lhs, rhs = Tx.sum(0), (Sx * const).sum(0)
  err = ((lhs - rhs).abs().max() / rhs.abs().max()).item()
  bytes_alg = N * 4 + 2 * Sx.numel() * 8
  print(json.dumps({"config": "ssq_stft N=160k n_fft=%d hop=1 f32" % n_fft, "shape": list(Sx.shape), "ms": ms,
                    "GBps_alg": bytes_alg / ms / 1e6, "colsum_rel_err": err}))
  assert err < 2e-5
