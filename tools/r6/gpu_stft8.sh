#!/bin/bash
# round 6: the mixed-radix STFT kernel summing Tx itself (n_fft = 598, hop 1: the reference's published shape) against its
# two-pass route, one box; then the GPU suite's STFT tests
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y12}; mkdir -p $O
for rep in 1 2; do for ft in 0 1; do
  echo -n "FUSED_TX=$ft hop1 598: "; SSQ_DEBUG_STFT_FUSED_TX=$ft timeout 300 python tools/stft_hop1_probe.py 598 2>/dev/null | cut -c1-130
  echo -n "FUSED_TX=$ft hop1 1024: "; SSQ_DEBUG_STFT_FUSED_TX=$ft timeout 300 python tools/stft_hop1_probe.py 1024 2>/dev/null | cut -c1-130
done; done | tee $O/stft_generic.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "stft" 2>&1 | tail -3
