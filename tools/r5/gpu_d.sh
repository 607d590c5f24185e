#!/bin/bash
# round 5: the mixed-radix STFT kernel against the power-of-two one on the sizes both can run (A/B), C3 at 1 / 64 / 512 signals
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5i; mkdir -p $O
for m in 0 1; do
  echo "SSQ_DEBUG_STFT_MIXED=$m"
  SSQ_DEBUG_STFT_MIXED=$m timeout 300 python tools/stft_hop1_probe.py 1024 512 2048 256 2>/dev/null
  SSQ_DEBUG_STFT_MIXED=$m timeout 300 python tools/run_configs.py c3 2>/dev/null | cut -c1-200
done 2>&1 | tee $O/ab.txt
