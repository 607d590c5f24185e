#!/bin/bash
# round 6: the persistent form of the fused STFT kernel at 3 / 4 / 5 / 8 workgroups per CU (how many are resident?)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y6}; mkdir -p $O
for p in 3 4 5 8; do
  echo -n "per_cu=$p B=512: "; SSQ_DEBUG_STFT_WG_PER_CU=$p timeout 300 python tools/probes/c3_batched_probe.py 512 2>/dev/null
  echo -n "per_cu=$p hop1: "; SSQ_DEBUG_STFT_WG_PER_CU=$p timeout 300 python tools/stft_hop1_probe.py 1024 2>/dev/null | cut -c1-120
done | tee $O/stft_percu.txt
