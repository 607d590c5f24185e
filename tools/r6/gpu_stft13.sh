#!/bin/bash
# round 6: the mixed-radix kernel with two wavefronts per butterfly in its short prime-radix passes (teams) against one
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y24}; mkdir -p $O
for rep in 1 2; do for tm in 1 0; do
  echo -n "teams=$tm hop1 598: "; SSQ_DEBUG_STFT_GEN_TEAMS=$tm timeout 300 python tools/stft_hop1_probe.py 598 2>/dev/null | cut -c1-130
  echo -n "teams=$tm 598/149 B=512: "; SSQ_DEBUG_STFT_GEN_TEAMS=$tm timeout 300 python tools/probes/c3_batched_probe.py 512 598 149 2>/dev/null
  echo -n "teams=$tm 899/225 B=256: "; SSQ_DEBUG_STFT_GEN_TEAMS=$tm timeout 300 python tools/probes/c3_batched_probe.py 256 899 225 2>/dev/null
done; done | tee $O/stft_generic_teams.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "stft" 2>&1 | tail -2
