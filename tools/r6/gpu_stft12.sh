#!/bin/bash
# round 6: the mixed-radix kernel with its rows' frequencies asked for ahead of the passes (new) against at their use (nosf)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y17}; mkdir -p $O
for rep in 1 2; do for v in "" gold; do
  if [ -z "$v" ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  echo -n "lib=${v:-new} hop1 598: "; timeout 300 python tools/stft_hop1_probe.py 598 2>/dev/null | cut -c1-130
  echo -n "lib=${v:-new} 598/149 B=512: "; timeout 300 python tools/probes/c3_batched_probe.py 512 598 149 2>/dev/null
done; done | tee $O/stft_generic_sf.txt
