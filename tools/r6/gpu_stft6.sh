#!/bin/bash
# round 6: is the fused STFT kernel bound by its 32-byte output pieces? the same bytes written as full lines (wrong layout)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y9}; mkdir -p $O
for rep in 1 2; do for v in "" nt; do
  if [ -z "$v" ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  echo -n "lib=${v:-new} B=512: "; timeout 300 python tools/probes/c3_batched_probe.py 512 2>/dev/null
  echo -n "lib=${v:-new} hop1: "; timeout 300 python tools/stft_hop1_probe.py 1024 2>/dev/null | cut -c1-130
done; done | tee $O/stft_linear.txt
