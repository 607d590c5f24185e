#!/bin/bash
# round 6: the fused STFT kernel's staged sample loads and select-form bin screen, each against the build without it
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6x}; mkdir -p $O
for rep in 1 2; do
for v in "" nostage oldbins; do
  if [ -z "$v" ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  for B in 512 64 8 1; do
    echo -n "lib=${v:-new} B=$B: "; timeout 300 python tools/probes/c3_batched_probe.py $B 2>/dev/null
  done
  echo -n "lib=${v:-new} hop1: "; timeout 300 python tools/stft_hop1_probe.py 1024 2>/dev/null | cut -c1-200
done; done | tee $O/stft_ab.txt
unset SSQ_HIP_LIB
echo -n "two-pass B=512: "; SSQ_DEBUG_STFT_FUSED_TX=0 timeout 300 python tools/probes/c3_batched_probe.py 512 2>/dev/null | tee -a $O/stft_ab.txt
echo -n "two-pass B=8: "; SSQ_DEBUG_STFT_FUSED_TX=0 timeout 300 python tools/probes/c3_batched_probe.py 8 2>/dev/null | tee -a $O/stft_ab.txt
echo -n "two-pass hop1: "; SSQ_DEBUG_STFT_FUSED_TX=0 timeout 300 python tools/stft_hop1_probe.py 1024 2>/dev/null | cut -c1-200 | tee -a $O/stft_ab.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "stft" 2>&1 | tail -3
# phase stamps
for v in stprof stprof_ns; do for hop in 256 1; do
  SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so timeout 300 python tools/r6/stft_prof.py $hop 2>&1 | grep -v Warning | sed "s/^/$v /"
done; done | tee $O/stft_stamps.txt
