#!/bin/bash
# round 6: the fused STFT kernel with its table loads (window pairs, row frequencies, weights) issued ahead of use
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y}; mkdir -p $O
for rep in 1 2; do
for v in ""; do
  if [ -z "$v" ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  for B in 512 64 8 1; do
    echo -n "lib=${v:-new} B=$B: "; timeout 300 python tools/probes/c3_batched_probe.py $B 2>/dev/null
  done
  echo -n "lib=${v:-new} hop1: "; timeout 300 python tools/stft_hop1_probe.py 1024 2>/dev/null | cut -c1-200
done; done | tee $O/stft_ab.txt
unset SSQ_HIP_LIB
for hop in 256 1; do
  SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_stprof.so timeout 300 python tools/r6/stft_prof.py $hop 2>&1 | grep -v "Warning\|WARNING\|amdgpu.ids"
done | tee $O/stft_stamps.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "stft" 2>&1 | tail -3
