#!/bin/bash
# round 6: phase stamps of the fused STFT kernel (staged and unstaged sample loads), config 3 at 512 signals and hop 1
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6x}; mkdir -p $O
for v in stprof stprof_ns; do for hop in 256 1; do
  SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so timeout 300 python tools/r6/stft_prof.py $hop 2>&1 | grep -v Warning | sed "s/^/$v /"
done; done | tee $O/stft_stamps.txt
