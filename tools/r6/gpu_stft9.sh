#!/bin/bash
# round 6: the mixed-radix kernel's framing loop unrolled 1 / 4 / 8 times (n_fft = 598, hop 1)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y13}; mkdir -p $O
for rep in 1 2; do for v in "" gen_u1 gen_u8; do
  if [ -z "$v" ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  echo -n "lib=${v:-u4} hop1 598: "; timeout 300 python tools/stft_hop1_probe.py 598 2>/dev/null | cut -c1-130
done; done | tee $O/stft_generic_unroll.txt
