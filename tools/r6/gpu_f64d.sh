#!/bin/bash
# round 6: nontemporal hints in the float64 path -- accumulate_f64_kernel's Wx loads + Tx stores (accnt), blockzoom_f64_kernel's Wx stores (blknt)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y22}; mkdir -p $O
for rep in 1 2 3; do for v in "" accnt blknt; do
  if [ -z "$v" ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  echo -n "lib=${v:-new}: "; timeout 300 python tools/run_configs.py c5 2>/dev/null | cut -c1-110
done; done | tee $O/f64_nt.txt
