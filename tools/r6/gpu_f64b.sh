#!/bin/bash
# round 6: accumulate_f64_kernel at config 5 -- loads in flight per wavefront (U = 2 / 4 / 8) x wavefronts per tile (8 / 16)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y20}; mkdir -p $O
for rep in 1 2; do for v in "" acc_u8 acc_u2; do for nw in 8 16; do
  if [ -z "$v" ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  echo -n "lib=${v:-u4} NW=$nw: "; SSQ_DEBUG_ACC64_NW=$nw timeout 300 python tools/run_configs.py c5 2>/dev/null | cut -c1-110
done; done; done | tee $O/f64_acc.txt
