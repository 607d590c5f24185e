#!/bin/bash
# round 6: the float64 block kernel with its band loads issued ahead of the staging barrier and no barrier in front of its
# first transform (new) against the kernel before (old); config 5, then the float64 tests
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y21}; mkdir -p $O
for rep in 1 2 3; do for v in "" old; do
  if [ -z "$v" ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  echo -n "lib=${v:-new}: "; timeout 300 python tools/run_configs.py c5 2>/dev/null | cut -c1-110
done; done | tee $O/f64_early.txt
unset SSQ_HIP_LIB
timeout 1200 python -m pytest tests -q -m gpu -x -k "float64 or f64 or configs" 2>&1 | tail -2
