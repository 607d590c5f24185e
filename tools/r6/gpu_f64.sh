#!/bin/bash
# round 6: the float64 bin map behind a float32 screen (new) against the exact map at every point (f64noscr: block kernels only);
# config 5, then every float64 test of the GPU suite and the fuzz
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y18}; mkdir -p $O
for rep in 1 2; do for v in "" f64noscr; do
  if [ -z "$v" ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  echo -n "lib=${v:-new}: "; timeout 300 python tools/run_configs.py c5 2>/dev/null | cut -c1-150
done; done | tee $O/f64_screen.txt
unset SSQ_HIP_LIB
timeout 1200 python -m pytest tests -q -m gpu -x -k "float64 or f64 or c5 or C5 or configs or kernels" 2>&1 | tail -3
timeout 600 python tools/fuzz_parity.py 60 77 2>&1 | tail -2
