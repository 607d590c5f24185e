#!/bin/bash
# round 4: tile2_kernel A/B on one box -- wavefronts of interpolated rows in a loop of their own (default)
# against the general loop for all (oneloop), and ablations (WRONG RESULTS) that show what the time is made of
cd /root/repo; O=gpurun_out/r4t; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps 8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
timeout 900 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_00_configs.py -x -q -m gpu -k "tile or config2_ssq" 2>&1 | tail -2 | cut -c1-200
for v in "" oneloop x256 x512 x1024 x1792 ""; do
  if [ -n "$v" ]; then export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; else unset SSQ_HIP_LIB; fi
  run "lib=${v:-default}"
done 2>&1 | tee $O/ab.txt
unset SSQ_HIP_LIB
run lp --scales log-piecewise | tee -a $O/ab.txt
