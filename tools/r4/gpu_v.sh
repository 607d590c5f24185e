#!/bin/bash
# round 4: tile2_kernel A/B on one box: libs named on the command line (default = in-tree build)
cd /root/repo; O=gpurun_out/r4v; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps 8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }

for v in "$@"; do
  if [ "$v" != default ]; then export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; else unset SSQ_HIP_LIB; fi
  run "lib=$v"
done 2>&1 | tee -a $O/ab.txt
