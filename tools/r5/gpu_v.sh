#!/bin/bash
# round 5: A/B on one box: the short bench per variant library named on the command line (`default` = in-tree build;
# libssq_hip_<name>.so from tools/ab_build.sh). OUT=<dir under gpurun_out> STEPS=<n>
cd /root/repo; O=gpurun_out/${OUT:-r5v}; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps ${STEPS:-8} "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for v in "$@"; do
  if [ "$v" != default ]; then export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; else unset SSQ_HIP_LIB; fi
  run "lib=$v"
done 2>&1 | tee -a $O/ab.txt
