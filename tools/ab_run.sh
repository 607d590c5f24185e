#!/bin/bash
# usage inside ONE gpurun call: bash tools/ab_run.sh base v1 v2 ...  (libssq_hip_<v>.so built by tools/ab_build.sh)
# env NWS="16 12": wavefronts per workgroup to try; REPS: repetitions
cd /root/repo
for i in $(seq 1 ${REPS:-1}); do for v in "$@"; do for nw in ${NWS:-16}; do
  if [ $v = base ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  echo -n "$v nw=$nw "; SSQ_TILE_NW=$nw timeout 60 python bench.py --no-cpu --steps 6 ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['stages_us_per_transform']['reassignment_us'],1))"
done; done; done
