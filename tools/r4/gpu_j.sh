#!/bin/bash
# round 4: does tile2_kernel's time follow its count of vector-memory instructions? (weights loaded once instead of per item: wrong results)
cd /root/repo; O=gpurun_out/r4j; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
L=/root/repo/ssqueezepy_amd
export SSQ_TILE2_RB_COST=0.7
run f64 A=1
run f64-noweights SSQ_HIP_LIB=$L/libssq_hip_e512.so
run f64-noweights-nostore SSQ_HIP_LIB=$L/libssq_hip_e768.so
run f64-noweights-nw12 SSQ_HIP_LIB=$L/libssq_hip_e512.so SSQ_TILE_NW=12
