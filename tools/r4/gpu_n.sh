#!/bin/bash
cd /root/repo; O=gpurun_out/r4n; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
L=/root/repo/ssqueezepy_amd
export SSQ_TILE_ORDER=f64
for rep in 1 2; do
run f64-prio A=1
run f64-noprio SSQ_HIP_LIB=$L/libssq_hip_noprio.so
run f64-prio-rb0.5 SSQ_TILE2_RB_COST=0.5
run f64-prio-rb1.0 SSQ_TILE2_RB_COST=1.0
done 2>&1 | tee $O/ab.txt
SSQ_HIP_LIB=$L/libssq_hip_prof.so SSQ_TILE2_PROF_DUMP=1 timeout 100 python bench.py --no-cpu --steps 1 --warmup 1 > $O/b.json 2> $O/prof.err
grep "tile2 prof" $O/prof.err | tail -16 > $O/prof.txt; cat $O/prof.txt
