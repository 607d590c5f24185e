#!/bin/bash
# round 4: tile2_kernel v4 (one scalar record per item, resident weights, regular addressing)
cd /root/repo; O=gpurun_out/r4m; mkdir -p $O
timeout 200 python tools/r4/atomic_vs_ordered.py 160000 300 4 > $O/f64_vs_ordered.json 2> $O/avo.err; tail -1 $O/f64_vs_ordered.json; tail -2 $O/avo.err
timeout 200 python tools/r4/atomic_vs_ordered.py 20011 40 3 2>&1 | tail -1
run() { local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
L=/root/repo/ssqueezepy_amd
for rep in 1 2; do
run ordered SSQ_TILE_ORDER=ordered
run f64-nw16 SSQ_TILE_ORDER=f64 SSQ_TILE_NW=16
run f64-nw12 SSQ_TILE_ORDER=f64 SSQ_TILE_NW=12
run f64-w16-nw16 SSQ_TILE_ORDER=f64 SSQ_TILE_NW=16 SSQ_HIP_LIB=$L/libssq_hip_w16.so
done 2>&1 | tee $O/ab.txt
SSQ_TILE_ORDER=f64 SSQ_HIP_LIB=$L/libssq_hip_prof.so SSQ_TILE2_PROF_DUMP=1 timeout 100 python bench.py --no-cpu --steps 1 --warmup 1 > $O/b.json 2> $O/prof.err
grep "tile2 prof" $O/prof.err | tail -16 > $O/prof.txt; cat $O/prof.txt
