#!/bin/bash
# round 4: tile2_kernel with every load outside divergent branches (exact vmcnt waits), 4-slot ring
cd /root/repo; O=gpurun_out/r4g; mkdir -p $O
timeout 200 python tools/r4/atomic_vs_ordered.py 160000 300 4 > $O/f64_vs_ordered.json 2> $O/avo.err; tail -1 $O/f64_vs_ordered.json
timeout 200 python tools/r4/atomic_vs_ordered.py 20011 40 3 2>&1 | tail -1
run() { # label env...
  local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
}
L=/root/repo/ssqueezepy_amd
for rep in 1 2; do
run ordered SSQ_TILE_ORDER=ordered
run f64-nw16 SSQ_TILE_NW=16
run f64-nw12 SSQ_TILE_NW=12
run f64-nw8 SSQ_TILE_NW=8
run f64-wt2-nw12 SSQ_TILE_NW=12 SSQ_HIP_LIB=$L/libssq_hip_wt2.so
run f64-wt2-nw16 SSQ_TILE_NW=16 SSQ_HIP_LIB=$L/libssq_hip_wt2.so
done 2>&1 | tee $O/ab.txt
