#!/bin/bash
# round 4: first run of tile2_kernel (float64 tile, unordered ds_add_f64) on the MI355X
cd /root/repo; O=gpurun_out/r4f; mkdir -p $O
timeout 200 python tools/r4/atomic_vs_ordered.py 160000 300 4 > $O/f64_vs_ordered.json 2> $O/avo.err; tail -1 $O/f64_vs_ordered.json; tail -3 $O/avo.err
timeout 200 python tools/r4/atomic_vs_ordered.py 20011 40 3 2>&1 | tail -1
timeout 200 python tools/r4/atomic_vs_ordered.py 100003 456 2 2>&1 | tail -1
run() { # label env...
  local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
}
for rep in 1 2; do
run ordered SSQ_TILE_ORDER=ordered
run f64-nw16 SSQ_TILE_NW=16
run f64-nw12 SSQ_TILE_NW=12
run f64-nw8 SSQ_TILE_NW=8
done 2>&1 | tee $O/ab.txt
run f64-lp A=1 --scales log-piecewise
