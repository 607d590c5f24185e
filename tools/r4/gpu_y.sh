#!/bin/bash
# round 4: cost of a row read back next to an interpolated one in the wavefronts' row blocks (tile2_kernel), after the
# interpolated rows' path got shorter
cd /root/repo; O=gpurun_out/r4y2; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps 8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for rb in 0.7 0.5 0.85 1.0 1.2 0.7; do SSQ_TILE2_RB_COST=$rb run "rb_cost=$rb"; done 2>&1 | tee $O/ab.txt
SSQ_TILE_NW=12 run "nw=12" | tee -a $O/ab.txt
