#!/bin/bash
# round 4: tile2_kernel shapes on one box: 32-column tile x 16 wavefronts (one workgroup per CU) against 16-column
# tiles -- 16 wavefronts, and 8 wavefronts x two workgroups per CU (one's tile end overlaps the other's arithmetic)
cd /root/repo; O=gpurun_out/r4w; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps 8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for rep in 1 2; do
run "cols32/nw16"
SSQ_DEBUG_TILE2_COLS=16 run "cols16/nw16"
SSQ_DEBUG_TILE2_COLS=16 SSQ_TILE_NW=8 run "cols16/nw8x2"
SSQ_DEBUG_TILE2_COLS=16 SSQ_TILE_NW=8 SSQ_TILE2_RB_COST=0.5 run "cols16/nw8x2/rb0.5"
SSQ_DEBUG_TILE2_COLS=16 SSQ_TILE_NW=8 SSQ_TILE2_RB_COST=1.0 run "cols16/nw8x2/rb1.0"
done 2>&1 | tee $O/ab.txt
SSQ_DEBUG_TILE2_COLS=16 SSQ_TILE_NW=8 timeout 900 python -m pytest tests/test_gpu_00_configs.py -x -q -m gpu -k "config2_ssq" 2>&1 | tail -2 | cut -c1-200
