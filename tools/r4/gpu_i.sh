#!/bin/bash
# round 4: is the tile kernels' cost per (row x tile) visit an address-translation cost? The same number of
# points with rows 160 KB apart (N = 20 000, 128 signals) instead of 1.28 MB (N = 160 000, 16 signals)
cd /root/repo; O=gpurun_out/r4i; mkdir -p $O
run() { # label args... (env via ENVV)
  local label=$1; shift
  echo -n "$label "; env $ENVV timeout 200 python bench.py --no-cpu --steps 4 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
}
ENVV="SSQ_TILE_ORDER=ordered" run ordered-N160k --n 160000 --batch 16
ENVV="SSQ_TILE_ORDER=ordered" run ordered-N20k --n 20000 --batch 128
ENVV="SSQ_TILE_ORDER=ordered" run ordered-N40k --n 40000 --batch 64
ENVV="SSQ_TILE2_RB_COST=0.7" run f64-N160k --n 160000 --batch 16
ENVV="SSQ_TILE2_RB_COST=0.7" run f64-N20k --n 20000 --batch 128
ENVV="SSQ_TILE2_RB_COST=0.7" run f64-N40k --n 40000 --batch 64
