#!/bin/bash
# round 4: tile2_kernel -- balance of the row blocks, launch group size, store ablation
cd /root/repo; O=gpurun_out/r4h; mkdir -p $O
run() { # label env...
  local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
}
L=/root/repo/ssqueezepy_amd
for rb in 0.4 0.7 1.0 1.5 2.0; do run rb$rb SSQ_TILE2_RB_COST=$rb; done 2>&1 | tee $O/ab.txt
run rb1-nw12 SSQ_TILE2_RB_COST=1.0 SSQ_TILE_NW=12 | tee -a $O/ab.txt
for g in 8 4 2 1; do run rb1-group$g SSQ_TILE2_RB_COST=1.0 SSQ_DEBUG_CWT_GROUP=$g; done 2>&1 | tee -a $O/ab.txt
run rb1-nostore SSQ_TILE2_RB_COST=1.0 SSQ_HIP_LIB=$L/libssq_hip_nost.so | tee -a $O/ab.txt
for g in 4 1; do run ordered-group$g SSQ_TILE_ORDER=ordered SSQ_DEBUG_CWT_GROUP=$g; done 2>&1 | tee -a $O/ab.txt
