#!/bin/bash
# round 4: what tile2_kernel's time is made of -- ablation builds (WRONG RESULTS) on one box. All from the
# general loop (-DSSQ_TILE2_ONELOOP); a = no Wx store, no bin arithmetic, no gather (SSQ_TILE_EXP 1792), then a plus:
# b no ds_add_f64, c no write-out stores, d no tile end at all, e no tap arithmetic, f no data loads, g all; h, i = base
# without the ds_add_f64 / tap arithmetic only
cd /root/repo; O=gpurun_out/r4u; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps 8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for v in base a b c d e f g h i base; do
  export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so
  run "lib=$v"
done 2>&1 | tee $O/ab.txt
