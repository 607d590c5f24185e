#!/bin/bash
# round 6: what tile3_kernel's time is made of -- builds with parts switched off (-DSSQ_T3_ABL=<bits>, wrong results,
# timing only; tools/ab_build.sh ssq_tile_pair abl<bits> -DSSQ_T3_ABL=<bits>): 1 gather, 2 taps, 4 modulation, 8 Wx store,
# 16 bin arithmetic, 32 LDS adds, 64 tile end, 256 priority toggle. One box; the short bench per library.
cd /root/repo; O=gpurun_out/${OUT:-r6c}; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps ${STEPS:-8} "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for v in "$@"; do
  if [ "$v" != default ]; then export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; else unset SSQ_HIP_LIB; fi
  run "lib=$v"
done 2>&1 | tee -a $O/ab.txt
