#!/bin/bash
# round 4: tile2_kernel's row blocks sized by the wavefronts' measured speeds (older wavefronts of a SIMD win the issue
# arbitration): speed factors per group of four wavefronts
cd /root/repo; O=gpurun_out/r4ws; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps 8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for ws in "1,1,1,1" "1,1,0.91,0.84" "1,1,0.85,0.75" "1.05,1,0.9,0.8" "1,1,0.95,0.9" "1,1,1,1" "1,1,0.91,0.84"; do SSQ_TILE2_WAVE_SPEED=$ws run "speed=$ws"; done 2>&1 | tee $O/ab.txt
SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_prof.so true
