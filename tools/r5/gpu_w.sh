#!/bin/bash
# round 5: why one launch for the fused form's block classes is faster: per-class kernels at three wavefronts per SIMD
# (the multi-class kernel's occupancy), and the multi-class launch walked from its heaviest items
cd /root/repo; O=gpurun_out/${OUT:-r5w}; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps ${STEPS:-8} "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for i in 1 2; do
  run "multi"
  SSQ_DEBUG_CWT_BLOCKS_MULTI=0 run "per-class"
  SSQ_DEBUG_CWT_BLOCKS_MULTI=0 SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_w3.so run "per-class, 3 waves"
  SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_rev.so run "multi, reversed"
done 2>&1 | tee -a $O/ab.txt
