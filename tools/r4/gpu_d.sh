#!/bin/bash
# round 4: is the tile kernel bound inside the CU or by the memory system? -- fewer persistent workgroups; available counters
cd /root/repo; O=gpurun_out/r4d; mkdir -p $O
run() { # label env...
  local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
}
L=/root/repo/ssqueezepy_amd
for g in 256 224 192 160 128 96 64; do run grid$g SSQ_DEBUG_TILE_GRID=$g; done 2>&1 | tee $O/grid.txt
for g in 256 192 128 64; do run e17-grid$g SSQ_DEBUG_TILE_GRID=$g SSQ_HIP_LIB=$L/libssq_hip_e17.so SSQ_TILE_ORDER=atomic; done 2>&1 | tee -a $O/grid.txt
export TMPDIR=/tmp
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
grep -c . $O/list_avail.txt
