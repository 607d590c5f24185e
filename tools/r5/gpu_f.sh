#!/bin/bash
# round 5: least decimation for which a row leaves the block kernels (SSQ_DEBUG_TILE_RMIN), on today's kernels
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5l; mkdir -p $O
for r in 4 2 8 4; do
  echo -n "SSQ_DEBUG_TILE_RMIN=$r : "; SSQ_DEBUG_TILE_RMIN=$r timeout 300 python bench.py --no-cpu --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
done 2>&1 | tee $O/ab.txt
