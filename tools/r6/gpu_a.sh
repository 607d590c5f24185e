#!/bin/bash
# round 6, first call: tile3_kernel (column pair per lane) on the MI355X -- the full-size config-2 tests (parity, the 48 M
# bin indices), then the short bench: tile2_kernel, tile3_kernel with 16 wavefronts (one resident class), with 12 (two)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_00_configs.py -q -m gpu -x -k "config2" > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps ${STEPS:-10} "$@" 2>$O/err_$label.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()}, d['config'].get('tile_kernel'))"; }
for rep in 1 2; do
SSQ_DEBUG_TILE_PAIR=0 run tile2
run pair16
SSQ_TILE3_NW=12 run pair12
done 2>&1 | tee -a $O/ab.txt
