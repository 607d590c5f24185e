#!/bin/bash
# round-3 call A: tile-path parity first, then the bench under the tuning switches (one box)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r3a}; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_edge_cases.py tests/test_gpu_transforms.py -x -q -m gpu \
   -k "config2 or default_arguments or every_instantiation or few_scales or launch_group or lean or full_size" > $O/pytest_tiles.log 2>&1
tail -5 $O/pytest_tiles.log
for nw in 16 12 8; do
  SSQ_TILE_NW=$nw timeout 300 python bench.py --no-cpu --steps 10 > $O/bench_nw$nw.json 2> $O/bench_nw$nw.err
  python - <<PY
import json
try:
    d=json.load(open('$O/bench_nw$nw.json')); print('nw$nw', round(d['value']), d['stages_us_per_transform'])
except Exception as e: print('nw$nw failed', e)
PY
done
SSQ_TILE_NW=16 timeout 300 python bench.py --no-cpu --steps 10 --scales log-piecewise > $O/bench_lp.json 2> $O/bench_lp.err
python -c "import json; d=json.load(open('$O/bench_lp.json')); print('log-piecewise', round(d['value']), d['stages_us_per_transform'])"
SSQ_CWT_TILES=0 timeout 300 python bench.py --no-cpu --steps 10 > $O/bench_notiles.json 2> $O/bench_notiles.err
python -c "import json; d=json.load(open('$O/bench_notiles.json')); print('notiles', round(d['value']), d['stages_us_per_transform'])"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r3a -- python bench.py --no-cpu --steps 5 > $O/prof.log 2>&1
ls $O/prof | head
DB=$(find $O/prof -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats.txt | head -30
