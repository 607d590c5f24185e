#!/bin/bash
# round 5: after the split of the tile path into translation units and the release / acquire ticket of the ordered kernel:
# the tile tests in both modes, the bench line in both modes
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5g; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_edge_cases.py tests/test_gpu_transforms.py -q -m gpu -x \
   -k "config2 or tile or more_rows or full_size or default_arguments" > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
for mode in f64 ordered f64 ordered; do
  if [ $mode = ordered ]; then export SSQ_TILE_ORDER=ordered; else unset SSQ_TILE_ORDER; fi
  echo -n "mode=$mode "; timeout 200 python bench.py --no-cpu --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()}, d['config']['build_sha'][:12])"
done 2>&1 | tee $O/ab.txt
