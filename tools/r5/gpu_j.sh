#!/bin/bash
# round 5: the ordered tile kernel with its first tiles permuted per XCD (tests in the ordered mode + bench)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5q; mkdir -p $O
SSQ_TILE_ORDER=ordered timeout 900 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_edge_cases.py -q -m gpu -x -k "config2_ssq or tile or default_arguments" > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-200
for i in 1 2; do echo -n "ordered : "; SSQ_TILE_ORDER=ordered timeout 200 python bench.py --no-cpu --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; done | tee $O/ab.txt
