#!/bin/bash
cd /root/repo; O=gpurun_out/r4r; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_transforms.py tests/test_gpu_00_configs.py tests/test_gpu_cabi_c.py -x -q -m gpu -k "nyquist_rows or default_arguments or two_ranks" 2>&1 | tail -2 | cut -c1-200
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps 8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
run log
run lp --scales log-piecewise
SSQ_TILE_ORDER=ordered run lp-ordered --scales log-piecewise
run log-again
