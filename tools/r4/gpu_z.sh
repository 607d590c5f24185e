#!/bin/bash
# round 4: every block class of a small transform in one launch (blockzoom_multi_kernel): tests, C1 both ways, headline unchanged
cd /root/repo; O=gpurun_out/r4zz; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_transforms.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "cwt" 2>&1 | tail -2 | cut -c1-200
SSQ_DEBUG_CWT_BLOCKS_MULTI=1 timeout 900 python -m pytest tests/test_gpu_transforms.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "cwt" 2>&1 | tail -2 | cut -c1-200
for v in "SSQ_DEBUG_CWT_BLOCKS_MULTI=0" "" "SSQ_DEBUG_CWT_BLOCKS_MULTI=0" ""; do echo "== $v"; env $v timeout 300 python tools/run_configs.py c1 2>&1 | grep config | cut -c1-160; done | tee $O/c1.txt
timeout 200 python bench.py --no-cpu --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
