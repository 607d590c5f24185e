#!/bin/bash
# A/B of an environment switch inside one box: the tile / config tests, then the bench with the switch on / off
# usage: SWITCH=SSQ_DEBUG_CWT_NYQ_EXT bash tools/r3_gpu_v.sh   (C5=1 adds the float64 config, both ways)
cd /root/repo; mkdir -p gpurun_out/r3v
SW=${SWITCH:-SSQ_DEBUG_CWT_NYQ_EXT}
timeout 400 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_transforms.py tests/test_gpu_00_configs.py -x -q -m gpu -k "${TESTS:-every_instantiation or few_scales or launch_group or config2 or default_arguments or lean or block_fast_path or paddings}" 2>&1 | tail -3 | cut -c1-300
for i in 1 2; do for v in ${VALS:-1 0}; do
  echo -n "$SW=$v "; env $SW=$v timeout 60 python bench.py --no-cpu --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()}, d['config'].get('algo'))"
done; done | tee gpurun_out/r3v/ab.txt
if [ "${C5:-0}" = 1 ]; then for v in ${VALS:-1 0}; do echo -n "C5 $SW=$v "; env $SW=$v timeout 200 python tools/run_configs.py c5 2>&1 | tail -1 | cut -c1-300; done | tee -a gpurun_out/r3v/ab.txt; fi
