#!/bin/bash
# quick check of a tile-kernel change: the tile tests, then the bench at 12 / 16 wavefronts (2 repetitions)
cd /root/repo; mkdir -p gpurun_out/r3u
timeout 300 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_transforms.py tests/test_gpu_00_configs.py -x -q -m gpu -k "every_instantiation or few_scales or launch_group or config2 or default_arguments or lean" 2>&1 | tail -2 | cut -c1-200
REPS=2 NWS="${NWS:-12 16}" bash tools/ab_run.sh ${VARIANTS:-base} | tee gpurun_out/r3u/ab.txt
