#!/bin/bash
# round 4: extract_ridges over a batch -- tests, then one transform against 16 in one call (N = 40 000)
cd /root/repo; O=gpurun_out/r4r2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ridges.py -x -q -m gpu 2>&1 | tail -2 | cut -c1-200
RIDGE_N=40000 timeout 600 python tools/run_configs.py ridges 2>&1 | grep config | tee $O/ridges.txt
