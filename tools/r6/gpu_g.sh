#!/bin/bash
# round 6: the GPU suite (default mode) + the bench line + per-variant bench after tile3_kernel became the default for every length
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6g2; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite.txt 2>&1; tail -5 $O/gpu_suite.txt | cut -c1-300
OUT=r6g2 bash tools/r6/gpu_d.sh tile2:SSQ_DEBUG_TILE_PAIR=0 pair pairb lp:BENCH_ARGS=--scales=log-piecewise
