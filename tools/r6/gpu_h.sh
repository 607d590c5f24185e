#!/bin/bash
# round 6: after the switch renames -- GPU suite in the three modes + bench lines (16 / 1 signals, default scales, ordered, 456 rows)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6h2; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite.txt 2>&1; tail -2 $O/gpu_suite.txt | cut -c1-200
SSQ_TILE_ORDER=ordered timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite_ordered.txt 2>&1; tail -1 $O/gpu_suite_ordered.txt | cut -c1-200
SSQ_CWT_TILES=0 timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite_no_tiles.txt 2>&1; tail -1 $O/gpu_suite_no_tiles.txt | cut -c1-200
OUT=r6h2 bash tools/r6/gpu_d.sh pair b1:BENCH_ARGS=--batch=1 lp:BENCH_ARGS=--scales=log-piecewise ordered:SSQ_TILE_ORDER=ordered na456:BENCH_ARGS=--na=456 pairb
