#!/bin/bash
# round 5: the GPU suite in the other modes (ordered sums; fused ssq_cwt without the tile path)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5r; mkdir -p $O
SSQ_TILE_ORDER=ordered timeout 1500 python -m pytest tests -q -m gpu -x > $O/suite_ordered.txt 2>&1; tail -2 $O/suite_ordered.txt | cut -c1-200
SSQ_CWT_TILES=0 timeout 1500 python -m pytest tests -q -m gpu -x -k "not tile and not bin_indices and not more_rows and not default_arguments" > $O/suite_notiles.txt 2>&1; tail -2 $O/suite_notiles.txt | cut -c1-200
