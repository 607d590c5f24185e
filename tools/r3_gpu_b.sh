#!/bin/bash
# one box: parity of the tile path, A/B variants, trace
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r3d}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_edge_cases.py -x -q -m gpu \
   -k "config2 or default_arguments or every_instantiation or few_scales or launch_group" > $O/pytest_tiles.log 2>&1
tail -3 $O/pytest_tiles.log
NWS="${NWS:-16 12}" bash tools/ab_run.sh ${VARIANTS:-base d2 d8 pri noprod} | tee $O/ab.txt
SSQ_TILE_TRACE=$O/trace_base.bin timeout 100 python bench.py --no-cpu --steps 3 > $O/b1.json 2>$O/b1.err
BENCH_ARGS="--scales log-piecewise" NWS=16 bash tools/ab_run.sh base | tee -a $O/ab.txt
