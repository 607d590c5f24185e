#!/bin/bash
# one box: quick parity gate of the tile path (stop at the first failure), A/B variants, trace
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r3d}; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_edge_cases.py -x -q -m gpu -k "every_instantiation or few_scales or launch_group" > $O/pytest_gate.log 2>&1
rc=$?; tail -3 $O/pytest_gate.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "GATE FAILED rc=$rc"; exit 1; fi
timeout 60 python bench.py --no-cpu --steps 3 > $O/b0.json 2>$O/b0.err || { echo "bench failed"; tail -3 $O/b0.err | cut -c1-300; exit 1; }
NWS="${NWS:-16 12}" bash tools/ab_run.sh ${VARIANTS:-base d2 d8 pri noprod} | tee $O/ab.txt
SSQ_TILE_TRACE=$O/trace_base.bin timeout 100 python bench.py --no-cpu --steps 3 > $O/b1.json 2>$O/b1.err
BENCH_ARGS="--scales log-piecewise" NWS=16 bash tools/ab_run.sh base | tee -a $O/ab.txt
timeout 600 python -m pytest tests/test_gpu_00_configs.py -x -q -m gpu -k "config2 or default_arguments" > $O/pytest_cfg.log 2>&1; tail -3 $O/pytest_cfg.log | cut -c1-300
