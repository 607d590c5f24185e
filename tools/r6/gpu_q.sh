#!/bin/bash
# round 6: float64 block kernel with one transform in registers at a time: parity tests + config 5 timing
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6q; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "float64 or config5 or block_fast_path" 2>&1 | tail -3
timeout 500 python tools/run_configs.py c5 2>/dev/null | tee $O/c5.jsonl | cut -c1-300
