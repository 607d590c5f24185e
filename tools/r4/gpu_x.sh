#!/bin/bash
# round 4: ssq_stft with Tx summed inside the fused STFT kernel -- STFT tests, C3 timings (default / ordered)
cd /root/repo; O=gpurun_out/r4x; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "stft" 2>&1 | tail -3 | cut -c1-300
for v in "" "SSQ_TILE_ORDER=ordered"; do
  echo "== $v"; env $v timeout 600 python tools/run_configs.py c3 2>&1 | grep -v WARN | cut -c1-200
done | tee $O/configs.txt
