#!/bin/bash
# round 5: config 3 (ssq_stft n_fft 1024, hop 256) at 512 signals: shape of the bin-map reassignment (accumulate_f64_kernel)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5n; mkdir -p $O
for cfg in "A=0" "SSQ_DEBUG_ACC64_NW=16" "SSQ_DEBUG_ACC64_NW=4" "SSQ_DEBUG_ACC64_COLS=8" "SSQ_DEBUG_ACC64_COLS=8 SSQ_DEBUG_ACC64_NW=16" "SSQ_DEBUG_ACC64_COLS=8 SSQ_DEBUG_ACC64_NW=4" "SSQ_DEBUG_STFT_FUSED_TX=1" "A=0"; do
  echo -n "$cfg : "; env $cfg timeout 200 python tools/probes/c3_parts_probe.py 2>/dev/null
done 2>&1 | tee $O/ab.txt
