#!/bin/bash
# round 6: the STFT after the padding moved into the fused kernel: its GPU tests, config 3 at 1 / 64 / 512 signals, hop 1
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r6p; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x -k "stft" 2>&1 | tail -3
timeout 300 python tools/probes/c3_batched_probe.py 512 2>/dev/null | tee $O/c3_b512.txt
timeout 300 python tools/probes/c3_batched_probe.py 64 2>/dev/null | tee -a $O/c3_b512.txt
timeout 300 python tools/stft_hop1_probe.py 598 1024 2>/dev/null | tee $O/stft_hop1.txt
