#!/bin/bash
cd /root/repo; O=gpurun_out/r4s; mkdir -p $O; rm -f gpurun_out/parity_measured.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite.txt 2>&1; tail -5 $O/gpu_suite.txt | cut -c1-300
cp gpurun_out/parity_measured.jsonl $O/ 2>/dev/null; cat $O/parity_measured.jsonl | cut -c1-400
