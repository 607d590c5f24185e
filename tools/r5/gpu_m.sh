#!/bin/bash
# round 5: the four-step intermediates in sub-batches of n signals (Y of a sub-batch in the Infinity Cache?)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5v; mkdir -p $O
for n in 0 4 2 8 0; do
  echo -n "SSQ_TILE_FFT_SUB=$n : "; SSQ_TILE_FFT_SUB=$n timeout 300 python bench.py --no-cpu --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
done 2>&1 | tee $O/ab.txt
