#!/bin/bash
# round 5: one signal per step (the reference's published figure is a single transform per call): launch-structure switches
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5m; mkdir -p $O
run() { echo -n "$* : "; env "$@" timeout 300 python bench.py --no-cpu --steps 200 --warmup 20 --batch 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1e3,1), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
run A=0 | tee $O/ab.txt
run SSQ_DEBUG_CWT_BLOCKS_MULTI=1 | tee -a $O/ab.txt
run SSQ_DEBUG_TILE_FFT=rocfft | tee -a $O/ab.txt
run SSQ_DEBUG_CWT_BLOCKS_MULTI=1 SSQ_DEBUG_TILE_FFT=rocfft | tee -a $O/ab.txt
run SSQ_TILE_SERIAL=1 | tee -a $O/ab.txt
run A=0 | tee -a $O/ab.txt
