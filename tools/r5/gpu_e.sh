#!/bin/bash
# round 5: signals per step / per launch group
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5j; mkdir -p $O
run() { echo -n "$* : "; env "$@" timeout 300 python bench.py --no-cpu --steps 10 --warmup 3 --batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for B in 16 32 64; do export B; run SSQ_DEBUG_CWT_GROUP=16; done 2>&1 | tee $O/ab.txt
B=32; export B; run SSQ_DEBUG_CWT_GROUP=32 | tee -a $O/ab.txt
B=64; export B; run SSQ_DEBUG_CWT_GROUP=32 | tee -a $O/ab.txt
B=64; export B; run SSQ_DEBUG_CWT_GROUP=8 | tee -a $O/ab.txt
B=16; export B; run SSQ_DEBUG_CWT_GROUP=16 | tee -a $O/ab.txt
