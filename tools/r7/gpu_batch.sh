#!/bin/bash
# round 6 (second session): why 64 signals per step run 3.7 % faster than 16 -- signals per step x signals per launch group, one box
cd /root/repo; O=gpurun_out/r7a; mkdir -p $O
run() { label=$1; shift; echo -n "$label "; timeout 200 python bench.py --no-cpu "$@" 2>$O/err_$label.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for rep in 1 2; do
run b16 --batch 16 --steps 20
run b32 --batch 32 --steps 10
run b64 --batch 64 --steps 5
run b128 --batch 128 --steps 3
SSQ_DEBUG_CWT_GROUP=8 run b64g8 --batch 64 --steps 5
SSQ_DEBUG_CWT_GROUP=32 run b64g32 --batch 64 --steps 5
SSQ_DEBUG_CWT_GROUP=32 run b32g32 --batch 32 --steps 10
SSQ_DEBUG_CWT_GROUP=8 run b16g8 --batch 16 --steps 20
done 2>&1 | tee $O/ab.txt
