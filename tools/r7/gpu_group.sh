#!/bin/bash
# round 6 (second session): signals per launch group (SSQ_DEBUG_CWT_GROUP) at the final stamp, 16 and 64 signals per step; one box, three rounds
cd /root/repo; O=gpurun_out/r7u; mkdir -p $O
run() { label=$1; shift; echo -n "$label "; timeout 200 python bench.py --no-cpu "$@" 2>$O/err_$label.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3))"; }
for rep in 1 2 3; do
for g in 16 8 4; do
SSQ_DEBUG_CWT_GROUP=$g run b16g$g --batch 16 --steps 20
done
for g in 16 8; do
SSQ_DEBUG_CWT_GROUP=$g run b64g$g --batch 64 --steps 5
done
done 2>&1 | tee $O/ab.txt
