#!/bin/bash
# round 4: phases of an item in tile2_kernel (shader-clock sums per wavefront of one workgroup)
cd /root/repo; O=gpurun_out/r4k; mkdir -p $O
export SSQ_TILE2_RB_COST=0.7
SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_prof.so SSQ_TILE2_PROF_DUMP=1 timeout 100 python bench.py --no-cpu --steps 1 --warmup 1 > $O/b.json 2> $O/prof.err
grep "tile2 prof" $O/prof.err | tail -16 > $O/prof.txt; cat $O/prof.txt
python -c "import json; d=json.load(open('$O/b.json')); print(round(d['value']), d['stages_us_per_transform'])"
