#!/bin/bash
# round 4: shader-clock time per phase and wavefront of tile2_kernel (one workgroup, summed over a launch; -DSSQ_TILE2_PROF build)
cd /root/repo; O=gpurun_out/r4prof; mkdir -p $O
SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_prof.so SSQ_TILE2_PROF_DUMP=1 timeout 300 python bench.py --no-cpu --steps 1 --warmup 1 2> $O/prof.txt | cut -c1-100
tail -16 $O/prof.txt
