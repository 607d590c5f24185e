#!/bin/bash
# round 6: phase stamps of tile3_kernel (profiling build libssq_hip_prof.so = tools/ab_build.sh ssq_tile_pair prof -DSSQ_T3_PROF=1)
cd /root/repo; O=gpurun_out/r6e; mkdir -p $O
export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_prof.so
timeout 300 python tools/r6/tile_prof.py 16 2>&1 | tee $O/prof16.txt | cut -c1-200
SSQ_TILE3_NW=12 timeout 300 python tools/r6/tile_prof.py 16 2>&1 | tee $O/prof12.txt | cut -c1-200
