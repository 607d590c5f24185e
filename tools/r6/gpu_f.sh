#!/bin/bash
# round 6: the dealt-lists planner of tile3_kernel -- plan print, bench variants, phase stamps
cd /root/repo; O=gpurun_out/r6f; mkdir -p $O
SSQ_DEBUG_TILE_PLAN_PRINT=1 timeout 200 python bench.py --no-cpu --steps 2 --warmup 1 2>&1 >/dev/null | grep "tile3 wave" | head -16 | tee $O/plan16.txt
OUT=r6f bash tools/r6/gpu_d.sh tile2:SSQ_DEBUG_TILE_PAIR=0 pair16 pair12:SSQ_TILE3_NW=12 pair16rb35:SSQ_DEBUG_TILE3_RB=0.35 pair16rb55:SSQ_DEBUG_TILE3_RB=0.55 pair16b pair12b:SSQ_TILE3_NW=12
export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_prof.so
timeout 300 python tools/r6/tile_prof.py 16 2>&1 | grep -v amdgpu.ids | tee $O/prof16.txt | cut -c1-200
SSQ_TILE3_NW=12 timeout 300 python tools/r6/tile_prof.py 16 2>&1 | grep -v amdgpu.ids | tee $O/prof12.txt | cut -c1-200
