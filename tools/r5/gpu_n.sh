#!/bin/bash
# round 5: tile2_kernel with the exact path's parameters read from the kernel-argument segment in the rare branch
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5y; mkdir -p $O
SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_spmem.so timeout 600 python -m pytest tests/test_gpu_00_configs.py -q -m gpu -x -k "bin_indices or config2_ssq" 2>&1 | tail -2 | cut -c1-200
OUT=r5y bash tools/r5/gpu_v.sh default spmem default spmem
