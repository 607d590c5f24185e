#!/bin/bash
cd /root/repo; O=gpurun_out/r4p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES -d /root/repo/$O/pmc/pass1 -o pmc --output-format csv -- python /root/repo/bench.py --no-cpu --steps 2 --warmup 1 > /root/repo/$O/pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU -d /root/repo/$O/pmc/pass2 -o pmc --output-format csv -- python /root/repo/bench.py --no-cpu --steps 2 --warmup 1 > /root/repo/$O/pmc2.log 2>&1
cd /root/repo; python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1; grep -A18 "tile2_kernel" $O/pmc_summary.txt | head -20
rm -rf $O/pmc/*/
