#!/bin/bash
# round 6: SQ counters of the current tile3_kernel (instruction mix, waits, LDS) + HBM traffic passes
cd /root/repo; export TMPDIR=/tmp
O=$PWD/gpurun_out/r6k; mkdir -p $O
bash tools/pmc_sq.sh $O/p16 --steps 3 --warmup 1 > $O/p16.log 2>&1
cd /root/repo
python tools/pmc_summary.py $O/p16 | grep -A26 "tile3_kernel" > $O/pmc_tile3.txt
python tools/pmc_summary.py $O/p16 | grep -A26 "blockzoom_multi" > $O/pmc_blockzoom.txt
rm -rf $O/p16
cat $O/pmc_tile3.txt
