#!/bin/bash
# round 6: SQ counters of tile3_kernel (16 wavefronts / one resident class; 12 / two) -- instruction mix, waits, LDS
cd /root/repo; export TMPDIR=/tmp
O=$PWD/gpurun_out/r6b; mkdir -p $O
bash tools/pmc_sq.sh $O/p16 --steps 3 --warmup 1 > $O/p16.log 2>&1
SSQ_TILE3_NW=12 bash tools/pmc_sq.sh $O/p12 --steps 3 --warmup 1 > $O/p12.log 2>&1
cd /root/repo
python tools/pmc_summary.py $O/p16 | grep -A26 "tile3_kernel" > $O/pmc_pair16.txt
python tools/pmc_summary.py $O/p12 | grep -A26 "tile3_kernel" > $O/pmc_pair12.txt
rm -rf $O/p16 $O/p12
cat $O/pmc_pair16.txt; cat $O/pmc_pair12.txt
