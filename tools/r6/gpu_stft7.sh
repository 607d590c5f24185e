#!/bin/bash
# round 6: PMC passes + kernel trace of config 3 at 512 signals with the fused kernel that sums Tx itself
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y10}; mkdir -p $O
bash tools/pmc_cmd.sh $O/pmc_c3 python tools/probes/c3_batched_probe.py 512 > $O/pmc_c3.log 2>&1
python tools/pmc_summary.py $O/pmc_c3 > $O/pmc_summary_c3.txt 2>&1; rm -rf $O/pmc_c3/*/
grep -A22 "stft_fused" $O/pmc_summary_c3.txt | head -30
