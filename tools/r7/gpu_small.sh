#!/bin/bash
# round 6 (second session): ssq_cwt of one short signal -- timing and kernel trace
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${OUT:-r7s}; mkdir -p $O
python tools/r7/ssq_small_probe.py 10000 2>$O/err.txt | tee $O/small.txt
python tools/r7/ssq_small_probe.py 40000 2>>$O/err.txt | tee -a $O/small.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o s -- python tools/r7/ssq_small_probe.py 10000 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_small.txt | head -24 | cut -c1-150
rm -rf $O/prof
