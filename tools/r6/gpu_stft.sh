#!/bin/bash
# round 6: rocprofv3 evidence for the STFT side -- config 3 at 512 signals per step and ssq_stft at hop 1 (n_fft = 598: the
# reference's published shape, examples/benchmarks.py:78-82; 1024): kernel trace + stats, then the PMC passes
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6s}; mkdir -p $O
timeout 300 python tools/probes/c3_batched_probe.py 512 2>/dev/null | tee $O/c3_b512.txt
timeout 300 python tools/stft_hop1_probe.py 598 1024 2>/dev/null | tee $O/stft_hop1.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python tools/probes/c3_batched_probe.py 512 > $O/prof_c3.log 2>&1
DB=$(find $O/prof_c3 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_c3.txt | head -12 | cut -c1-170
rm -rf $O/prof_c3
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_h1 -o h1 -- python tools/stft_hop1_probe.py 598 1024 > $O/prof_h1.log 2>&1
DB=$(find $O/prof_h1 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_stft_hop1.txt | head -12 | cut -c1-170
rm -rf $O/prof_h1
bash tools/pmc_cmd.sh $O/pmc_c3 python tools/probes/c3_batched_probe.py 512 > $O/pmc_c3.log 2>&1
python tools/pmc_summary.py $O/pmc_c3 > $O/pmc_summary_c3.txt 2>&1; rm -rf $O/pmc_c3/*/
bash tools/pmc_cmd.sh $O/pmc_h1 python tools/stft_hop1_probe.py 598 1024 > $O/pmc_h1.log 2>&1
python tools/pmc_summary.py $O/pmc_h1 > $O/pmc_summary_stft_hop1.txt 2>&1; rm -rf $O/pmc_h1/*/
head -60 $O/pmc_summary_c3.txt
