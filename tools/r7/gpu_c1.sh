#!/bin/bash
# round 6 (second session): config 1 (cwt, N = 10 000) with the pre-stage in one launch (small_prestage_kernel) and the block spectra of every class
# in one launch (block_spectra_multi_kernel); nopre: without the first; 4096: without both (the P = 4096-only spectra kernel, as before)
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${OUT:-r7b}; mkdir -p $O
for rep in 1 2; do
  for mode in own nopre 4096; do
    ( [ $mode = 4096 ] && export SSQ_DEBUG_BLOCK_SPECTRA=4096; [ $mode = nopre ] && export SSQ_DEBUG_SMALL_PRE=0
      echo -n "$mode "; timeout 200 python tools/run_configs.py c1 2>$O/err_$mode.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms']*1e3,1), 'us')" )
  done
done 2>&1 | tee $O/c1.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o c1 -- python tools/run_configs.py c1 > $O/prof_c1.log 2>&1
DB=$(find $O/prof_c1 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_c1.txt | head -16 | cut -c1-150
rm -rf $O/prof_c1
timeout 900 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_transforms.py tests/test_gpu_edge_cases.py -q -m gpu -x 2>&1 | tail -4 | tee $O/tests.txt
