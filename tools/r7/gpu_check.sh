#!/bin/bash
# round 6 (second session): config 1 after the one-launch block spectra + the Wx-only block kernels, the bench line, the whole GPU suite
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${OUT:-r7h}; mkdir -p $O
for rep in 1 2; do
  for mode in own 4096; do
    ( [ $mode = 4096 ] && export SSQ_DEBUG_BLOCK_SPECTRA=4096
      echo -n "$mode "; timeout 200 python tools/run_configs.py c1 2>$O/err_$mode.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms']*1e3,1), 'us')" )
  done
done 2>&1 | tee $O/c1.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o c1 -- python tools/run_configs.py c1 > $O/prof_c1.log 2>&1
DB=$(find $O/prof_c1 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_c1.txt | head -12 | cut -c1-150
rm -rf $O/prof_c1
timeout 300 python bench.py --no-cpu --steps 20 > $O/bench.json 2>$O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value']), round(d['roofline']['frac'],4), d['stages_us_per_transform'])"
timeout 300 python bench.py --no-cpu --steps 200 --warmup 20 --batch 1 > $O/bench_b1.json 2>$O/bench_b1.err; python -c "import json; d=json.load(open('$O/bench_b1.json')); print('bench B=1', round(d['value']), round(d['ms_per_step'],4))"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4 | tee $O/tests.txt
