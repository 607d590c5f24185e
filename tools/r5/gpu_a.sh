#!/bin/bash
# round 5, first call: the new tests (bin dump at full size, 456 rows in both modes, per-row parity), the bench
# line at 16 and at 1 signal per step, what a single C2 transform's time is made of (rocprofv3 kernel trace, B = 1)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_edge_cases.py -q -m gpu -x \
   -k "bin_indices or more_rows or bench_seeds or config2_ssq" > $O/new_tests.txt 2>&1; tail -4 $O/new_tests.txt | cut -c1-300
timeout 300 python bench.py --steps 20 --no-cpu > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print('bench B=16', round(d['value']), round(d['roofline']['frac'],4), d['stages_us_per_transform'], d['roofline']['traffic_stale'], d['config']['build_sha'])"
timeout 300 python bench.py --steps 200 --warmup 20 --batch 1 --no-cpu > $O/bench_b1.json 2> $O/bench_b1.err
python -c "import json; d=json.load(open('$O/bench_b1.json')); print('bench B=1', round(d['value']), d['ms_per_step'], round(d['roofline']['frac'],4), d['stages_us_per_transform'])"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o b1 -- python bench.py --no-cpu --steps 50 --warmup 10 --batch 1 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_b1.txt | head -30 | cut -c1-170
rm -rf $O/prof
