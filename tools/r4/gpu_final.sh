#!/bin/bash
# one box: the whole GPU suite, the bench lines (log, log-piecewise, ordered kernel, 456 rows, batch 64), rocprof kernel
# stats, PMC passes, the other BASELINE configs.   RUNSHA=<git sha> RUNTAG=r4z bash tools/r4/gpu_final.sh
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r4z}; mkdir -p $O; rm -f gpurun_out/parity_measured.jsonl
echo "${RUNSHA:-unknown}" > $O/git_sha.txt
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt | cut -c1-300
cp gpurun_out/parity_measured.jsonl $O/ 2>/dev/null
timeout 300 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value']), d['roofline']['frac'], d['stages_us_per_transform'], d.get('cpu_baseline',{}).get('value'))"
timeout 120 python bench.py --steps 10 --no-cpu --scales log-piecewise > $O/bench_lp.json 2> $O/bench_lp.err; python -c "import json; d=json.load(open('$O/bench_lp.json')); print('bench lp', round(d['value']), d['stages_us_per_transform'])"
SSQ_TILE_ORDER=ordered timeout 120 python bench.py --steps 10 --no-cpu > $O/bench_ordered.json 2> $O/bench_ordered.err; python -c "import json; d=json.load(open('$O/bench_ordered.json')); print('bench ordered', round(d['value']), d['stages_us_per_transform'])"
timeout 200 python bench.py --steps 6 --no-cpu --na 456 > $O/bench_na456.json 2> $O/bench_na456.err; python -c "import json; d=json.load(open('$O/bench_na456.json')); print('bench na456', round(d['value']), d['stages_us_per_transform'], d['config']['algo'], d['config']['tile_kernel'])"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r4 -- python bench.py --no-cpu --steps 5 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats.txt | head -14 | cut -c1-160
rm -rf $O/prof
bash tools/pmc_collect.sh $O/pmc --steps 3 > $O/pmc.log 2>&1
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
python tools/pmc_traffic.py $O/pmc $(( (3+3+3) * 16 )) $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1; tail -3 $O/pmc_traffic.txt
rm -rf $O/pmc/*/
timeout 200 python bench.py --no-cpu --steps 4 --warmup 3 --batch 64 > $O/bench_b64.json 2> $O/bench_b64.err; python -c "import json; d=json.load(open('$O/bench_b64.json')); print('B=64', round(d['value']), d['ms_per_step'])"
timeout 400 python tools/run_configs.py c1 c3 c5 > $O/configs.jsonl 2> $O/configs.err; cut -c1-220 $O/configs.jsonl
# what a small transform's time is made of (C1: N = 10 000): kernel time against the wall time run_configs reports
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o c1 -- python tools/run_configs.py c1 > $O/prof1.log 2>&1
DB=$(find $O/prof1 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_c1.txt | head -12 | cut -c1-160
rm -rf $O/prof1
