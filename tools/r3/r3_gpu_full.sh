#!/bin/bash
# one box: the whole GPU suite, the bench lines (log and log-piecewise), rocprof kernel stats, PMC passes
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r3full}; mkdir -p $O; rm -f gpurun_out/parity_measured.jsonl
timeout 900 python -m pytest tests -q -m gpu -x > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt | cut -c1-300
cp gpurun_out/parity_measured.jsonl $O/ 2>/dev/null
timeout 300 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value']), d['roofline']['frac'], d['stages_us_per_transform'], d.get('cpu_baseline',{}).get('value'))"
timeout 120 python bench.py --steps 10 --no-cpu --scales log-piecewise > $O/bench_lp.json 2> $O/bench_lp.err; python -c "import json; d=json.load(open('$O/bench_lp.json')); print('bench lp', round(d['value']), d['stages_us_per_transform'])"
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r3 -- python bench.py --no-cpu --steps 5 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats.txt | head -12
if [ "${PMC:-1}" = 1 ]; then
  bash tools/pmc_collect.sh $O/pmc --steps 3 > $O/pmc.log 2>&1
  python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
  python tools/pmc_traffic.py $O/pmc $(( (3+3+3) * 16 )) $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1; tail -3 $O/pmc_traffic.txt
  rm -rf $O/pmc/*/  # raw csv dirs are large
fi
