#!/bin/bash
# round 4: the float64-tile accumulate kernel (bin map) -- suite, C3/C5 timings, A/B of its shapes
cd /root/repo; O=gpurun_out/r4s; mkdir -p $O; rm -f gpurun_out/parity_measured.jsonl
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite.txt 2>&1; tail -5 $O/gpu_suite.txt | cut -c1-300
cp gpurun_out/parity_measured.jsonl $O/ 2>/dev/null
for v in "" "SSQ_TILE_ORDER=ordered" "SSQ_DEBUG_ACC64_NW=16" "SSQ_DEBUG_ACC64_NW=4" "SSQ_DEBUG_ACC64_COLS=16" "SSQ_DEBUG_ACC64_COLS=32"; do
  echo "== $v" >> $O/configs.txt
  env $v timeout 600 python tools/run_configs.py c3 c5 >> $O/configs.txt 2>&1
done
cut -c1-220 $O/configs.txt
for v in "SSQ_CWT_TILES=0" "SSQ_CWT_TILES=0 SSQ_TILE_ORDER=ordered" ""; do
  echo "== bench $v"; env $v timeout 300 python bench.py --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-400
done > $O/bench.txt 2>&1
cat $O/bench.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o c5 -- python tools/run_configs.py c3 c5 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats.txt | head -24 | cut -c1-160
