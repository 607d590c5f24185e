#!/bin/bash
# Round 6, the last device-code action: everything profiles/r6z_* holds, from ONE box, at the device code the library is
# stamped with (ssq_build_sha: the last commit that touched csrc/ or include/; every JSON line below carries it).
#   bash tools/r6/gpu_final.sh            (RUNTAG=r6z)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6z}; mkdir -p $O; rm -f gpurun_out/parity_measured.jsonl
python - > $O/build_sha.txt <<'PY'
import ctypes
l = ctypes.CDLL('/root/repo/ssqueezepy_amd/libssq_hip.so'); l.ssq_build_sha.restype = ctypes.c_char_p
print(l.ssq_build_sha().decode())
PY
echo "library: $(cat $O/build_sha.txt)"
# (evidence belongs to a committed state: rebuild AFTER committing -- the stamp is taken at build time)
if grep -q dirty $O/build_sha.txt && [ -z "$ALLOW_DIRTY" ]; then echo "the library was built from an uncommitted tree: commit, rebuild, run again (ALLOW_DIRTY=1 overrides)"; exit 3; fi
# 1. the GPU suite (full-size BASELINE-config tests first)
timeout 1500 python -m pytest tests -q -m gpu -x > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt | cut -c1-300
cp gpurun_out/parity_measured.jsonl $O/ 2>/dev/null
# 2. PMC passes of the bench command (separate --pmc passes, --kernel-trace only), HBM traffic per transform -> the
#    figure bench.py reports as roofline.traffic (profiles/pmc_traffic.json carries the library's stamp)
bash tools/pmc_collect.sh $O/pmc --steps 3 > $O/pmc.log 2>&1
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1
python tools/pmc_traffic.py $O/pmc $(( (3+3+3) * 16 )) $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1; tail -3 $O/pmc_traffic.txt
rm -rf $O/pmc/*/
cp $O/pmc_traffic.json profiles/pmc_traffic.json     # (so that the bench line below reads THIS run's traffic)
# 3. the bench lines: headline (with the CPU baseline), one signal per step, the reference's default scales, the ordered
#    kernel, 456 rows, config 4's per-GPU shape
timeout 400 python bench.py --steps 20 > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print('bench', round(d['value']), round(d['roofline']['frac'],4), d['roofline']['traffic_stale'], d['stages_us_per_transform'], d.get('cpu_baseline',{}).get('value'))"
timeout 300 python bench.py --steps 200 --warmup 20 --batch 1 --no-cpu > $O/bench_b1.json 2> $O/bench_b1.err; python -c "import json; d=json.load(open('$O/bench_b1.json')); print('bench B=1', round(d['value']), round(d['ms_per_step'],4), round(d['roofline']['frac'],4))"
timeout 120 python bench.py --steps 10 --no-cpu --scales log-piecewise > $O/bench_lp.json 2> $O/bench_lp.err; python -c "import json; d=json.load(open('$O/bench_lp.json')); print('bench lp', round(d['value']), d['stages_us_per_transform'])"
SSQ_TILE_ORDER=ordered timeout 120 python bench.py --steps 10 --no-cpu > $O/bench_ordered.json 2> $O/bench_ordered.err; python -c "import json; d=json.load(open('$O/bench_ordered.json')); print('bench ordered', round(d['value']), d['stages_us_per_transform'])"
timeout 200 python bench.py --steps 6 --no-cpu --na 456 > $O/bench_na456.json 2> $O/bench_na456.err; python -c "import json; d=json.load(open('$O/bench_na456.json')); print('bench na456', round(d['value']), d['config']['algo'])"
timeout 200 python bench.py --no-cpu --steps 4 --warmup 3 --batch 64 > $O/bench_b64.json 2> $O/bench_b64.err; python -c "import json; d=json.load(open('$O/bench_b64.json')); print('B=64', round(d['value']), d['ms_per_step'])"
# 4. rocprofv3 kernel stats of the bench command and of one signal per step
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r6 -- python bench.py --no-cpu --steps 5 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats.txt | head -14 | cut -c1-160
rm -rf $O/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof1 -o b1 -- python bench.py --no-cpu --steps 50 --warmup 10 --batch 1 > $O/prof1.log 2>&1
DB=$(find $O/prof1 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_b1.txt | head -8 | cut -c1-160
rm -rf $O/prof1
# 5. the other BASELINE configurations, each line with its own roofline block; the STFT at the reference's published shape
timeout 500 python tools/run_configs.py c1 c3 c5 > $O/configs.jsonl 2> $O/configs.err; cut -c1-260 $O/configs.jsonl
timeout 300 python tools/stft_hop1_probe.py 1024 598 2>/dev/null | tee $O/stft_hop1.txt
# 5b. the STFT side's kernel trace + counters (config 3 at 512 signals; hop 1) and config 5's kernel trace
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python tools/probes/c3_batched_probe.py 512 > $O/prof_c3.log 2>&1
DB=$(find $O/prof_c3 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_c3.txt | head -6 | cut -c1-160
rm -rf $O/prof_c3
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_h1 -o h1 -- python tools/stft_hop1_probe.py 598 1024 > $O/prof_h1.log 2>&1
DB=$(find $O/prof_h1 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_stft_hop1.txt | head -6 | cut -c1-160
rm -rf $O/prof_h1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o c5 -- python tools/run_configs.py c5 > $O/prof_c5.log 2>&1
DB=$(find $O/prof_c5 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_c5.txt | head -6 | cut -c1-160
rm -rf $O/prof_c5
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o c1 -- python tools/run_configs.py c1 > $O/prof_c1.log 2>&1
DB=$(find $O/prof_c1 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_c1.txt | head -6 | cut -c1-160
rm -rf $O/prof_c1
bash tools/pmc_cmd.sh $O/pmc_c3 python tools/probes/c3_batched_probe.py 512 > $O/pmc_c3.log 2>&1
python tools/pmc_summary.py $O/pmc_c3 > $O/pmc_summary_c3.txt 2>&1; rm -rf $O/pmc_c3/*/
# 6. the GPU suite in the other two modes, and the randomised sweep (MODES=0 skips)
if [ "${MODES:-1}" != 0 ]; then
  SSQ_TILE_ORDER=ordered timeout 1200 python -m pytest tests -q -m gpu > $O/gpu_suite_ordered_mode.txt 2>&1; tail -1 $O/gpu_suite_ordered_mode.txt
  SSQ_CWT_TILES=0 timeout 1200 python -m pytest tests -q -m gpu > $O/gpu_suite_no_tiles.txt 2>&1; tail -1 $O/gpu_suite_no_tiles.txt
  timeout 600 python tools/fuzz_parity.py 150 2025 > $O/fuzz_gpu.txt 2>&1; tail -1 $O/fuzz_gpu.txt
fi
