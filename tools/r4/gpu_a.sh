#!/bin/bash
# round 4, first GPU call: the atomic form of the tile kernel against the ticketed one
cd /root/repo; O=gpurun_out/r4a; mkdir -p $O
python tools/r4/atomic_vs_ordered.py 160000 300 4 > $O/atomic_vs_ordered.json 2> $O/avo.err; tail -1 $O/atomic_vs_ordered.json
SSQ_TILE_ORDER=ordered timeout 600 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "config2 or every_instantiation or few_scales or default_arguments" 2>&1 | tail -3 | cut -c1-200
run() { # label env...
  local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
}
L=/root/repo/ssqueezepy_amd
for rep in 1 2; do
run ordered SSQ_TILE_ORDER=ordered
run atomic12 SSQ_TILE_NW=12
run atomic16 SSQ_TILE_NW=16
run atomic8 SSQ_TILE_NW=8
run nowait12 SSQ_HIP_LIB=$L/libssq_hip_exp16.so SSQ_TILE_NW=12
run nowait16 SSQ_HIP_LIB=$L/libssq_hip_exp16.so SSQ_TILE_NW=16
run nowait-noadd12 SSQ_HIP_LIB=$L/libssq_hip_exp17.so SSQ_TILE_NW=12
run nowait-noarith12 SSQ_HIP_LIB=$L/libssq_hip_exp20.so SSQ_TILE_NW=12
run noarith-ordered12 SSQ_HIP_LIB=$L/libssq_hip_exp20.so SSQ_TILE_ORDER=ordered
done 2>&1 | tee $O/ab.txt
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o r4 -- python bench.py --no-cpu --steps 5 > $O/prof.log 2>&1
DB=$(find $O/prof -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats.txt | head -14 | cut -c1-160
rm -rf $O/prof
