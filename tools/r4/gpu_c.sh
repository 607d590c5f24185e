#!/bin/bash
# round 4, third GPU call: transposed weight table, stores behind the next step's loads, LDS-only hand-over
cd /root/repo; O=gpurun_out/r4c; mkdir -p $O
run() { # label env...
  local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
}
L=/root/repo/ssqueezepy_amd
timeout 300 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "config2 or every_instantiation or few_scales" 2>&1 | tail -2 | cut -c1-200
SSQ_HIP_LIB=$L/libssq_hip_bothfencepipe.so timeout 300 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "config2 or every_instantiation or few_scales" 2>&1 | tail -2 | cut -c1-200
for rep in 1 2; do
run old SSQ_HIP_LIB=$L/libssq_hip_old.so
run wsoa SSQ_HIP_LIB=$L/libssq_hip_wsoa.so
run late SSQ_HIP_LIB=$L/libssq_hip_late.so
run both A=1
run both-nw16 SSQ_TILE_NW=16
run bothpipe SSQ_HIP_LIB=$L/libssq_hip_bothpipe.so
run bothfence SSQ_HIP_LIB=$L/libssq_hip_bothfence.so
run bothfencepipe SSQ_HIP_LIB=$L/libssq_hip_bothfencepipe.so
run both-nostore SSQ_HIP_LIB=$L/libssq_hip_both256.so
done 2>&1 | tee $O/ab.txt
