#!/bin/bash
# round 4, second GPU call: LDS read-modify-write rates; where the tile kernel's arithmetic pipeline spends its time
cd /root/repo; O=gpurun_out/r4b; mkdir -p $O
timeout 120 tools/probes/lds_atomic_probe > $O/lds_atomic_probe.txt 2>&1; cat $O/lds_atomic_probe.txt
run() { # label env...
  local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
}
L=/root/repo/ssqueezepy_amd
export SSQ_TILE_ORDER=ordered
SSQ_HIP_LIB=$L/libssq_hip_pipe.so timeout 300 python -m pytest tests/test_gpu_00_configs.py tests/test_gpu_edge_cases.py -x -q -m gpu -k "config2 or every_instantiation or few_scales" 2>&1 | tail -2 | cut -c1-200
for rep in 1 2; do
run ordered A=1
run ordered-pipe SSQ_HIP_LIB=$L/libssq_hip_pipe.so
run e17-arith-only12 SSQ_HIP_LIB=$L/libssq_hip_e17.so SSQ_TILE_ORDER=atomic
run e17-arith-only16 SSQ_HIP_LIB=$L/libssq_hip_e17.so SSQ_TILE_ORDER=atomic SSQ_TILE_NW=16
run e17-arith-only8 SSQ_HIP_LIB=$L/libssq_hip_e17.so SSQ_TILE_ORDER=atomic SSQ_TILE_NW=8
run pipe17-12 SSQ_HIP_LIB=$L/libssq_hip_pipe17.so SSQ_TILE_ORDER=atomic
run pipe17-16 SSQ_HIP_LIB=$L/libssq_hip_pipe17.so SSQ_TILE_ORDER=atomic SSQ_TILE_NW=16
run e49-nogather SSQ_HIP_LIB=$L/libssq_hip_e49.so SSQ_TILE_ORDER=atomic
run e81-nomodulation SSQ_HIP_LIB=$L/libssq_hip_e81.so SSQ_TILE_ORDER=atomic
run e145-nobin SSQ_HIP_LIB=$L/libssq_hip_e145.so SSQ_TILE_ORDER=atomic
run e273-nostore SSQ_HIP_LIB=$L/libssq_hip_e273.so SSQ_TILE_ORDER=atomic
run e497-alloff SSQ_HIP_LIB=$L/libssq_hip_e497.so SSQ_TILE_ORDER=atomic
run e497-alloff16 SSQ_HIP_LIB=$L/libssq_hip_e497.so SSQ_TILE_ORDER=atomic SSQ_TILE_NW=16
done 2>&1 | tee $O/ab.txt
