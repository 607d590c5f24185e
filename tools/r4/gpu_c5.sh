#!/bin/bash
# round 4: float64 block kernels at two wavefronts per SIMD (amdgpu_waves_per_eu(2,2): 256 registers + ~90 bytes of scratch)
# against one (276-280 registers): C5 timing, float64 tests
cd /root/repo; O=gpurun_out/r4c5; mkdir -p $O
for v in occ1 default occ1 default; do
  if [ "$v" != default ]; then export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; else unset SSQ_HIP_LIB; fi
  echo -n "lib=$v "; timeout 300 python tools/run_configs.py c5 2>&1 | grep config | cut -c1-120
done | tee $O/c5.txt
unset SSQ_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_transforms.py tests/test_gpu_00_configs.py -x -q -m gpu -k "float64 or config5" 2>&1 | tail -2 | cut -c1-200
