#!/bin/bash
# round 5: the P = 4096 block spectra in one launch of our own against gather + rocFFT (SSQ_DEBUG_BLOCK_SPECTRA=rocfft)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_transforms.py tests/test_gpu_00_configs.py -q -m gpu -x -k "block or nyquist or config2_ssq or cwt" > $O/tests.txt 2>&1; tail -3 $O/tests.txt | cut -c1-200
for m in own rocfft own rocfft; do
  echo -n "spectra=$m B=16: "; SSQ_DEBUG_BLOCK_SPECTRA=$m timeout 200 python bench.py --no-cpu --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
  echo -n "spectra=$m B=1 : "; SSQ_DEBUG_BLOCK_SPECTRA=$m timeout 200 python bench.py --no-cpu --steps 200 --warmup 20 --batch 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step']*1e3,1), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
done 2>&1 | tee $O/ab.txt
SSQ_DEBUG_BLOCK_SPECTRA=own timeout 100 python tools/run_configs.py c1 2>/dev/null | cut -c1-120 | tee -a $O/ab.txt
SSQ_DEBUG_BLOCK_SPECTRA=rocfft timeout 100 python tools/run_configs.py c1 2>/dev/null | cut -c1-120 | tee -a $O/ab.txt
