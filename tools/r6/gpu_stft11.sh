#!/bin/bash
# round 6: short mixed-radix transforms -- is it the workgroups per CU or the frames per workgroup? n_fft = 300 and 150
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y16}; mkdir -p $O
for rep in 1 2; do for kb in 39 20 10; do
  echo -n "LDS=$kb KB 300/75 B=512: "; SSQ_DEBUG_STFT_GEN_LDS=$kb timeout 300 python tools/probes/c3_batched_probe.py 512 300 75 2>/dev/null
  echo -n "LDS=$kb KB 150/37 B=512: "; SSQ_DEBUG_STFT_GEN_LDS=$kb timeout 300 python tools/probes/c3_batched_probe.py 512 150 37 2>/dev/null
done; done | tee $O/stft_generic_lds3.txt
