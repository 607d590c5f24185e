#!/bin/bash
# round 6: the mixed-radix STFT kernel at 2 / 4 workgroups per CU (frames per workgroup 8 / 4 at n_fft = 598): hop 1 and batched
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6y15}; mkdir -p $O
for rep in 1 2; do for kb in 78 39; do
  echo -n "LDS=$kb KB hop1 598: "; SSQ_DEBUG_STFT_GEN_LDS=$kb timeout 300 python tools/stft_hop1_probe.py 598 2>/dev/null | cut -c1-130
  echo -n "LDS=$kb KB 598/149 B=512: "; SSQ_DEBUG_STFT_GEN_LDS=$kb timeout 300 python tools/probes/c3_batched_probe.py 512 598 149 2>/dev/null
  echo -n "LDS=$kb KB 300/75 B=512: "; SSQ_DEBUG_STFT_GEN_LDS=$kb timeout 300 python tools/probes/c3_batched_probe.py 512 300 75 2>/dev/null
  echo -n "LDS=$kb KB 1000/250 B=256: "; SSQ_DEBUG_STFT_GEN_LDS=$kb timeout 300 python tools/probes/c3_batched_probe.py 256 1000 250 2>/dev/null
done; done | tee $O/stft_generic_lds2.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "stft" 2>&1 | tail -2
