#!/bin/bash
# round 5: the mixed-radix fused STFT kernel: its tests on the device, the reference's published shape (n_fft = 598, hop 1)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_transforms.py -q -m gpu -x -k "stft" > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
timeout 300 python tools/stft_hop1_probe.py 1024 598 600 1000 2>&1 | tee $O/hop1.txt
SSQ_DEBUG_STFT_GENERIC=1 timeout 300 python tools/stft_hop1_probe.py 598 2>&1 | sed 's/^/rocfft route: /' | tee -a $O/hop1.txt
