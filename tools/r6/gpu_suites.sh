#!/bin/bash
# round 6: the GPU suite in its three modes only (after tests were added behind the evidence run; same library stamp)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6z}; mkdir -p $O
python - > $O/build_sha.txt <<'PY'
import ctypes
l = ctypes.CDLL('/root/repo/ssqueezepy_amd/libssq_hip.so'); l.ssq_build_sha.restype = ctypes.c_char_p
print(l.ssq_build_sha().decode())
PY
echo "library: $(cat $O/build_sha.txt)"
timeout 1500 python -m pytest tests -q -m gpu > $O/gpu_suite.txt 2>&1; tail -1 $O/gpu_suite.txt
SSQ_TILE_ORDER=ordered timeout 1200 python -m pytest tests -q -m gpu > $O/gpu_suite_ordered_mode.txt 2>&1; tail -1 $O/gpu_suite_ordered_mode.txt
SSQ_CWT_TILES=0 timeout 1200 python -m pytest tests -q -m gpu > $O/gpu_suite_no_tiles.txt 2>&1; tail -1 $O/gpu_suite_no_tiles.txt
