#!/bin/bash
# round 5: the block kernels' undecided points through a queue + fix-up kernel (default) against evaluated in place
# (SSQ_BLOCK_QUEUE=0), same library, same box; then the full-size parity checks. Measured and not kept
# (profiles/r5_ab_history.txt, block "r5zz"): the queue, its fix-up kernel and the switch are not in the tree any more.
cd /root/repo; O=gpurun_out/${OUT:-r5q}; mkdir -p $O
run() { local label=$1; shift
  echo -n "$label "; timeout 200 python bench.py --no-cpu --steps ${STEPS:-8} "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for i in 1 2; do
  SSQ_BLOCK_QUEUE=0 run "in-place"
  run "queue"
  SSQ_BLOCK_QUEUE=0 run "in-place B=1" --batch 1
  run "queue B=1" --batch 1
done 2>&1 | tee -a $O/ab.txt
timeout 1200 python -m pytest tests/test_gpu_00_configs.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/ab.txt
