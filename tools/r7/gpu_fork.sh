#!/bin/bash
# round 6 (second session): the first launch group's decimated samples forked to the side stream right behind the forward transform (new)
# against behind the block spectra (SSQ_DEBUG_EARLY_FORK=0): one short signal, the bench at 16 / 1 signals per step; one box
cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/${OUT:-r7t}; mkdir -p $O
for rep in 1 2; do
for ef in 1 0; do
  export SSQ_DEBUG_EARLY_FORK=$ef
  echo "early fork $ef"
  python tools/r7/ssq_small_probe.py 10000 2>>$O/err.txt | grep ssq_cwt
  python tools/r7/ssq_small_probe.py 40000 2>>$O/err.txt | grep ssq_cwt
  timeout 200 python bench.py --no-cpu --steps 20 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=16', round(d['value']), round(d['ms_per_step'],3))"
  timeout 200 python bench.py --no-cpu --steps 200 --warmup 20 --batch 1 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=1', round(d['value']), round(d['ms_per_step']*1e3,1), 'us')"
  timeout 200 python bench.py --no-cpu --steps 5 --batch 64 2>>$O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B=64', round(d['value']), round(d['ms_per_step'],3))"
done; done 2>&1 | tee $O/fork.txt
