#!/bin/bash
cd /root/repo; O=gpurun_out/r4q; mkdir -p $O
timeout 200 python tools/r4/atomic_vs_ordered.py 160000 300 4 2>/dev/null | tail -1 | cut -c1-200
run() { local label=$1; shift
  echo -n "$label "; env "$@" timeout 120 python bench.py --no-cpu --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"; }
for rep in 1 2; do
run f64 A=1
run ordered SSQ_TILE_ORDER=ordered
done 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES -d /root/repo/$O/pmc/pass1 -o pmc --output-format csv -- python /root/repo/bench.py --no-cpu --steps 2 --warmup 1 > /root/repo/$O/pmc1.log 2>&1
cd /root/repo; python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1; grep -A9 "tile2_kernel" $O/pmc_summary.txt | head -10
rm -rf $O/pmc/*/
