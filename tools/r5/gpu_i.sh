#!/bin/bash
# round 5: first tiles of tile2_kernel's workgroups permuted per XCD (adjacent tiles meet in one L2): time and HBM reads
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r5p; mkdir -p $O
for x in 1 0 1 0; do
  echo -n "SSQ_DEBUG_TILE2_XCD=$x : "; SSQ_DEBUG_TILE2_XCD=$x timeout 300 python bench.py --no-cpu --steps 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()})"
done 2>&1 | tee $O/ab.txt
cd /tmp
for x in 1 0; do
  for c in FETCH_SIZE WRITE_SIZE; do
    SSQ_DEBUG_TILE2_XCD=$x rocprofv3 --kernel-trace --pmc $c -d $GRAFT_REPO_ROOT/$O/pmc_${x}_$c -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 3 > /dev/null 2>&1
  done
  echo "SSQ_DEBUG_TILE2_XCD=$x"; python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $GRAFT_REPO_ROOT/$O/pmc_${x}_FETCH_SIZE $(( 9 * 16 )) 2>/dev/null | grep -E "tile2|total" | sed "s/^/  read : /"
  python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $GRAFT_REPO_ROOT/$O/pmc_${x}_WRITE_SIZE $(( 9 * 16 )) 2>/dev/null | grep -E "tile2|total" | sed "s/^/  write: /"
done 2>&1 | tee -a $GRAFT_REPO_ROOT/$O/ab.txt
rm -rf $GRAFT_REPO_ROOT/$O/pmc_*
