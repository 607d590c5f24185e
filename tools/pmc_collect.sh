#!/bin/bash
# Collect rocprofv3 PMC counters for the bench, one counter group per pass
# (gfx950: FETCH_SIZE and WRITE_SIZE cannot share a pass; never combined with tracing
# beyond --kernel-trace). Usage: tools/pmc_collect.sh <outdir> [bench args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pass$i" -o pmc --output-format csv -- \
      python "$GRAFT_REPO_ROOT/bench.py" --no-cpu "$@" > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($grp): rc=$?"
done
