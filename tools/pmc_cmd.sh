#!/bin/bash
# rocprofv3 PMC passes (one counter group per pass, --kernel-trace only) of ANY command:
#   tools/pmc_cmd.sh <outdir> <command ...>      (relative paths in the command: from the repository's root)
set -u
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  ( cd "${GRAFT_REPO_ROOT:-/root/repo}" && rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pass$i" -o pmc --output-format csv -- "$@" ) > "$OUT/pass$i.log" 2>&1
  echo "pass $i ($grp): rc=$?"
done
