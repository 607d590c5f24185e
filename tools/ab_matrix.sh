#!/bin/bash
# One gpurun call, many A/B points: runs the short bench (no CPU baseline) under each of the
# library's tuning switches and prints transforms/s per setting, then the other configurations.
#   gpurun --timeout 600 -- 'bash tools/ab_matrix.sh'
# Box-to-box variance is +-1.5 %: compare settings within one call only.
cd "$(dirname "$0")/.."
run() {  # label, env assignments...
    local label=$1; shift
    local v
    v=$(env "$@" python bench.py --steps 10 --warmup 3 --no-cpu 2>/dev/null | tail -1 |
        python -c 'import sys, json; d = json.loads(sys.stdin.read()); print("%.1f  stages=%s" % (d["value"], {k: round(v, 1) for k, v in d.get("stages_us_per_transform", {}).items()}))')
    printf '%-28s %s\n' "$label" "$v"
}
run default            SSQ_NONE=1
run default-again      SSQ_NONE=1
run group=4            SSQ_DEBUG_CWT_GROUP=4
run group=2            SSQ_DEBUG_CWT_GROUP=2
run acc=16lanes        SSQ_DEBUG_ACC_VARIANT=1
run acc=8lanes-tc16    SSQ_DEBUG_ACC_VARIANT=3
run generic-cwt        SSQ_DEBUG_CWT_ALGO=generic
for v in 1 0; do SSQ_DEBUG_STFT_XCD=$v python tools/run_configs.py c3 2>/dev/null | tail -1 | cut -c1-140; done
python tools/stft_hop1_probe.py 2>/dev/null | tail -1 | cut -c1-120
RIDGE_N=40000 python tools/run_configs.py ridges 2>/dev/null | tail -1
SSQ_DEBUG_RIDGE_GENERIC=1 RIDGE_N=40000 python tools/run_configs.py ridges 2>/dev/null | tail -1
