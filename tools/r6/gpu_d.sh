#!/bin/bash
# round 6: the short bench per environment / library variant given as "label[:ENV=VAL[,ENV=VAL]][@lib]" words, one box
cd /root/repo; O=gpurun_out/${OUT:-r6d}; mkdir -p $O
for w in "$@"; do
  label=${w%%[:@]*}; rest=${w#$label}
  lib=""; envs=""
  case "$rest" in *@*) lib=${rest##*@}; rest=${rest%@*};; esac
  envs=${rest#:}
  ( if [ -n "$lib" ]; then export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$lib.so; fi
    IFS=','; for e in $envs; do [ -n "$e" ] && export "$e"; done; unset IFS
    echo -n "$label "; timeout 200 python bench.py --no-cpu --steps ${STEPS:-10} $BENCH_ARGS 2>$O/err_$label.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), {k: round(v,1) for k,v in d['stages_us_per_transform'].items()}, (d['config'].get('tile_kernel') or '')[-14:])" )
done 2>&1 | tee -a $O/ab.txt
