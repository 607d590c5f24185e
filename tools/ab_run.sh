# usage inside ONE gpurun call: bash tools/ab_run.sh base v1 v2 ...  (libssq_hip_<v>.so built by tools/ab_variant.sh)
for i in 1 2; do for v in "$@"; do
  if [ $v = base ]; then unset SSQ_HIP_LIB; else export SSQ_HIP_LIB=/root/repo/ssqueezepy_amd/libssq_hip_$v.so; fi
  echo -n "$v "; timeout 40 python bench.py --no-cpu --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['stages_us_per_transform']['reassignment_us'],1))"
done; done
