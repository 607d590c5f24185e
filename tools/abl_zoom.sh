set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for ABL in 0 1 2; do
  cd $GRAFT_REPO_ROOT/ssqueezepy_amd/csrc
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSSQ_ABL=$ABL -c ssq_cwt_blocks.hip -o _obj/ssq_cwt_blocks.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libssq_hip.so _obj/*.o -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
  cd /tmp
  rocprofv3 --kernel-trace --stats -d /tmp/pa_$ABL -o p -- python $GRAFT_REPO_ROOT/tools/cwt_only_probe.py > /tmp/c.log 2>&1
  echo "== ABL=$ABL"; python $GRAFT_REPO_ROOT/tools/prof_summary.py /tmp/pa_$ABL/p_results.db | grep blockzoom | cut -c1-50,112-160
done
