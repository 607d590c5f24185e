#!/bin/bash
# round 6: the fused STFT kernel summing Tx itself (LDS planes in the FFT buffer's place) beside the two-pass route, one box:
# config 3 at 512 / 64 / 1 signals per step, and hop 1 at n_fft = 1024
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/${RUNTAG:-r6w}; mkdir -p $O
for ft in 0 1 default; do
  if [ $ft = default ]; then unset SSQ_DEBUG_STFT_FUSED_TX; else export SSQ_DEBUG_STFT_FUSED_TX=$ft; fi
  for B in 512 64 8; do
    echo -n "FUSED_TX=$ft B=$B: "
    timeout 300 python tools/probes/c3_batched_probe.py $B 2>/dev/null
  done
  echo -n "FUSED_TX=$ft hop1: "
  timeout 300 python tools/stft_hop1_probe.py 1024 2>/dev/null | cut -c1-200
done | tee $O/stft_ab.txt
unset SSQ_DEBUG_STFT_FUSED_TX
SSQ_DEBUG_STFT_FUSED_TX=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c3 -o c3 -- python tools/probes/c3_batched_probe.py 512 > $O/prof_c3.log 2>&1
DB=$(find $O/prof_c3 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_c3_fused.txt | head -12 | cut -c1-170
rm -rf $O/prof_c3
timeout 600 python -m pytest tests -q -m gpu -x -k "stft" 2>&1 | tail -3
# ablations (wrong results, time only) of the kernel that sums Tx itself
export SSQ_DEBUG_STFT_FUSED_TX=1
for v in NOLOAD NOWIN NOFFT NOSX NOBIN NOTX NOOUT; do
  L=/root/repo/ssqueezepy_amd/libssq_hip_abl_$v.so; [ -f $L ] || continue
  echo -n "abl $v B=512: "; SSQ_HIP_LIB=$L timeout 300 python tools/probes/c3_batched_probe.py 512 2>/dev/null
  echo -n "abl $v hop1: "; SSQ_HIP_LIB=$L timeout 300 python tools/stft_hop1_probe.py 1024 2>/dev/null | cut -c1-140
done | tee $O/stft_ablations.txt
unset SSQ_DEBUG_STFT_FUSED_TX
# config 1 (one cwt call, N = 10 000): what is launched
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c1 -o c1 -- python tools/run_configs.py c1 > $O/prof_c1.log 2>&1
DB=$(find $O/prof_c1 -name "*_results.db" | head -1); [ -n "$DB" ] && python tools/prof_summary.py $DB $O/kernel_stats_c1.txt | head -16 | cut -c1-170
rm -rf $O/prof_c1
