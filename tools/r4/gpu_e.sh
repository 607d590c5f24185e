#!/bin/bash
# round 4: where a wavefront of the tile kernel spends a step (shader-clock stamps), and the CU-side counters
cd /root/repo; O=gpurun_out/r4e; mkdir -p $O
L=/root/repo/ssqueezepy_amd
SSQ_HIP_LIB=$L/libssq_hip_trace.so SSQ_TILE_TRACE=$O/trace_ordered.bin timeout 100 python bench.py --no-cpu --steps 3 > $O/b1.json 2>$O/b1.err
python tools/tile_trace.py $O/trace_ordered.bin > $O/trace_ordered.txt; head -50 $O/trace_ordered.txt
SSQ_HIP_LIB=$L/libssq_hip_trace17.so SSQ_TILE_ORDER=atomic SSQ_TILE_TRACE=$O/trace_e17.bin timeout 100 python bench.py --no-cpu --steps 3 > $O/b2.json 2>$O/b2.err
python tools/tile_trace.py $O/trace_e17.bin > $O/trace_e17.txt; head -40 $O/trace_e17.txt
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d /root/repo/$O/pmc/pass$i -o pmc --output-format csv -- python /root/repo/bench.py --no-cpu --steps 2 --warmup 1 > /root/repo/$O/pmc_pass$i.log 2>&1
  echo "pass $i: rc=$?"
done
cd /root/repo; python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt 2>&1; grep -A60 "tile_kernel" $O/pmc_summary.txt | head -70
rm -rf $O/pmc/*/ $O/*.bin
