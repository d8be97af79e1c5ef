#!/bin/bash
# PMC passes over tools/bench_bf16_one.py (N P dbg); usage: tools/pmc_bf16.sh N P DBG
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" \
           "TCP_TCC_READ_REQ_LATENCY_sum SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm_$i -- python $R/tools/bench_bf16_one.py $1 $2 $3 > /tmp/pm.log 2>&1
  python $R/tools/pmc_dump.py /tmp/pm_$i conv64_bf16
  rm -rf /tmp/pm_$i
done
