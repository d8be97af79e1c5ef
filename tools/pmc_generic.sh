#!/bin/bash
# usage: tools/pmc_generic.sh <kernel-name-substring> <python script> [args...]   -- PMC passes, per-kernel means
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; K=$1; shift
i=0
for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pg_$i -- python $R/$@ > /tmp/pg.log 2>&1
  python $R/tools/pmc_dump.py /tmp/pg_$i $K
  rm -rf /tmp/pg_$i
done
