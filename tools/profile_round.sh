#!/bin/bash
# Round profile on the GPU box: kernel-trace stats, HBM traffic (two PMC passes), MFMA/LDS counters of the default bench.
# usage (from the repo root, through gpurun): bash tools/profile_round.sh r2   -> gpurun_out/<tag>_*
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $B --steps 5 --warmup 2 > $OUT/${TAG}_prof_bench.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $B --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $B --steps 2 --warmup 1 > /dev/null 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt
: > $OUT/${TAG}_pmc_sq.txt
i=0
for set in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pq_$i -- $B --steps 1 --warmup 1 > /dev/null 2>&1
  echo "# pass $i: $set" >> $OUT/${TAG}_pmc_sq.txt
  python $R/tools/pmc_dump.py /tmp/pq_$i | grep -i "wino\|conv64_mfma\|wgrad64\|head_\|fold_halo" >> $OUT/${TAG}_pmc_sq.txt
  rm -rf /tmp/pq_$i
done
$B --steps 10 --warmup 3 > $OUT/${TAG}_bench_line_short.json 2>/dev/null
echo done
