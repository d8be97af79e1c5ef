#!/bin/bash
# Round profile on the GPU box: kernel-trace stats, HBM traffic (two PMC passes), MFMA/LDS counters of the default bench, the
# same for cfg4 (bf16), and the full bench line.  Everything lands in gpurun_out/<tag>_*; tools/collect_profiles.py then copies
# it into profiles/ with the commit and the library's source stamp recorded inside every file.
# usage (from the repo root, through gpurun): bash tools/profile_round.sh r3 [nocfg4]
TAG=${1:-r3}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python - <<PY > $OUT/${TAG}_source_stamp.txt
import importlib, sys
sys.path.insert(0, "$R")
b = importlib.import_module("4dflownet_amd.build")
b.build_library()
print(b.source_stamp())
PY
B="python $R/bench.py --no-cpu-baseline --no-secondary"
# the traced / counted passes run every launch on ONE stream (FDN_OVERLAP_WGRAD=0: the product step overlaps the weight gradients with
# the dgrad chain on a second stream, and a kernel that shares the chip has no duration of its own); the bench lines at the end do not
export FDN_OVERLAP_WGRAD=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $B --steps 5 --warmup 2 > $OUT/${TAG}_prof_bench.log 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $B --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $B --steps 2 --warmup 1 > /dev/null 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt
cp $OUT/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json    # (box-local copy) the bench lines below quote THIS run's counters
rm -rf /tmp/pf /tmp/pw /tmp/kt
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
if [ "$2" != "nocfg4" ]; then
  B4="python $R/bench.py --config cfg4 --no-cpu-baseline --no-secondary"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -- $B4 --steps 3 --warmup 1 > /dev/null 2>&1
  cp $(find /tmp/kt4 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cfg4_kernel_stats.csv
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf4 -- $B4 --steps 1 --warmup 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw4 -- $B4 --steps 1 --warmup 1 > /dev/null 2>&1
  python $R/tools/pmc_traffic.py /tmp/pf4 /tmp/pw4 $OUT/${TAG}_cfg4_pmc_traffic.json > $OUT/${TAG}_cfg4_pmc_traffic.txt
  cp $OUT/${TAG}_cfg4_pmc_traffic.json $R/profiles/${TAG}_cfg4_pmc_traffic.json
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pq4 -- $B4 --steps 1 --warmup 1 > /dev/null 2>&1
  echo "# cfg4: GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" > $OUT/${TAG}_cfg4_pmc_sq.txt
  python $R/tools/pmc_dump.py /tmp/pq4 | grep -i "bf16\|head_" >> $OUT/${TAG}_cfg4_pmc_sq.txt
  rm -rf /tmp/pf4 /tmp/pw4 /tmp/kt4 /tmp/pq4
  $B4 --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_cfg4_bench_line.json
fi
unset FDN_OVERLAP_WGRAD
$B --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line_nosecondary.json
python $R/bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_line.json
echo done
