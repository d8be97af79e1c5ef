#!/bin/bash
# cfg4 (bf16) kernel-trace summary: bash tools/profile_cfg4.sh r2  -> gpurun_out/<tag>_cfg4_kernel_stats.csv, <tag>_cfg4_bench_line.json
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --config cfg4 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -- $B --steps 3 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/kt4 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cfg4_kernel_stats.csv
$B --steps 10 --warmup 3 2>/dev/null | tail -1 > $OUT/${TAG}_cfg4_bench_line.json
echo done
