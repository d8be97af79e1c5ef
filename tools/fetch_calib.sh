#!/bin/bash
# FETCH_SIZE / TCC_EA0_RDREQ of known access patterns (tools/fetch_calib.hip) -> gpurun_out/<tag>_fetch_calib.txt
TAG=${1:-r4}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o /tmp/fetch_calib || exit 1
: > $OUT/${TAG}_fetch_calib.txt
/tmp/fetch_calib >> $OUT/${TAG}_fetch_calib.txt
for set in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum TCP_TCC_READ_REQ_sum"; do
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/fc -- /tmp/fetch_calib > /dev/null 2>&1
  echo "# pass: $set" >> $OUT/${TAG}_fetch_calib.txt
  python $R/tools/pmc_dump.py /tmp/fc k_ >> $OUT/${TAG}_fetch_calib.txt
  rm -rf /tmp/fc
done
cat $OUT/${TAG}_fetch_calib.txt
