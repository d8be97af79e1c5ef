#!/bin/bash
# Same-box A/B of one source file: runs CMD on the tree as it is (NEW), swaps FILE for ALT (a copy of the other version shipped under
# gpurun_out/ or tools/), rebuilds, runs CMD again (OLD), restores and runs NEW once more.  Box-to-box differences (+-3 % on the bf16
# path) make any other comparison meaningless.   usage: bash tools/ab_source.sh <file> <alt copy> '<cmd>'
F=$1; ALT=$2; CMD=$3
cp $F /tmp/ab_new
echo "== NEW"; python 4dflownet_amd/build.py > /dev/null 2>&1; bash -c "$CMD"
cp $ALT $F; python 4dflownet_amd/build.py > /dev/null 2>&1
echo "== OLD"; bash -c "$CMD"
cp /tmp/ab_new $F; python 4dflownet_amd/build.py > /dev/null 2>&1
echo "== NEW again"; bash -c "$CMD"
