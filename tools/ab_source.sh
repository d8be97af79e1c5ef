#!/bin/bash
# Same-box A/B of one source file: runs CMD on the tree as it is (NEW), swaps FILE for ALT (a copy of the other version shipped under
# gpurun_out/ or tools/), rebuilds, runs CMD again (OLD), restores and runs NEW once more.  Box-to-box differences (+-3 % on the bf16
# path) make any other comparison meaningless.   usage: bash tools/ab_source.sh <file> <alt copy> '<cmd>'
# The tracked file is restored on every exit path (trap), the backup is a private mktemp file, and a failed build aborts the run.
set -e
F="$1"; ALT="$2"; CMD="$3"
[ -f "$F" ] && [ -f "$ALT" ] || { echo "usage: $0 <file> <alt copy> '<cmd>'" >&2; exit 2; }
bak=$(mktemp /tmp/ab_source.XXXXXX)
cp "$F" "$bak"
trap 'cp "$bak" "$F"; rm -f "$bak"' EXIT
build() { python 4dflownet_amd/build.py > /dev/null || { echo "build failed ($1)" >&2; exit 1; }; }
echo "== NEW"; build new; bash -c "$CMD"
cp "$ALT" "$F"; build old
echo "== OLD"; bash -c "$CMD"
cp "$bak" "$F"; build new
echo "== NEW again"; bash -c "$CMD"
