#!/bin/bash
# Same-box A/B of SEVERAL source files (tools/ab_source.sh handles one): runs CMD on the tree as it is (NEW), swaps every FILE for its
# ALT copy (the other version, shipped under gpurun_out/), rebuilds, runs CMD again (OLD), restores and runs NEW once more.
# usage: bash tools/ab_files.sh '<cmd>' <file1> <alt1> [<file2> <alt2> ...]
set -e
CMD="$1"; shift
[ $# -ge 2 ] && [ $(( $# % 2 )) -eq 0 ] || { echo "usage: $0 '<cmd>' <file> <alt> [...]" >&2; exit 2; }
bak=$(mktemp -d /tmp/ab_files.XXXXXX)
files=(); alts=()
while [ $# -gt 0 ]; do files+=("$1"); alts+=("$2"); [ -f "$1" ] && [ -f "$2" ] || { echo "missing $1 or $2" >&2; exit 2; }; shift 2; done
for i in "${!files[@]}"; do cp "${files[$i]}" "$bak/$i"; done
restore() { for i in "${!files[@]}"; do cp "$bak/$i" "${files[$i]}"; done; }
trap 'restore; rm -rf "$bak"' EXIT
build() { python 4dflownet_amd/build.py > /dev/null || { echo "build failed ($1)" >&2; exit 1; }; }
echo "== NEW"; build new; bash -c "$CMD"
for i in "${!files[@]}"; do cp "${alts[$i]}" "${files[$i]}"; done; build old
echo "== OLD"; bash -c "$CMD"
restore; build new
echo "== NEW again"; bash -c "$CMD"
