#!/bin/bash
# The round's parity record: the whole `-m gpu` suite and `__graft_entry__.smoke()` on the GPU box, one log that names the sources
# it ran on (library source stamp = csrc + include + flags; suite stamp = package python + tests + entry points, both from
# 4dflownet_amd/build.py).  tools/collect_profiles.py copies it to profiles/<tag>_gputest.txt and refuses it when either stamp
# differs from the tree's.   usage (repo root, through gpurun): bash tools/gputest_round.sh r4 [extra pytest args]
TAG=${1:-r4}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_gputest.txt
cd $R
python - <<PY > $LOG
import importlib, sys
sys.path.insert(0, "$R")
b = importlib.import_module("4dflownet_amd.build")
b.build_library(); b.build_library(test_hooks=True)
print("lib_source_stamp", b.source_stamp())
print("lib_source_stamp_test_build", b.source_stamp(True))
print("suite_stamp", b.suite_stamp())
PY
echo "== rocminfo: $(/opt/rocm/bin/rocminfo 2>/dev/null | grep -m1 -o 'gfx9[0-9a-f]*') ; $(date -u +%Y-%m-%dT%H:%M:%SZ)" >> $LOG
echo "== python -m pytest tests -m gpu -q -rA -p no:cacheprovider $@" >> $LOG
python -m pytest tests -m gpu -q -rA -p no:cacheprovider "$@" >> $LOG 2>&1
echo "== pytest exit code $?" >> $LOG
echo "== python -c 'import __graft_entry__ as g; g.smoke()'" >> $LOG
python -c 'import __graft_entry__ as g; g.smoke()' >> $LOG 2>&1
echo "== smoke exit code $?" >> $LOG
echo "== loaded libraries are in-tree: $(ls -la 4dflownet_amd/*.so | awk '{print $NF, $5}' | tr '\n' ' ')" >> $LOG
tail -5 $LOG
