cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ov in 0 1; do
  FDN_OVERLAP_WGRAD=$ov rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_$ov -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-pmc --event-steps 0 > /tmp/gap_$ov.log 2>&1
  echo "== FDN_OVERLAP_WGRAD=$ov"; tail -1 /tmp/gap_$ov.log | cut -c1-120
  python $R/tools/gap_analysis.py /tmp/gap_$ov
done
