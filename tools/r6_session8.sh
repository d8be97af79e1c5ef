cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 300 python tools/bench_small_grids.py > gpurun_out/r6_small_grids_b.txt 2>&1; cat gpurun_out/r6_small_grids_b.txt
