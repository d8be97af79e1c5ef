cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "thin or pack_batch" 2>&1 | tail -4
timeout 300 python tools/bench_heads.py 2>&1 | tail -12
