cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino2d or fused_fold or one_launch or pack_batch" 2>&1 | tail -15
timeout 300 python tools/bench_wino2d.py --ablate 2>&1 | grep -v "2-D tile\|wgrad\|1-D\|direct\|F(2,3)" > gpurun_out/r6_bench_wino2d_c.txt 2>&1
cat gpurun_out/r6_bench_wino2d_c.txt
