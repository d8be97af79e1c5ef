cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_wino_realistic.py -x -q -m gpu -s -k "offset or trained_cfg2_operands" 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r6_wino_realistic.txt
cat gpurun_out/r6_wino_realistic.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py -x -q -m gpu 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r6_bench_a.json 2> gpurun_out/r6_bench_a.err; tail -3 gpurun_out/r6_bench_a.err; cat gpurun_out/r6_bench_a.json
FDN_CONV_ALGO=winograd_bf16x3 timeout 300 python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r6_bench_a_bf16x3.json 2>/dev/null; cat gpurun_out/r6_bench_a_bf16x3.json
