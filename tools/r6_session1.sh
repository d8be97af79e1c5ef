set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/wino_bf16_kloop tools/wino_bf16_kloop.hip && timeout 300 /tmp/wino_bf16_kloop > gpurun_out/r6_wino_bf16_kloop.txt 2>&1
cat gpurun_out/r6_wino_bf16_kloop.txt
timeout 900 python tools/bench_halftile.py > gpurun_out/r6_halftile.txt 2>&1
cat gpurun_out/r6_halftile.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wino2d or fused_fold" 2>&1 | tail -5
