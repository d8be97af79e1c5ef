cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train_step.py tests/test_gpu_device_loader.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_one_stream'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline_wgrad']['frac'])"; done
FDN_OVERLAP_WGRAD=0 timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no overlap', d['value'], d['ms_per_step'])"
