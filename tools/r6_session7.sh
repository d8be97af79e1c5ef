cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
timeout 300 python tools/bench_small_grids.py > gpurun_out/r6_small_grids_b.txt 2>&1; cat gpurun_out/r6_small_grids_b.txt
timeout 600 python bench.py > gpurun_out/r6_bench_b.json 2> gpurun_out/r6_bench_b.err; tail -3 gpurun_out/r6_bench_b.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6_bench_b.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","ms_per_step_one_stream","products")})
print("roofline", {k:d["roofline"][k] for k in ("frac","avg_launch_ms","traffic","traffic_source")})
print("wgrad", {k:d["roofline_wgrad"][k] for k in ("frac","avg_launch_ms")})
for k,v in d.get("secondary",{}).items(): print(k, {kk:vv for kk,vv in v.items() if kk in ("value","ms_per_step","error","loader_alone_patches_per_s")} if isinstance(v,dict) else v)
print("cpu", d.get("cpu_baseline",{}).get("value"))
PY
