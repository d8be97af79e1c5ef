# A/B of the XCD-aware orders (conv64_wino bit 128 = off, wgrad64_wino bit 8 = off): times, then HBM read traffic per launch (FETCH_SIZE pass)
cd /root/repo
export TMPDIR=/tmp
python - <<'PY' 2>&1 | grep -v amdgpu
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
for P in (48, 24):
    x = torch.randn((8, P, P, P, 64), device="cuda"); dz = torch.randn_like(x)
    ws = torch.empty(ops.wgrad_workspace_bytes(8, P, P, P, 64, 64, 3) // 4 + 1, device="cuda"); dw = torch.empty((3, 3, 3, 64, 64), device="cuda")
    for bits in (24, 16, 8, 0, 24, 16, 8, 0):
        lib.fdn_debug_set_wgrad64_wino_dbg(bits)
        for _ in range(3): ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        print("wgrad P=%d bits %2d (8 = no xcd placement, 16 = w-fastest order) %7.4f ms (incl. reduce)" % (P, bits, e0.elapsed_time(e1) / 20))
lib.fdn_debug_set_wgrad64_wino_dbg(0)
PY
cat > /tmp/wg_one.py <<'PY'
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
fdn = importlib.import_module("4dflownet_amd"); ops = fdn.ops
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
lib.fdn_debug_set_wgrad64_wino_dbg(int(sys.argv[1]))
for P in (48, 24):
    x = torch.randn((8, P, P, P, 64), device="cuda"); dz = torch.randn_like(x)
    ws = torch.empty(ops.wgrad_workspace_bytes(8, P, P, P, 64, 64, 3) // 4 + 1, device="cuda"); dw = torch.empty((3, 3, 3, 64, 64), device="cuda")
    for _ in range(6): ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=dw, workspace=ws)
torch.cuda.synchronize()
PY
for b in 24 16 0; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/xcdw_$b -o f -- python /tmp/wg_one.py $b > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob('gpurun_out/xcdw_$b/**/f_counter_collection.csv',recursive=True)
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if 'wgrad64_wino' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
        agg[(r['Kernel_Name'][:40], r['Grid_Size'])].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print('bits=$b',k,len(v),'FETCH_SIZE/launch %.0f (x tensor = %s)'%(sum(v)/len(v), ''))
PY
done
