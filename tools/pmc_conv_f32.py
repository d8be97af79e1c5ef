"""fp32 conv64 forward only (for PMC passes): python tools/pmc_conv_f32.py [N] [P] [dbg] [variant]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
ops = fdn.ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dbg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
var = int(sys.argv[4]) if len(sys.argv) > 4 else 0
_tb = fdn._lib.test_build()                     # test build: the fdn_debug_* hooks are not in the product library
lib = _tb.__enter__()                          # (keep _tb alive: closing it restores the product library)
x = torch.randn((N, P, P, P, 64), device="cuda")
w = torch.randn((3, 3, 3, 64, 64), device="cuda") * 0.02
wf, wd = ops.pack_conv64_weights(w)
y = torch.empty_like(x)
lib.fdn_debug_set_conv64_mt(var)
lib.fdn_debug_set_conv64_dbg(dbg)
for _ in range(6):
    ops.conv3d_fwd(x, w, None, ops.ACT_RELU, wpack=wf, out=y)
torch.cuda.synchronize()
