"""One shape of the bf16 forward conv, a few launches (for rocprofv3 --pmc passes)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
bops = importlib.import_module("4dflownet_amd.ops_bf16")
N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 128)
dbg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
fdn = importlib.import_module("4dflownet_amd")
_tb = fdn._lib.test_build()
_tb.__enter__().fdn_debug_set_conv64_bf16_dbg(dbg)
w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
wf, wd = bops.pack_conv64_weights(w)
x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
out = torch.empty_like(x)
for _ in range(5):
    bops.conv64_fwd(x, wf, None, 1, 0.2, None, out)
torch.cuda.synchronize()
