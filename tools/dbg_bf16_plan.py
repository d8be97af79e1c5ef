"""Print the tile plans of the bf16 fused dgrad's regions (test build, FDN_DEBUG_PLAN) and time the FAST / general launches."""
import importlib, os, sys
os.environ["FDN_DEBUG_PLAN"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd"); bops = importlib.import_module("4dflownet_amd.ops_bf16")
_tb = fdn._lib.test_build(); lib = _tb.__enter__()
N, P = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 128)
w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
wf, wd = bops.pack_conv64_weights(w)
x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
out = torch.empty_like(x); pad = torch.empty(N, P + 2, P + 2, P + 2, 64, device="cuda")
bops.conv64_dgrad_fused(x, wd, pad, out, skip=None, y_prev=None, act=0)
torch.cuda.synchronize()
