import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fdn = importlib.import_module("4dflownet_amd")
bops = importlib.import_module("4dflownet_amd.ops_bf16")
_tb = fdn._lib.test_build()                     # test build: the fdn_debug_* hooks are not in the product library
lib = _tb.__enter__()                          # (keep _tb alive: closing it restores the product library)
torch.manual_seed(0)
w = torch.randn(3, 3, 3, 64, 64, device="cuda") * 0.05
wf, wd = bops.pack_conv64_weights(w)
wt = w.to(torch.bfloat16).float().permute(4, 3, 0, 1, 2).contiguous()
for P in (16, 24, 32, 40):
    for N in (1, 2, 3):
        x = torch.randn(N, P, P, P, 64, device="cuda").to(torch.bfloat16)
        xp = torch.nn.functional.pad(x.float().permute(0, 4, 1, 2, 3), (1, 1, 1, 1, 1, 1), mode="replicate")
        ref = torch.nn.functional.conv3d(xp, wt).permute(0, 2, 3, 4, 1)
        for mt in (0, 4, 8):
            lib.fdn_debug_set_conv64_bf16_mt(mt)
            for dbg in (0, 2):
                lib.fdn_debug_set_conv64_bf16_dbg(dbg)
                y = bops.conv64_fwd(x, wf, None, 0).float()
                err = (y - ref).abs().max().item()
                nbad = ((y - ref).abs() > 0.05).sum().item()
                print("P=%d N=%d mt=%d dbg=%d maxerr %.3f nbad %d" % (P, N, mt, dbg, err, nbad), flush=True)
lib.fdn_debug_set_conv64_bf16_mt(0); lib.fdn_debug_set_conv64_bf16_dbg(0)
