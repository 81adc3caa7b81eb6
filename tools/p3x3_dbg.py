"""debug: where does a launch form of conv_p3x3 (algo bits 13, 14) differ from form 1 (the round-2 kernel), run to run"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
import torch
from gimmvfi_hip import lib as L
from gimmvfi_hip.ops import ConvLayer, Runtime, View
rt = Runtime(L.get(), "bf16", "cuda:0")
torch.manual_seed(0)
VAR = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, H, W, Cin, Cout = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), 256, 256) if len(sys.argv) > 4 else (2, 64, 64, 256, 256)
lay = ConvLayer(rt, torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5, torch.randn(Cout), slope=torch.rand(Cout) * 0.3 + 0.1)
x = torch.randn(N, H, W, Cin, device="cuda").to(rt.tdtype)
res = torch.randn(N, H, W, Cout, device="cuda").to(rt.tdtype)
for with_res in (False, True):
    kw = dict(res=res, act2=L.ACT_PRELU, slope2=lay.slope) if with_res else {}
    outs = []
    for v in (1, VAR, VAR, VAR):
        out = rt.act(N, H, W, Cout)
        out.fill_(3.0)
        rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_PRELU, algo=4 + (v << 13), **kw)
        torch.cuda.synchronize()
        outs.append(out.float().cpu())
    for k in (1, 2, 3):
        d = (outs[k] != outs[0])
        print("res" if with_res else "nores", "run", k, "mismatches", int(d.sum()), "of", d.numel())
        if d.any():
            idx = d.nonzero()
            print("  first", idx[:8].tolist())
            n, y, xx, c = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
            print("  y%16 hist", torch.bincount(y % 16, minlength=16).tolist())
            print("  x%16 hist", torch.bincount(xx % 16, minlength=16).tolist())
            print("  c//8 hist", torch.bincount(c // 8, minlength=32).tolist())
            print("  values", outs[k][d][:6].tolist(), "ref", outs[0][d][:6].tolist())
