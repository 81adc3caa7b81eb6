import os, sys
sys.path.insert(0, '/root/repo/gimm-vfi_amd')
import torch
from gimmvfi_hip import lib as L
from gimmvfi_hip.ops import ConvLayer, Runtime, View
rt = Runtime(L.get(), "bf16", "cuda:0")
torch.manual_seed(0)
N,H,W,Cin,Cout = 2,64,112,256,24
w = torch.randn(Cout,Cin,3,3)/48
lay = ConvLayer(rt, w, torch.zeros(Cout))
x = torch.randn(N,H,W,Cin, device='cuda').to(rt.tdtype)
outs={}
for algo in (1,2,130):
    out = rt.f32(N,H,W,Cout)
    rt.conv(lay, View(x,0,Cin), out, algo=algo)
    torch.cuda.synchronize()
    outs[algo]=out.clone()
for algo in (2,130):
    d=(outs[algo]-outs[1]).abs()
    print("algo",algo,"max",float(d.max()), "nbad", int((d>0.05).sum()))
    bad = (d>0.05).nonzero()
    if len(bad):
        print(" bad n", bad[:,0].unique().tolist()[:5], "y", bad[:,1].unique().tolist()[:20], "x", bad[:,2].unique().tolist()[:20], "c", bad[:,3].unique().tolist())
