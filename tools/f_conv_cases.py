"""Unit shapes of the convolutions the FlowFormer path adds (patch / sub-sampling convolutions with stride == kernel,
6x6 stride-2 cost-map convolutions, token-matrix linears, grouped 'weights are activations' contractions, GELU) against
a torch statement.  Runs on the GPU (default) or in the host emulator (--sim: real kernel sources, lanes as fibers)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [os.path.join(ROOT, "gimm-vfi_amd"), os.path.join(ROOT, "tests", "hostsim"), os.path.join(ROOT, "tests")]
import torch
import torch.nn.functional as F

from gimmvfi_hip import lib as L
from gimmvfi_hip.ops import ConvLayer, Runtime, View


def cases():
    # name, N,H,W,Cin, Cout,k,stride,pad, act
    return [
        ("patch4 3->128", 2, 32, 48, 3, 128, 4, 4, 0, L.ACT_NONE),
        ("patch2 128->256", 2, 16, 24, 128, 256, 2, 2, 0, L.ACT_NONE),
        ("sr8 128->128", 2, 16, 24, 128, 128, 8, 8, 0, L.ACT_NONE),
        ("sr4 256->256", 2, 8, 12, 256, 256, 4, 4, 0, L.ACT_NONE),
        ("sr4 192->128", 4, 8, 12, 192, 128, 4, 4, 0, L.ACT_NONE),
        ("cost6 16->32", 8, 8, 12, 16, 32, 6, 2, 2, L.ACT_RELU),
        ("cost6 32->64", 8, 4, 6, 32, 64, 6, 2, 2, L.ACT_NONE),
        ("lin 128->384 rows 1536", 1, 1, 1536, 128, 384, 1, 1, 0, L.ACT_NONE),
        ("lin 128->512 gelu", 1, 1, 1536, 128, 512, 1, 1, 0, L.ACT_GELU),
        ("lin 512->128", 1, 1, 1536, 512, 128, 1, 1, 0, L.ACT_NONE),
        ("lin 192->256", 1, 1, 1536, 192, 256, 1, 1, 0, L.ACT_NONE),
        ("lin 128->128 rows 8", 1, 1, 8, 128, 128, 1, 1, 0, L.ACT_NONE),
        ("lin 81->64 gelu (view coff 64)", 1, 1, 384, 81, 64, 1, 1, 0, L.ACT_GELU),
        ("lin 64->64", 1, 1, 384, 64, 64, 1, 1, 0, L.ACT_NONE),
        ("lin 256->64", 1, 1, 384, 256, 64, 1, 1, 0, L.ACT_NONE),
    ]


def main():
    sim = "--sim" in sys.argv
    prec = "bf16" if "--bf16" in sys.argv else "fp32"
    if sim:
        from sim_runtime import SimRuntime

        rt = SimRuntime(prec, emulate_conv=True)
    else:
        rt = Runtime(L.get(), prec, "cuda:0")
    dev = rt.device
    g = torch.Generator().manual_seed(0)
    worst = 0.0
    only = [a.split("=", 1)[1] for a in sys.argv if a.startswith("--only=")]
    for name, N, H, W, Ci, Co, k, st, pad, act in cases():
        if only and not any(o in name for o in only):
            continue
        w = (torch.rand(Co, Ci, k, k, generator=g) - 0.5) * (2.0 / (Ci * k * k) ** 0.5)
        b = torch.rand(Co, generator=g) - 0.5
        lay = ConvLayer(rt, w, b, stride=st, pad=(pad, pad))
        coff = 64 if "coff" in name else 0
        ld = rt.cp64(coff + Ci) if coff else rt.cp(Ci)
        x = torch.zeros(N, H, W, ld)
        x[..., coff:coff + Ci] = torch.rand(N, H, W, Ci, generator=g) - 0.5
        xq = x.to(rt.tdtype)
        Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        out = torch.empty(N, Ho, Wo, rt.cp(Co), dtype=rt.tdtype, device=dev)
        rt.conv(lay, View(xq.to(dev), coff, Ci), out, act1=act)
        if not sim:
            torch.cuda.synchronize()
        wq = w.to(rt.tdtype).float()
        ref = F.conv2d(xq.float()[..., coff:coff + Ci].permute(0, 3, 1, 2), wq, b, stride=st, padding=pad)
        if act == L.ACT_RELU:
            ref = F.relu(ref)
        elif act == L.ACT_GELU:
            ref = F.gelu(ref)
        got = out.float().cpu()[..., :Co].permute(0, 3, 1, 2)
        err = float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-9)
        worst = max(worst, err)
        print(f"{name:34s} rel err {err:.3e}")
    # grouped contractions with activations as weights (GMA):  vT = Wv_rep x mf^T ;  out = attn x vT^T + res
    n, P8 = 2, 192
    wv = (torch.rand(128, 128, generator=g) - 0.5).to(rt.tdtype)
    mf = (torch.rand(n, 12, 16, 128, generator=g) - 0.5).to(rt.tdtype)
    att = torch.softmax(torch.rand(n, P8, P8, generator=g) * 4, -1).to(rt.tdtype)
    wv_rep = wv.view(1, 1, 128, 128).expand(n, 1, 128, 128).contiguous().to(dev)
    vT = torch.empty(n, 1, 128, P8, dtype=rt.tdtype, device=dev)
    mfd = mf.to(dev)
    rt.conv(None, wv_rep, vT, groups=n, w_group_stride=P8 * 128, w_raw=mfd, cout=P8)
    X = torch.zeros(n, 1, P8, 256, dtype=rt.tdtype, device=dev)
    X[..., :128] = mfd.view(n, 1, P8, 128)
    rt.conv(None, att.view(n, 1, P8, P8).to(dev), View(X, 128, 128), groups=n, w_group_stride=128 * P8, w_raw=vT,
            cout=128, res=View(X, 0, 128))
    if not sim:
        torch.cuda.synchronize()
    v_ref = torch.einsum("dc,bjc->bdj", wv.float(), mf.float().view(n, P8, 128))
    e1 = float((vT.float().cpu().view(n, 128, P8) - v_ref).abs().max()) / float(v_ref.abs().max())
    vq = vT.float().cpu().view(n, 128, P8)
    o_ref = torch.einsum("bij,bdj->bid", att.float(), vq) + mf.float().view(n, P8, 128)
    e2 = float((X.float().cpu()[:, 0, :, 128:] - o_ref).abs().max()) / float(o_ref.abs().max())
    print(f"{'grouped vT (x = weights, w = acts)':34s} rel err {e1:.3e}")
    print(f"{'grouped attn @ vT + res':34s} rel err {e2:.3e}")
    worst = max(worst, e1, e2)
    print("worst", worst)


if __name__ == "__main__":
    main()
