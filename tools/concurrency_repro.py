"""Minimal reproducer of the round-6 finding: a plain gather kernel (gvfi_splat_weights: 3x3 neighbourhood + one bilinear
warp of a float flow field, no LDS, no atomics) produces DIFFERENT results in a few 16-pixel runs when another stream runs heavy
kernels beside it.  Stream A: the metric kernel NA times on fixed flows, every result compared with the solo result.  Stream B:
one of several partner kernels in a loop.  usage: python tools/concurrency_repro.py"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, Runtime, View  # noqa: E402

DEV = "cuda:0"
rt = Runtime(L.get(), "bf16", DEV)
lib = rt.lib
torch.manual_seed(0)
B, H, W = 1, 544, 1024
base = torch.randn(B, 2, H // 8, W // 8, device=DEV) * 3
f01 = torch.nn.functional.interpolate(base, size=(H, W), mode="bilinear").permute(0, 2, 3, 1).contiguous()
f10 = -torch.nn.functional.interpolate(base.flip(1), size=(H, W), mode="bilinear").permute(0, 2, 3, 1).contiguous()
g9 = torch.tensor([1, 2, 1, 2, 4, 2, 1, 2, 1], dtype=torch.float32, device=DEV) / 16


def metric(z0, z1):
    rt._chk(lib.splat_weights(f01.data_ptr(), f10.data_ptr(), g9.data_ptr(), 1.0, 1.0, z0.data_ptr(), z1.data_ptr(), B, H, W, rt.stream()),
            "splat_weights")


ref0, ref1 = torch.empty(B, H, W, device=DEV), torch.empty(B, H, W, device=DEV)
metric(ref0, ref1)
torch.cuda.synchronize()
NA = 200
zs = [(torch.empty_like(ref0), torch.empty_like(ref1)) for _ in range(NA)]

# partner kernels
P8 = 68 * 128
fa = torch.randn(2, 68, 128, 256, device=DEV).to(rt.tdtype)
vol = rt.f32(2 * P8, P8)
lay = ConvLayer(rt, torch.randn(256, 256, 3, 3) / 48, torch.randn(256))
xh = torch.randn(2, 544, 1024, 256, device=DEV).to(rt.tdtype)
yh = rt.act(2, 544, 1024, 256)
big = torch.empty(256 << 20, dtype=torch.float32, device=DEV)
big2 = torch.empty_like(big)


def p_volume():
    rt.conv(None, fa[0:1], View(vol.view(2, 68, 128, P8)[0:1]), groups=1, w_group_stride=P8 * 256, w_raw=fa[1:2], cout=P8, out_scale=1 / 16.0)


def p_hot():
    rt.conv(lay, View(xh, 0, 256), yh, act1=L.ACT_RELU)


def p_copy():
    big2.copy_(big)


def p_fill():
    big.fill_(1.0)


lay1 = ConvLayer(rt, torch.randn(256, 256, 1, 1) / 16, torch.randn(256))
lay3 = ConvLayer(rt, torch.randn(128, 128, 3, 3) / 34, torch.randn(128))
x128 = torch.randn(2, 544, 1024, 128, device=DEV).to(rt.tdtype)
y128 = rt.act(2, 544, 1024, 128)
yf32 = rt.f32(2, 544, 1024, 256)
lay64 = ConvLayer(rt, torch.randn(64, 64, 3, 3) / 24, torch.randn(64))
x64 = torch.randn(2, 544, 1024, 64, device=DEV).to(rt.tdtype)
y64 = rt.act(2, 544, 1024, 64)


def mk(layer, x, c, out, **kw):
    return lambda: rt.conv(layer, View(x, 0, c), out, **kw)


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
partners = [("none", None), ("volume GEMM (LDS-DMA 256x256 tile, f32 out)", p_volume),
            ("hot 3x3 conv, stream form", p_hot), ("hot 3x3 conv, tile per workgroup", mk(lay, xh, 256, yh, algo=4 + (2 << 13))),
            ("hot 3x3 on the LDS-DMA 256x256 tile", mk(lay, xh, 256, yh, algo=2, tile=256)),
            ("1x1 256->256 LDS-DMA 256 tile, bf16 out", mk(lay1, xh, 256, yh, algo=2, tile=256)),
            ("1x1 256->256 LDS-DMA 256 tile, f32 out", mk(lay1, xh, 256, yf32, algo=2, tile=256)),
            ("1x1 256->256 LDS-DMA 128 tile", mk(lay1, xh, 256, yh, algo=2, tile=128)),
            ("3x3 128->128 LDS-DMA 128 tile", mk(lay3, x128, 128, y128, algo=2)),
            ("3x3 64->64 mid-channel halo kernel", mk(lay64, x64, 64, y64)),
            ("1 GB copy", p_copy), ("none", None)]
for name, partner in partners:
    torch.cuda.synchronize()
    if partner is not None:
        with torch.cuda.stream(sb):
            for _ in range(40):
                partner()
    with torch.cuda.stream(sa):
        for z0, z1 in zs:
            metric(z0, z1)
    torch.cuda.synchronize()
    bad = [int((z0 != ref0).sum()) + int((z1 != ref1).sum()) for z0, z1 in zs]
    nb = sum(1 for b_ in bad if b_)
    print(f"partner {name:46s}: {nb:3d} of {NA} launches differ from the solo result; differing pixels per bad launch: "
          f"{sorted(set(b_ for b_ in bad if b_))[:8]}")

# ---- victims: plain kernels (no LDS-DMA) whose result must not depend on what runs beside them
img = torch.randn(1, H, W, 4, device=DEV)
wout = torch.empty(1, H, W, 4, device=DEV)
pw = torch.randn(1, 2, H, W, device=DEV)
acth = torch.randn(1, H, W, 64, device=DEV).to(rt.tdtype)
wact = rt.act(1, H, W, 64)
torch_a, torch_b = torch.randn(1 << 22, device=DEV), torch.randn(1 << 22, device=DEV)


def v_metric(o):
    metric(o[0], o[1])


def v_warp_f32(o):
    rt.warp(View(img, 0, 4), 4, View(f01, 0, 2), View(o[0], 0, 4))


def v_warp_bf16(o):
    rt.warp(View(acth, 0, 64), 64, View(f01, 0, 2), View(o[0], 0, 64))


def v_avgpool(o):
    rt._chk(lib.avgpool2_f32(pw.data_ptr(), o[0].data_ptr(), 2, H, W, rt.stream()), "avgpool2_f32")


def v_torch_add(o):
    torch.add(torch_a, torch_b, out=o[0])


def v_torch_gather(o):
    torch.index_select(torch_a, 0, gidx, out=o[0])


gidx = torch.randint(0, 1 << 22, (1 << 22,), device=DEV)
victims = [("splat metric (3x3 + warp, float2 loads)", v_metric, lambda: (torch.empty_like(ref0), torch.empty_like(ref1))),
           ("warp_nhwc float4 image", v_warp_f32, lambda: (torch.empty_like(wout),)),
           ("warp_nhwc bf16 64 ch", v_warp_bf16, lambda: (torch.empty_like(wact),)),
           ("avgpool2 f32", v_avgpool, lambda: (torch.empty(2, H // 2, W // 2, device=DEV),)),
           ("torch.add (ATen, elementwise)", v_torch_add, lambda: (torch.empty_like(torch_a),)),
           ("torch.index_select (ATen, gather)", v_torch_gather, lambda: (torch.empty_like(torch_a),))]
NA = 100
dma = partners[7][1]        # the 1x1 LDS-DMA 128 tile
for vname, vfn, mkout in victims:
    refo = mkout()
    vfn(refo)
    torch.cuda.synchronize()
    for pname, partner in (("none", None), ("1x1 LDS-DMA 128 tile", dma), ("1 GB copy", p_copy)):
        outs_ = [mkout() for _ in range(NA)]
        torch.cuda.synchronize()
        if partner is not None:
            with torch.cuda.stream(sb):
                for _ in range(40):
                    partner()
        with torch.cuda.stream(sa):
            for o in outs_:
                vfn(o)
        torch.cuda.synchronize()
        bad = [sum(int((a_ != b_).sum()) for a_, b_ in zip(o, refo)) for o in outs_]
        print(f"victim {vname:40s} beside {pname:22s}: {sum(1 for b_ in bad if b_):3d} of {NA} launches differ; values per bad launch "
              f"{sorted(set(b_ for b_ in bad if b_))[:6]}")
