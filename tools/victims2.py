"""Victim test of the three kernels between the decoder convolution and the first combine convolution of the synthesis
(decoder_head, combine_warps_up, the 7x7 column kernel): each launched NA times on fixed inputs beside the LDS-DMA partner
convolutions of the hazard test, every result compared with the solo result.  usage: python tools/victims2.py"""
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
g = torch.Generator().manual_seed(0)
B, H, W = 8, 256, 448
HW = H * W
dec0 = torch.randn(B, H, W, 24, generator=g).to(DEV)
fl0, fl1 = torch.randn(B, H, W, 2, generator=g).to(DEV) * 4, torch.randn(B, H, W, 2, generator=g).to(DEV) * 4
mk = torch.randn(B, H, W, 1, generator=g).to(DEV)
i0, i1 = torch.randn(B, H, W, 4, generator=g).to(DEV), torch.randn(B, H, W, 4, generator=g).to(DEV)
layc = ConvLayer(rt, torch.randn(18, 9, 7, 7, generator=g) / 21, torch.randn(18, generator=g), slope=torch.rand(18, generator=g) * 0.3 + 0.1)
cw0 = torch.randn(B, H, W, 16, generator=g).to(DEV).to(rt.tdtype)
cw0[..., 9:] = 0


def v_dechead(o):
    o[0].copy_(dec0)
    rt._chk(lib.decoder_head(o[0].data_ptr(), 24, fl0.data_ptr(), fl1.data_ptr(), mk.data_ptr(), B * HW, rt.stream()), "decoder_head")


def v_combine(o):
    rt._chk(lib.combine_warps_up(i0.data_ptr(), i1.data_ptr(), dec0.data_ptr(), 24, H, W, o[0].data_ptr(), 16, 16, o[1].data_ptr(),
                                 o[2].data_ptr(), o[3].data_ptr(), B, 0, H, W, rt.dtype, rt.stream()), "combine_warps_up")


def v_col7(o):
    rt.conv(layc, View(cw0, 0, 9), View(o[0], 0, 18), act1=L.ACT_PRELU, pad16=True, algo=rt.comb_algo)


victims = [("decoder_head", v_dechead, lambda: (torch.empty_like(dec0),)),
           ("combine_warps_up (direct form)", v_combine, lambda: (torch.zeros(B, H, W, 16, device=DEV, dtype=rt.tdtype), torch.empty(B, H, W, 4, device=DEV),
                                                                  torch.empty(B, 3, 2, H, W, device=DEV), torch.empty(B, 3, 2, H, W, device=DEV))),
           ("7x7 9->18 column kernel", v_col7, lambda: (torch.zeros(B, H, W, 24, device=DEV, dtype=rt.tdtype),))]
lay1 = ConvLayer(rt, torch.randn(256, 256, 1, 1, generator=g) / 16, torch.randn(256, generator=g))
lay64 = ConvLayer(rt, torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g))
px, py = torch.randn(2, 272, 512, 256, device=DEV).to(rt.tdtype), rt.act(2, 272, 512, 256)
qx, qy = torch.randn(2, 544, 1024, 64, device=DEV).to(rt.tdtype), rt.act(2, 544, 1024, 64)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def partner():
    with torch.cuda.stream(sb):
        for i in range(120):
            if i & 1:
                rt.conv(lay1, View(px, 0, 256), py, algo=2, tile=128)
            else:
                rt.conv(lay64, View(qx, 0, 64), qy)


def mkp(fn, n):
    def f():
        with torch.cuda.stream(sb):
            for _ in range(n):
                fn()
    return f


lay3 = ConvLayer(rt, torch.randn(256, 256, 3, 3, generator=g) / 48, torch.randn(256, generator=g))
big, big2 = torch.empty(64 << 20, device=DEV), torch.empty(64 << 20, device=DEV)
EXTRA = ()
if os.environ.get("PARTNERS"):
    EXTRA = (("1x1 LDS-DMA 128 tile only", mkp(lambda: rt.conv(lay1, View(px, 0, 256), py, algo=2, tile=128), 60)),
             ("1x1 LDS-DMA 256 tile only", mkp(lambda: rt.conv(lay1, View(px, 0, 256), py, algo=2, tile=256), 60)),
             ("mid-channel 3x3 halo kernel only", mkp(lambda: rt.conv(lay64, View(qx, 0, 64), qy), 60)),
             ("1x1 register-staged igemm", mkp(lambda: rt.conv(lay1, View(px, 0, 256), py, algo=1), 60)),
             ("hot 3x3 stream kernel", mkp(lambda: rt.conv(lay3, View(px, 0, 256), py, act1=L.ACT_RELU), 30)),
             ("256 MB copies", mkp(lambda: big2.copy_(big), 20)))
NA = int(os.environ.get("NA", 60))
for vname, vfn, mkout in victims:
    refo = mkout()
    vfn(refo)
    torch.cuda.synchronize()
    for pname, part in (("none", None), ("LDS-DMA partner convolutions", partner)) + EXTRA:
        outs_ = [mkout() for _ in range(NA)]
        torch.cuda.synchronize()
        if part is not None:
            part()
        with torch.cuda.stream(sa):
            for o in outs_:
                vfn(o)
        torch.cuda.synchronize()
        bad = [sum(int((a_ != b_).sum()) for a_, b_ in zip(o, refo)) for o in outs_]
        if os.environ.get("HIST") and vname.startswith("combine"):
            hist = torch.zeros(4, dtype=torch.long)
            for o in outs_:
                d = (o[1] != refo[1]).any(-1).cpu()          # mean4: (B, H, W)
                xs = d.nonzero()[:, 2]
                hist += torch.bincount((xs % 64) // 16, minlength=4)
            print("   wrong pixels by 16-lane quarter of the wave (x % 64 // 16):", hist.tolist())
        for o in outs_:
            per = [int((a_ != b_).sum()) for a_, b_ in zip(o, refo)]
            if sum(per) and os.environ.get("WHERE"):
                print("   per output:", per)
                for a_, b_ in zip(o, refo):
                    d = (a_ != b_)
                    if d.any():
                        nz = d.nonzero()
                        print("     shape", tuple(a_.shape), "first differing indices", nz[:12].tolist(), "values", a_[d][:6].float().tolist(), "expected", b_[d][:6].float().tolist())
                break
        print(f"victim {vname:32s} beside {pname:30s}: {sum(1 for b_ in bad if b_):3d} of {NA} launches differ; values per bad launch "
              f"{sorted(set(b_ for b_ in bad if b_))[:6]}")
