"""Ring depth / tile shape sweep of the LDS-DMA convolution on the shapes of the RAFT / FlowFormer recurrence (GPU only).
One launch of the recurrence works on M = 8 images x 32 x 56 = 14 336 pixels (one of the two lanes at 448x256, B = 8):
224-448 workgroups, all resident at once -- the launch time is the latency of ONE workgroup's chain of K steps.
usage: python tools/ring_bench.py [--stamps]     (--stamps: s_memtime phase stamps of the first shape per variant)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
import torch  # noqa: E402

from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, Runtime, View  # noqa: E402

SHAPES = [
    # name, N, H, W, Cin, Cout, KH, KW, split
    ("gru zr 128+128->256 1x5", 8, 32, 56, 256, 256, 1, 5, 128),
    ("gru q  128+128->128 5x1", 8, 32, 56, 256, 128, 5, 1, 128),
    ("convc2 256->192 3x3", 8, 32, 56, 256, 192, 3, 3, None),
    ("conv 256->126 3x3", 8, 32, 56, 256, 126, 3, 3, None),
    ("fh1 128->256 3x3", 8, 32, 56, 128, 256, 3, 3, None),
    ("convc1 384->256 1x1", 8, 32, 56, 384, 256, 1, 1, None),
    ("convf2 128->64 3x3", 8, 32, 56, 128, 64, 3, 3, None),
    ("convf1 128->128 1x1", 8, 32, 56, 128, 128, 1, 1, None),
    ("fh2 256->18 1x1", 8, 32, 56, 256, 18, 1, 1, None),
    ("F gru zr 128+256->256 1x5", 8, 32, 56, 384, 256, 1, 5, 128),
    ("F tok 128->128 1x1 rows", 1, 1, 14336, 128, 128, 1, 1, None),
    ("F tok 64+128->128 1x1 rows", 1, 1, 14336, 192, 128, 1, 1, 64),
    ("both lanes: gru zr @16 images", 16, 32, 56, 256, 256, 1, 5, 128),
]


def variants(cout):
    """(bm, bn, ring depth); depth 0 = the weights-direct variant (w_layout 2, register ring of 4)"""
    if cout > 64:
        return [(64, 128, 2), (64, 128, 0), (128, 128, 0)]
    if cout > 32:
        return [(128, 64, 2), (64, 128, 0), (128, 128, 0)]
    return [(128, 32, 2), (64, 128, 0)]


def main():
    stamps_mode = "--stamps" in sys.argv
    rt = Runtime(L.get(), "bf16", "cuda:0")
    only = os.environ.get("RING_ONLY")            # PMC passes: one shape (substring of its name) ...
    wdir_only = os.environ.get("RING_WDIR_ONLY")  # ... and only the weights-direct variant of the tile the engine picks
    for name, N, H, W, Cin, Cout, KH, KW, split in SHAPES:
        if only and only not in name:
            continue
        w = torch.randn(Cout, Cin, KH, KW) / (Cin * KH * KW) ** 0.5
        lay = ConvLayer(rt, w, torch.randn(Cout), wdir=True)
        x = torch.randn(N, H, W, Cin, device="cuda").to(rt.tdtype)
        if split is None:
            x0, x1 = View(x, 0, Cin), None
        else:
            xa, xb = x[..., :split].contiguous(), x[..., split:].contiguous()
            x0, x1 = View(xa, 0, split), View(xb, 0, Cin - split)
        out = rt.act(N, H, W, Cout)
        flops = 2.0 * N * H * W * Cout * Cin * KH * KW
        ref = None
        cells = []
        for bm, bn, ns in variants(Cout):
            if wdir_only and not (ns == 0 and bm == 64):
                continue
            tile = bn | (bm << 10) | (ns << 20)
            algo = 2 if ns else 6
            try:
                for _ in range(3):
                    rt.conv(lay, x0, out, x1=x1, act1=L.ACT_RELU, algo=algo, tile=tile)
            except RuntimeError as e:
                cells.append(f"{bm}x{bn}/s{ns} n/a")
                continue
            torch.cuda.synchronize()
            reps = 20
            torch.cuda._sleep(2_000_000)     # backlog: the launches below run back to back
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                rt.conv(lay, x0, out, x1=x1, act1=L.ACT_RELU, algo=algo, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / reps * 1e3
            o = out.float().clone()
            if ref is None:
                ref = o
            d = float((o - ref).abs().max())
            cells.append(f"{bm}x{bn}/{'s%d' % ns if ns else 'wdir'} {us:6.1f} us {flops / us / 1e6:6.0f} TF/s" + ("" if d == 0 else f" d={d:.1e}"))
            if stamps_mode:
                st = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
                rt.conv(lay, x0, out, x1=x1, act1=L.ACT_RELU, algo=algo + 256 * 128, tile=tile, aux1=st)
                torch.cuda.synchronize()
                s = st.cpu().view(-1, 16)
                s = s[s[:, 0] != 0].double()
                us_ = (s - s[:, 0].min()) / 100.0
                kt = KH * KW * (Cin // 64)
                cells[-1] += (f" [wg {s.shape[0]}: decode {float((us_[:, 7] - us_[:, 0]).mean()):.2f}, offs {float((us_[:, 8] - us_[:, 7]).mean()):.2f}, issue {float((us_[:, 9] - us_[:, 8]).mean()):.2f}, gc {float((us_[:, 1] - us_[:, 9]).mean()):.2f}, prologue {float((us_[:, 1] - us_[:, 0]).mean()):.2f}, chunk0 +{float((us_[:, 2] - us_[:, 1]).mean()):.2f}, "
                              f"K loop {float((us_[:, 3] - us_[:, 2]).mean()):.2f} = {float((us_[:, 3] - us_[:, 2]).mean()) / kt:.3f}/step x {kt}, "
                              f"epilogue {float((us_[:, 5] - us_[:, 3]).mean()):.2f}, span {float(us_[:, 5].max()):.1f} us]")
        print(f"{name:34s} " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
