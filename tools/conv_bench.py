"""A/B micro-benchmark of the convolution kernels on production shapes (GPU only).
usage: python tools/conv_bench.py [precision]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
import torch  # noqa: E402

from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, Runtime, View  # noqa: E402

SHAPES = [
    # name, N, H, W, Cin, Cout, KH, KW, split
    ("final.resblock 256->256 3x3 @256x448 B8", 8, 256, 448, 256, 256, 3, 3, None),
    ("final.resblock 192+64->256 3x3", 8, 256, 448, 256, 256, 3, 3, 192),
    ("final.side 64->64 3x3", 8, 256, 448, 64, 64, 3, 3, None),
    ("final.side 64->64 3x3 @544x1024 B1 (4K)", 1, 544, 1024, 64, 64, 3, 3, None),
    ("side 64->32 3x3 @256x448 B8", 8, 256, 448, 64, 32, 3, 3, None),
    ("final.resblock 256->256 3x3 @544x1024 B2 (2K/4K)", 2, 544, 1024, 256, 256, 3, 3, None),
    ("final.resblock 256->256 3x3 ragged @250x443 B4", 4, 250, 443, 256, 256, 3, 3, None),
    ("raft gru 128+256->256 1x5 @32x56 B16", 16, 32, 56, 384, 256, 1, 5, 128),
    ("raft convc2 256->192 3x3", 16, 32, 56, 256, 192, 3, 3, None),
    ("raft gruq 128+256->128 5x1", 16, 32, 56, 384, 128, 5, 1, 128),
    ("raft fh1 128->256 3x3", 16, 32, 56, 128, 256, 3, 3, None),
    ("raft convc1 324(384)->256 1x1", 16, 32, 56, 384, 256, 1, 1, None),
    ("init.resblock 128->128 3x3 @64x112 B8", 8, 64, 112, 128, 128, 3, 3, None),
    ("upd_high 320->192 3x3 @64x112", 8, 64, 112, 320, 192, 3, 3, None),
    ("cnn_enc 32->32 3x3 @256x448 B16", 16, 256, 448, 32, 32, 3, 3, None),
    ("fnet 64->64 3x3 @128x224 B16", 16, 128, 224, 64, 64, 3, 3, None),
    ("fnet 96->96 3x3 @64x112 B16", 16, 64, 112, 96, 96, 3, 3, None),
    ("final.up 32->64 3x3 @256x448 B16", 16, 256, 448, 32, 64, 3, 3, None),
    ("final.head 256->24 3x3 @256x448 B8", 8, 256, 448, 256, 24, 3, 3, None),
    ("raft fh2 256->2 3x3 @32x56 B16", 16, 32, 56, 256, 2, 3, 3, None),
    # few-channel layers at full resolution (generic kernel = algo 1, patch kernel = algo 3)
    ("comb0 9->18 7x7 @4K", 1, 2176, 4096, 9, 18, 7, 7, None),
    ("comb2 18->3 7x7 @4K", 1, 2176, 4096, 18, 3, 7, 7, None),
    ("comb0 9->18 7x7 @256x448 B8", 8, 256, 448, 9, 18, 7, 7, None),
    ("stem 3->64 7x7 s2 @256x448 B16", 16, 256, 448, 3, 64, 7, 7, None),
    ("dec.up 8->32 5x5 @256x448 B16", 16, 256, 448, 8, 32, 5, 5, None),
]


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    only = sys.argv[2] if len(sys.argv) > 2 else None
    rt = Runtime(L.get(), prec, "cuda:0")
    for name, N, H, W, Cin, Cout, KH, KW, split in SHAPES:
        if only is not None and only not in name:
            continue
        w = torch.randn(Cout, Cin, KH, KW) / (Cin * KH * KW) ** 0.5
        stride = 2 if " s2 " in name else 1
        lay = ConvLayer(rt, w, torch.randn(Cout), stride=stride)
        x = torch.zeros(N, H, W, rt.cp(Cin), device="cuda", dtype=rt.tdtype)
        x[..., :Cin] = torch.randn(N, H, W, Cin, device="cuda").to(rt.tdtype)
        if split is None:
            x0, x1 = View(x, 0, Cin), None
        else:
            xa = x[..., :split].contiguous()
            xb = x[..., split:].contiguous()
            x0, x1 = View(xa, 0, split), View(xb, 0, Cin - split)
        out = rt.act(N, H // stride, W // stride, Cout)
        flops = 2.0 * N * (H // stride) * (W // stride) * Cout * Cin * KH * KW
        res = {}
        outs = {}
        # (2, 256) = 2 x 128-byte stages, DMA pieces front-loaded; +32 = pieces spread over the MFMA groups; +128 = 4 x 64-byte stages
        # 4 = halo-staged 3x3 kernel (conv_p3x3.hip); the LDS-DMA default is timed again after it (DVFS drift within the call)
        variants = ((2, 256), (4, 0), (2, 256 | (1 << 20)), (4, 1 << 20)) if os.environ.get("P3") else ((2, 0), (2, 256), (2 + 32, 256), (2 + 128, 256))
        with_res = bool(os.environ.get("RES"))
        resid = rt.act(N, H // stride, W // stride, Cout) if with_res else None
        if Cin < 32:
            variants = ((1, 0), (3, 0))
            if KH == 7 and stride == 1:      # + the column kernel (conv_col7.hip); PAD16=1 gives the patch kernel the same store contract
                variants = ((1, 0), (3, 0), (7, 0), (3, 1 << 20), (7, 1 << 20))
        if os.environ.get("P3S"):        # LDS-DMA kernel against the mid-channel halo-staged kernel (conv_p3x3s.hip)
            variants = ((2, 0), (5, 0), (2, 1 << 20), (5, 1 << 20))
        if os.environ.get("PATCH64"):    # LDS-DMA kernel against the patch kernel on the <= 64-channel layers
            variants = ((2, 0), (3, 0), (2, 1 << 20), (3, 1 << 20))
        if os.environ.get("ABLATE0"):   # prologue / K loop / epilogue split on the auto tile
            variants = tuple((2 + 256 * m, 0) for m in (0, 8, 16, 24, 32))
        if os.environ.get("ONLY256"):     # single variant for PMC passes
            variants = ((2, 256),)
        if os.environ.get("ONLYP3"):
            variants = ((4, 0),)
        if os.environ.get("ABLATE"):
            variants = tuple((2 + 256 * m, 256) for m in (0, 0, 8, 16, 24))
        for algo, tile in variants:
            if (tile & 1023) == 256 and Cout < 192:
                continue
            if (tile & 1023) == 128 and Cout <= 64:
                tile = 0
            try:
                kw = dict(pad16=True) if ((algo == 3 and os.environ.get("PAD16")) or algo == 7) else {}
                if with_res:
                    kw.update(res=resid, act2=L.ACT_LRELU)
                for _ in range(2):
                    rt.conv(lay, x0, out, x1=x1, act1=L.ACT_RELU, algo=algo, tile=tile & 0xfffff, **kw)
            except RuntimeError:
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                rt.conv(lay, x0, out, x1=x1, act1=L.ACT_RELU, algo=algo, tile=tile & 0xfffff, **kw)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            res[(algo, tile)] = (ms, flops / ms / 1e9)
            outs[(algo, tile)] = out.float().clone()
        if not outs:
            continue
        ref = outs[(1, 0)] if (1, 0) in outs else next(iter(outs.values()))
        txt = " | ".join(f"a{k[0]}t{k[1] & 1023}m{(k[1] >> 10) & 1023}{chr(39) if k[1] >> 20 else str()} {v[0]:7.3f} ms {v[1]:6.1f} TF/s d={float((outs[k]-ref).abs().max()):.1e}" for k, v in res.items())
        print(f"{name:42s} {txt}")


if __name__ == "__main__":
    main()
