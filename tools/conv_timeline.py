"""Phase timeline of one LDS-DMA convolution launch from in-kernel s_memtime stamps (algo profiling bit 128).
usage: python tools/conv_timeline.py <shape filter of tools/conv_bench.py SHAPES> [tile]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from conv_bench import SHAPES  # noqa: E402
from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, Runtime, View  # noqa: E402


def main():
    filt = sys.argv[1]
    tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rt = Runtime(L.get(), "bf16", "cuda:0")
    for name, N, H, W, Cin, Cout, KH, KW, split in SHAPES:
        if filt not in name:
            continue
        w = torch.randn(Cout, Cin, KH, KW) / (Cin * KH * KW) ** 0.5
        lay = ConvLayer(rt, w, torch.randn(Cout))
        x = torch.randn(N, H, W, Cin, device="cuda").to(rt.tdtype)
        out = rt.act(N, H, W, Cout)
        stamps = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
        for _ in range(3):
            rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_RELU, algo=2, tile=tile)
        torch.cuda.synchronize()
        stamps.zero_()
        rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_RELU, algo=2 + 256 * 128, tile=tile, aux1=stamps)
        torch.cuda.synchronize()
        s = stamps.cpu().view(-1, 16)
        used = s[:, 0] != 0
        s = s[used].double()
        t0 = s[:, 0].min()
        # s_memtime ticks at a constant 100 MHz on gfx9: report microseconds
        us = (s - t0) / 100.0
        names = ["start", "prologue done", "chunk0 landed", "K loop done", "staged", "stores issued", "stores acked"]
        print(f"{name}: {s.shape[0]} workgroups; kernel span {float(us[:, 6].max()):.1f} us")
        print(f"  (of the prologue: start -> row decode + tap masks done: {float((us[:, 7] - us[:, 0]).mean()):7.2f})")
        for k, nm in enumerate(names):
            col = us[:, k]
            print(f"  {nm:15s} mean {float(col.mean()):7.2f}  min {float(col.min()):7.2f}  max {float(col.max()):7.2f}"
                  + (f"   (+{float((us[:, k] - us[:, k - 1]).mean()):6.2f} per workgroup)" if k else ""))


if __name__ == "__main__":
    main()
