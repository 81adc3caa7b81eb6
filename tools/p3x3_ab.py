"""Interleaved A/B of the launch forms of the halo-staged 3x3 kernel (conv_p3x3.hip; algo bits 13, 14: 1 = round-2 kernel, 2 = tile
per workgroup + wave-private epilogue, 3 = persistent stream kernel, 0 = what gvfi_conv2d picks) on the hot layer: same process, same tensors, ROUNDS rounds of REPS launches per variant, median / min per variant, outputs compared bit
for bit with variant 0, and the per-phase cycle stamps of the profiling build.
usage: python tools/p3x3_ab.py [variants, comma separated; default 1,2,3] [shape filter]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from conv_bench import SHAPES  # noqa: E402
from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, Runtime, View  # noqa: E402

# a variant is "V" or "V:D" (D: de-phasing knob of the stream kernel, tile_hint bits 10..19)
variants = [v for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")]
VN = lambda v: int(v.split(":")[0])
VD = lambda v: (int(v.split(":")[1]) << 10) if ":" in v else 0
flt = sys.argv[2] if len(sys.argv) > 2 else "final.resblock 256->256"
ROUNDS, REPS = 7, 10
rt = Runtime(L.get(), "bf16", "cuda:0")
torch.manual_seed(0)
for name, N, H, W, Cin, Cout, KH, KW, split in SHAPES:
    if flt not in name or split is not None:
        continue
    lay = ConvLayer(rt, torch.randn(Cout, Cin, KH, KW) / (Cin * KH * KW) ** 0.5, torch.randn(Cout), slope=torch.rand(Cout) * 0.3 + 0.1)
    # PReLU-shaped activations (what the layer sees in the forward), not plain randn: the shader clock follows the operand statistics
    x = torch.nn.functional.prelu(torch.randn(N, H, W, Cin, device="cuda"), torch.tensor(0.2, device="cuda")).to(rt.tdtype)
    res = torch.randn(N, H, W, Cout, device="cuda").to(rt.tdtype)
    flops = 2.0 * N * H * W * Cout * Cin * KH * KW
    st = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    for with_res in (False, True):
        kw = dict(res=res, act2=L.ACT_PRELU, slope2=lay.slope) if with_res else {}
        outs = {}
        times = {v: [] for v in variants}
        for v in variants:
            out = rt.act(N, H, W, Cout)
            out.fill_(3.0)
            rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_PRELU, algo=4 + (VN(v) << 13), tile=VD(v), **kw)
            torch.cuda.synchronize()
            outs[v] = out.clone()
        for _ in range(ROUNDS):
            for v in variants:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(REPS):
                    rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_PRELU, algo=4 + (VN(v) << 13), tile=VD(v), **kw)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / REPS)
        print(f"{name}{' +res' if with_res else ''}")
        for v in variants:
            t = sorted(times[v])
            med, mn = t[len(t) // 2], t[0]
            same = torch.equal(outs[v], outs[variants[0]])
            # cycle stamps of the profiling build
            st.zero_()
            rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_PRELU, algo=4 + (VN(v) << 13) + 256 * 128, tile=VD(v), aux1=st, **kw)
            torch.cuda.synchronize()
            raw = st.cpu().view(-1, 8 if VN(v) in (0, 3) else 4)      # (the stream kernel writes 8 words per workgroup)
            own = (raw[:, 2] >> 32).double()
            raw[:, 2] &= 0xffffffff
            ntile = (raw[:, 0] >> 32).double()          # (stream kernel: sums over a workgroup's tiles, tile count in word 0)
            raw[:, 0] &= 0xffffffff
            s = raw.double()
            keep = s[:, 1] > 0
            s, ntile = s[keep], ntile[keep]
            if float(ntile.sum()) > 0:
                s = s.sum(0, keepdim=True) / ntile.sum()
                s = s.repeat(int(ntile.sum()), 1)
            tot = (s[:, 0] + s[:, 1] + s[:, 3]).mean()
            print(f"  variant {v}: median {med * 1e3:7.1f} us  min {mn * 1e3:7.1f} us  {flops / med / 1e9:7.1f} TFLOP/s = {flops / med / 1e9 / 2500:.4f} of 2.5 PF"
                  f"  bit-identical to v{variants[0]}: {same} | cycles per tile {tot:.0f}: prologue {s[:, 0].mean():.0f}, K loop {s[:, 1].mean():.0f}"
                  f" (wave 0 waits {s[:, 2].mean():.0f}), epilogue {s[:, 3].mean():.0f}" + (f" (own part {s[:, 4].mean():.0f})" if VN(v) in (0, 3) else "") + f"; {s.shape[0]} tiles")
