"""Per-phase shader-clock cycles of the halo-staged 3x3 convolution (conv_p3x3.hip, profiling build: algo bit 15).
usage: python tools/p3x3_timeline.py [shape filter of tools/conv_bench.py SHAPES]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from conv_bench import SHAPES  # noqa: E402
from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, Runtime, View  # noqa: E402

flt = sys.argv[1] if len(sys.argv) > 1 else "final.resblock 256->256"
rt = Runtime(L.get(), "bf16", "cuda:0")
for name, N, H, W, Cin, Cout, KH, KW, split in SHAPES:
    if flt not in name or split is not None:
        continue
    lay = ConvLayer(rt, torch.randn(Cout, Cin, KH, KW) / (Cin * KH * KW) ** 0.5, torch.randn(Cout))
    x = torch.randn(N, H, W, Cin, device="cuda").to(rt.tdtype)
    out = rt.act(N, H, W, Cout)
    res = rt.act(N, H, W, Cout)
    st = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    for with_res, abl in ((False, 0), (False, 0), (True, 0), (True, 0)):
        kw = dict(res=res, act2=L.ACT_LRELU) if with_res else {}
        for _ in range(3):
            rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_RELU, algo=4, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.zero_()
        e0.record()
        rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_RELU, algo=4 + 256 * 128 + abl, aux1=st, **kw)
        e1.record()
        torch.cuda.synchronize()
        raw = st.cpu().view(-1, 4)
        own = (raw[:, 2] >> 32).double()
        raw[:, 2] &= 0xffffffff
        s = raw.double()
        own = own[s[:, 1] > 0]
        s = s[s[:, 1] > 0]
        us = e0.elapsed_time(e1) * 1e3
        tot = (s[:, 0] + s[:, 1] + s[:, 3]).mean()
        waves = s.shape[0] / 256.0
        print(f"{name}{' +res' if with_res else ''}: {s.shape[0]} workgroups ({waves:.1f} per CU), {us:.0f} us; cycles per workgroup "
              f"{tot:.0f} (=> {tot * waves / us:.0f} MHz if back to back): prologue {s[:, 0].mean():.0f}, K loop {s[:, 1].mean():.0f} "
              f"(of which wave 0 waits {s[:, 2].mean():.0f} for DMA + barrier, {own.mean():.0f} of that for its own DMA pieces; 36 steps x 32 MFMA x 2 waves x 32 cycles = 73728 busy), epilogue {s[:, 3].mean():.0f}")
