"""Per-phase cycles of the patch convolution (conv_patch.hip profiling stamps, algo bit 15).
usage: [ALGO=7] python tools/patch_timeline.py <shape filter of tools/conv_bench.py SHAPES>
ALGO=7: the column kernel (conv_col7.hip; same four phases: patch staging, K loop, epilogue to LDS, stores)"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from conv_bench import SHAPES  # noqa: E402
from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, Runtime, View  # noqa: E402

rt = Runtime(L.get(), "bf16", "cuda:0")
ALGO = int(os.environ.get("ALGO", "3"))
PAD16 = bool(os.environ.get("PAD16")) or ALGO == 7
for name, N, H, W, Cin, Cout, KH, KW, split in SHAPES:
    if sys.argv[1] not in name:
        continue
    stride = 2 if " s2 " in name else 1
    lay = ConvLayer(rt, torch.randn(Cout, Cin, KH, KW) / (Cin * KH * KW) ** 0.5, torch.randn(Cout), stride=stride)
    x = torch.zeros(N, H, W, rt.cp(Cin), device="cuda", dtype=rt.tdtype)
    x[..., :Cin] = torch.randn(N, H, W, Cin, device="cuda").to(rt.tdtype)
    out = rt.act(N, H // stride, W // stride, Cout)
    st = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    for _ in range(2):
        rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_RELU, algo=ALGO, pad16=PAD16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_RELU, algo=ALGO + 256 * 128, aux1=st, pad16=PAD16)
    e1.record()
    torch.cuda.synchronize()
    s = st.cpu().view(-1, 4).double()
    s = s[s.sum(1) > 0]
    tot = s.sum(1).mean()
    print(f"{name}: {s.shape[0]} workgroups, {e0.elapsed_time(e1) * 1e3:.0f} us; cycles per workgroup {tot:.0f} "
          f"(=> {tot / (e0.elapsed_time(e1) * 1e3):.0f} MHz): stage patch {s[:, 0].mean():.0f}, K loop {s[:, 1].mean():.0f}, "
          f"stage acc {s[:, 2].mean():.0f}, store {s[:, 3].mean():.0f}")
