"""Per-phase shader-clock cycles of the mid-channel halo-staged 3x3 convolution (conv_p3x3s.hip <64,64>, profiling build)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
import torch  # noqa: E402

from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, Runtime, View  # noqa: E402

rt = Runtime(L.get(), "bf16", "cuda:0")
N, H, W, Cin, Cout = 8, 256, 448, 64, 64
lay = ConvLayer(rt, torch.randn(Cout, Cin, 3, 3) / (Cin * 9) ** 0.5, torch.randn(Cout))
x = torch.randn(N, H, W, Cin, device="cuda").to(rt.tdtype)
out = rt.act(N, H, W, Cout)
st = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
for rep in range(3):
    for _ in range(3):
        rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_RELU, algo=5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.zero_()
    e0.record()
    rt.conv(lay, View(x, 0, Cin), out, act1=L.ACT_RELU, algo=5 + 256 * 128, aux1=st)
    e1.record()
    torch.cuda.synchronize()
    s = st.cpu().view(-1, 4)
    s = s[s[:, 1] > 0].double()
    us = e0.elapsed_time(e1) * 1e3
    span = (s[:, 3] + s[:, 0] + s[:, 1] + s[:, 2]).max() - s[:, 3].min()
    print(f"64->64 8x256x448: {s.shape[0]} workgroups, {us:.0f} us; cycles per workgroup {s[:, :3].sum(1).mean():.0f}: prologue {s[:, 0].mean():.0f}, "
          f"K loop {s[:, 1].mean():.0f} (9 taps x 16 MFMAs x 32 cycles = 4608 per wave), epilogue {s[:, 2].mean():.0f}; first start -> last end "
          f"{span:.0f} cycles (=> {span / us:.0f} MHz)")
