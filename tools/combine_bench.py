"""What bounds gvfi_combine_warps_up at 4K x 7 timesteps (3.56 ms, 8.4 GB = 2.4 TB/s against 4.6 TB/s for a plain copy)?
Times the launch with / without its planar flow outputs and with 1 / 7 source images.  usage: python tools/combine_bench.py"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
import torch  # noqa: E402

from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import Runtime  # noqa: E402

rt = Runtime(L.get(), "bf16", "cuda:0")
TB, H, W, Hf, Wf = 7, 544, 1024, 2176, 4096
g = torch.Generator(device="cuda").manual_seed(0)
dec = torch.randn(TB, H, W, 24, device="cuda", generator=g)
dec[..., :12] *= 3.0
dec[..., 12:15] = torch.sigmoid(dec[..., 12:15])
i0 = torch.rand(TB, Hf, Wf, 4, device="cuda", generator=g) * 2 - 1
i1 = torch.rand(TB, Hf, Wf, 4, device="cuda", generator=g) * 2 - 1
cw = rt.act(TB, Hf, Wf, 9, zero=False)
mean4 = rt.f32(TB, Hf, Wf, 4)
f0, f1 = rt.f32(TB, 3, 2, Hf, Wf), rt.f32(TB, 3, 2, Hf, Wf)


def run(planar, src_b):
    rt._chk(rt.lib.combine_warps_up(i0.data_ptr(), i1.data_ptr(), dec.data_ptr(), 24, H, W, cw.data_ptr(), cw.shape[-1], cw.shape[-1],
                                    mean4.data_ptr(), f0.data_ptr() if planar else None, f1.data_ptr() if planar else None, TB, src_b,
                                    Hf, Wf, rt.dtype, rt.stream()), "combine_warps_up")


for planar in (True, False):
    for src_b in (1, 0):
        for _ in range(2):
            run(planar, src_b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run(planar, src_b)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        wr = TB * Hf * Wf * (32 + 16 + (48 if planar else 0)) / 1e9
        print(f"planar flows {'on ' if planar else 'off'}, {'1 source image (modulo)' if src_b else '7 source images'}: {ms:.3f} ms, "
              f"{wr:.2f} GB written = {wr / ms:.2f} TB/s of writes")
