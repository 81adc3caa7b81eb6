"""A/B of the two correlation look-up kernels (gvfi_corr_lookup: one thread per window column, global loads;
gvfi_corr_lookup_lds: windows staged through LDS) on the bench shape: 8 images x 32 x 56 queries, 4-level pyramid of
56 x 32 maps (one lane of the RAFT recurrence at 448x256, B = 8).  GPU only."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
import torch  # noqa: E402

from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import Runtime, View  # noqa: E402


def main():
    rt = Runtime(L.get(), "bf16", "cuda:0")
    for n, h, w in ((8, 32, 56), (16, 32, 56), (2, 68, 128)):
        P = h * w
        pyr = [torch.randn(n * P, P, device="cuda")]
        hh, ww = h, w
        for _ in range(3):
            pyr.append(rt.avgpool2(pyr[-1], n * P, hh, ww))
            hh, ww = hh // 2, ww // 2
        base = torch.stack(torch.meshgrid(torch.arange(w), torch.arange(h), indexing="xy"), -1).float().cuda()   # [h, w, (x, y)]
        for spread in (0.0, 3.0, 15.0):
            coords = (base[None] + torch.randn(n, h, w, 2, device="cuda") * spread).contiguous()
            outs = {}
            line = f"{n} x {h} x {w} queries, flow spread {spread:4.1f} px: "
            for lds in (False, True, False, True):
                rt.lookup_lds = lds
                out = rt.act(n, h, w, 324, zero=True, pitch=rt.cp64(324))
                for _ in range(3):
                    rt.corr_lookup(pyr, coords, View(out, 0, 324), n, h, w, h, w)
                torch.cuda.synchronize()
                torch.cuda._sleep(2_000_000)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    rt.corr_lookup(pyr, coords, View(out, 0, 324), n, h, w, h, w)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 50 * 1e3
                outs[lds] = out.clone()
                alg = n * P * (4 * 100 * 4 + 324 * 2) / 1e6
                line += f"{'lds   ' if lds else 'global'} {us:6.1f} us ({alg / us:5.2f} TB/s of {alg:.0f} MB algorithmic) | "
            print(line + ("identical" if torch.equal(outs[False], outs[True]) else "DIFFERENT"), flush=True)


if __name__ == "__main__":
    main()
