"""Per-stage GPU time of one GIMM-VFI-F forward (eager launches, HIP events on the launch stream): which part of the
FlowFormer front end / shared synthesis path a step spends its time in.  `--sim` runs the same instrumentation on the
CPU emulator (a syntax check of this tool, the numbers are meaningless there)."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [os.path.join(ROOT, "gimm-vfi_amd"), os.path.join(ROOT, "tests", "hostsim"), os.path.join(ROOT, "tests")]
import torch

from gimmvfi_hip.engine_f import EngineF
from gimmvfi_hip.params import random_state_dict_f
from gimmvfi_hip.synth import synthetic_pairs

STAGES = ["_twins", "_cost_encoder", "_flowformer", "_bidir_pyramids", "_motion_encode", "_motion_inr", "_init_upsample",
          "_final_upsample", "_synthesize"]


def main():
    sim = "--sim" in sys.argv
    B, H, W = (1, 128, 128) if sim else (8, 256, 448)
    prec = "bf16"
    if sim:
        from sim_runtime import SimRuntime

        rt = SimRuntime(prec)
    else:
        from gimmvfi_hip import lib as L
        from gimmvfi_hip.ops import Runtime

        rt = Runtime(L.get(), prec, "cuda:0")
    eng = EngineF(rt, random_state_dict_f(0))
    x = synthetic_pairs(B, H, W, 100).to(rt.device)
    ys, xs = [(0.5 + torch.arange(n)) / n * 2 - 1 for n in (H, W)]
    g = torch.stack(torch.meshgrid(torch.tensor([0.5]), ys, xs, indexing="ij"), -1)      # (1, H, W, 3) = (t, y, x)
    coords = [(g.unsqueeze(0).repeat(B, 1, 1, 1, 1).to(rt.device), None)]
    ts = [0.5 * torch.ones(B, device=rt.device)]
    acc = {}

    def wrap(name):
        fn = getattr(eng, name)

        def timed(*a, **kw):
            if sim:
                t0 = time.perf_counter()
                r = fn(*a, **kw)
                acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
                return r
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **kw)
            e1.record()
            acc.setdefault(name, []).append((e0, e1))
            return r

        setattr(eng, name, timed)

    for s in STAGES:
        wrap(s)
    reps = 1 if sim else 3
    for i in range(reps + (0 if sim else 1)):
        if i == (0 if sim else 1):
            acc.clear()                      # first GPU pass = warm-up
        eng.forward(x, coords, ts, iters=None)
    if not sim:
        torch.cuda.synchronize()
    print(f"| stage (GIMM-VFI-F {W}x{H} B={B} {prec}, eager) | calls/step | ms/step |\n|---|---|---|")
    for name, v in acc.items():
        ms = sum(v) if sim else sum(a.elapsed_time(b) for a, b in v)
        print(f"| {name} | {len(v) // reps} | {ms / reps:.3f} |")
    print("\n(_flowformer includes both _twins calls and _cost_encoder; the decoder loop = _flowformer - those)")


if __name__ == "__main__":
    main()
