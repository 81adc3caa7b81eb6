"""Probe 2: StepsInFlight with the parallel launch sequences (graph branches) of slot 1 / both slots switched off: a linear graph runs
entirely on its launch stream.  usage: python tools/queue_probe2.py [r|f] [bench-order]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gimmvfi_hip.model import GIMMVFI_F, GIMMVFI_R, StepsInFlight  # noqa: E402
from gimmvfi_hip.params import random_state_dict, random_state_dict_f  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402

DEV = torch.device("cuda:0")
mdl = sys.argv[1] if len(sys.argv) > 1 else "f"
bench_order = len(sys.argv) > 2
B, H, W = 8, 256, 448
sd = random_state_dict_f(0) if mdl == "f" else random_state_dict(0)
SW = ("GVFI_ENC_LANES", "GVFI_POST_LANES", "GVFI_SYNTH_LANES")


def build(lanes):
    for k in SW:
        os.environ[k] = "1" if lanes else "0"
    os.environ["GVFI_RAFT_LANES"] = os.environ["GVFI_F_LANES"] = "2" if lanes else "1"
    m = (GIMMVFI_F if mdl == "f" else GIMMVFI_R)(precision="bf16")
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    m.static_outputs = True
    m.engine(DEV)
    return m


x = synthetic_pairs(B, H, W, seed=100).to(DEV)
for l0, l1 in ((True, True), (True, False), (False, False)):
    m0, m1 = build(l0), build(l1)
    coords = [(m0.sample_coord_input(B, (H, W), [0.5], device=DEV), None)]
    ts = [0.5 * torch.ones(B, device=DEV)]
    pipe = StepsInFlight(m0, depth=1)
    pipe.replicas.append(m1)
    pipe.streams.append(torch.cuda.Stream(device=DEV))
    if bench_order:
        m1(x, coords, t=ts)
        torch.cuda.synchronize()
    tab = pipe.calibrate(x, coords, ts, steps=6)
    print(f"{mdl} lanes slot0={l0} slot1={l1}:", {k: (round(v * B, 1) if isinstance(v, float) else v) for k, v in tab.items()})
    del pipe, m0, m1
    torch.cuda.empty_cache()
