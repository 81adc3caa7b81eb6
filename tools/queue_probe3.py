"""Probe 3: StepsInFlight with forked / linear slot graphs at depth 1..4, then what calibrate() picks.
usage: python tools/queue_probe3.py [r|f] [B H W ds NI]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gimmvfi_hip.model import GIMMVFI_F, GIMMVFI_R, StepsInFlight  # noqa: E402
from gimmvfi_hip.params import random_state_dict, random_state_dict_f  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402

DEV = torch.device("cuda:0")
mdl = sys.argv[1] if len(sys.argv) > 1 else "f"
B, H, W = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (8, 256, 448)
ds = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
NI = int(sys.argv[6]) if len(sys.argv) > 6 else 2
m = (GIMMVFI_F if mdl == "f" else GIMMVFI_R)(precision="bf16")
m.load_state_dict(random_state_dict_f(0) if mdl == "f" else random_state_dict(0), strict=True)
m = m.to(DEV).eval()
m.static_outputs = True
_pre = [torch.cuda.Stream(device=DEV) for _ in range(int(os.environ.get("PRE_STREAMS", "0")))]
for _s in _pre:
    with torch.cuda.stream(_s):
        torch.zeros(8, device=DEV).add_(1.0)
torch.cuda.synchronize()
if os.environ.get("ENGINE_FIRST"):
    m.engine(DEV)            # (bench.py builds slot 0's engine on the default stream before the pipeline exists)
x = synthetic_pairs(B, H, W, seed=100).to(DEV)
coords = [(m.sample_coord_input(B, (H, W), [i / NI], device=DEV, upsample_ratio=ds), None) for i in range(1, NI)]
ts = [(i / NI) * torch.ones(B, device=DEV) for i in range(1, NI)]
dsf = None if ds == 1.0 else ds


def rate(pipe, K=16):
    for i in range(2 * pipe.depth):
        pipe.submit(x, coords, ts, ds_factor=dsf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        pipe.submit(x, coords, ts, ds_factor=dsf)
    torch.cuda.synchronize()
    return B * (NI - 1) * K / (time.perf_counter() - t0)


for depth, serial in ((1, False), (1, True), (2, False), (2, True), (3, True), (4, True)):
    pipe = StepsInFlight(m, depth=depth, serial=serial)
    pipe.prime(x, coords, ts, ds_factor=dsf)
    print(f"{mdl} {W}x{H} depth {depth} {'linear' if serial else 'forked'} slots on the streams they were primed on: {rate(pipe):.1f} frames/s")
    del pipe
    torch.cuda.empty_cache()
pipe = StepsInFlight(m, depth=2)
tab = pipe.calibrate(x, coords, ts, ds_factor=dsf)
scale = B * (NI - 1)
print("StepsInFlight.calibrate (frames/s):", {k: ({kk: round(vv * scale, 1) for kk, vv in v.items()} if isinstance(v, dict) else (round(v * scale, 1) if isinstance(v, float) else v))
                                              for k, v in tab.items()})
print(f"   after calibration: depth {pipe.depth}, {rate(pipe):.1f} frames/s")
