"""Probe 4: StepsInFlight free-running against pipelined (two-stage graphs + events), forked and linear slots, depth 2 and 3.
usage: python tools/queue_probe4.py [r|f] [B H W ds NI]"""
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
xs = [synthetic_pairs(B, H, W, seed=100 + i).to(DEV) for i in range(3)]
coords = [(m.sample_coord_input(B, (H, W), [i / NI], device=DEV, upsample_ratio=ds), None) for i in range(1, NI)]
ts = [(i / NI) * torch.ones(B, device=DEV) for i in range(1, NI)]
dsf = None if ds == 1.0 else ds
u8 = lambda o, mm: mm.engine(DEV).rt.frames_to_u8(o["imgt_pred"][0]).clone()      # noqa: E731
ref = []
for x in xs:
    ref.append(u8(m(x, coords, t=ts, ds_factor=dsf), m))
torch.cuda.synchronize()


def rate(pipe, K=18):
    for i in range(2 * pipe.depth):
        pipe.submit(xs[i % 3], coords, ts, ds_factor=dsf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        pipe.submit(xs[i % 3], coords, ts, ds_factor=dsf)
    torch.cuda.synchronize()
    return B * (NI - 1) * K / (time.perf_counter() - t0)


for mdl_kind in ("linear", "forked"):
    pipe = StepsInFlight(m, depth=2, serial=mdl_kind == "linear")
    pipe.prime(xs[0], coords, ts, ds_factor=dsf)
    print(f"{mdl} {mdl_kind} depth 2, streams as created: K=18: " + " / ".join(f"{rate(pipe, 18):.1f}" for _ in range(2)) + "; K=4: "
          + " / ".join(f"{rate(pipe, 4):.1f}" for _ in range(4)) + "; K=8: " + " / ".join(f"{rate(pipe, 8):.1f}" for _ in range(3)))
    for trial in range(5):
        pipe.streams = [torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)]
        print(f"   two fresh streams #{trial}: K=18: " + " / ".join(f"{rate(pipe, 18):.1f}" for _ in range(2)) + "; K=4: "
              + " / ".join(f"{rate(pipe, 4):.1f}" for _ in range(3)))
    del pipe
    torch.cuda.empty_cache()
