"""Soak: N steps through a calibrated StepsInFlight pipeline over four rotating batches, every step's uint8 frames and INR flows compared
bit for bit with the same batch through the model alone.  usage: python tools/inflight_soak.py [r|f] [steps]"""
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
mdl = sys.argv[1] if len(sys.argv) > 1 else "r"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
B, H, W = 8, 256, 448
m = (GIMMVFI_F if mdl == "f" else GIMMVFI_R)(precision="bf16")
m.load_state_dict(random_state_dict_f(0) if mdl == "f" else random_state_dict(0), strict=True)
m = m.to(DEV).eval()
xs = [synthetic_pairs(B, H, W, seed=70 + i).to(DEV) for i in range(4)]
coords = [(m.sample_coord_input(B, (H, W), [0.5], device=DEV), None)]
ts = [0.5 * torch.ones(B, device=DEV)]


def pack(out, mm):
    return mm.engine(DEV).rt.frames_to_u8(out["imgt_pred"][0]).clone(), out["flowt"][0].float().clone()


alone = [pack(m(x, coords, t=ts), m) for x in xs]
torch.cuda.synchronize()
m.static_outputs = True
pipe = StepsInFlight(m, depth=2)
rep = pipe.calibrate(xs[0], coords, ts, max_pairs=int(os.environ.get("MAX_PAIRS", "8")))
if os.environ.get("TABLE"):
    print({k: ({kk: round(vv * B, 1) for kk, vv in v.items()} if isinstance(v, dict) else (round(v * B, 1) if isinstance(v, float) else v)) for k, v in rep.items()})
print(f"{mdl}: calibrate picked: {rep['picked']}; depth {pipe.depth}")
bad = 0
CH = 40
for s0 in range(0, N, CH):
    hs = [pipe.submit(xs[i % 4], coords, ts, then=pack) for i in range(s0, min(N, s0 + CH))]
    for i, h in zip(range(s0, s0 + len(hs)), hs):
        fr, fl = pipe.wait(h)
        torch.cuda.synchronize()
        if not (torch.equal(fr, alone[i % 4][0]) and torch.equal(fl, alone[i % 4][1])):
            bad += 1
            print(f"   step {i}: differs ({int((fr != alone[i % 4][0]).sum())} frame values, {int((fl != alone[i % 4][1]).sum())} flow values)")
print(f"{mdl}: {N} steps in flight, {bad} differ from the model alone")
