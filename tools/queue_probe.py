"""Probe: do the two slot streams of StepsInFlight run concurrently on the device, or do they alias onto one hardware queue?  Two long
spin kernels, one per stream, take 1x their duration when the streams are independent and 2x when they share a queue.  Then the
step rate with slot 1 on a series of freshly drawn streams.  usage: python tools/queue_probe.py [r|f]"""
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
B, H, W = 8, 256, 448
m = (GIMMVFI_F if mdl == "f" else GIMMVFI_R)(precision="bf16")
m.load_state_dict(random_state_dict_f(0) if mdl == "f" else random_state_dict(0), strict=True)
m = m.to(DEV).eval()
m.static_outputs = True
pipe = StepsInFlight(m, depth=2)
xs = [synthetic_pairs(B, H, W, seed=100 + i).to(DEV) for i in range(2)]
coords = [(m.sample_coord_input(B, (H, W), [0.5], device=DEV), None)]
ts = [0.5 * torch.ones(B, device=DEV)]
SPIN = 30_000_000


def spin_ratio(sa, sb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        torch.cuda._sleep(SPIN)
    torch.cuda.synchronize()
    one = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(sa):
        torch.cuda._sleep(SPIN)
    with torch.cuda.stream(sb):
        torch.cuda._sleep(SPIN)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / one


def rate(K=12):
    for i in range(4):
        pipe.submit(xs[i % 2], coords, ts, then=lambda o, mm: mm.engine(DEV).rt.frames_to_u8(o["imgt_pred"][0]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        pipe.submit(xs[i % 2], coords, ts, then=lambda o, mm: mm.engine(DEV).rt.frames_to_u8(o["imgt_pred"][0]))
    torch.cuda.synchronize()
    return B * K / (time.perf_counter() - t0)


print(f"{mdl}: slot streams as created: spin ratio {spin_ratio(*pipe.streams):.2f} (1 = concurrent, 2 = one queue); {rate():.1f} frames/s")
for trial in range(8):
    pipe.streams[1] = torch.cuda.Stream(device=DEV)
    r = spin_ratio(*pipe.streams)
    print(f"   slot 1 on fresh stream #{trial}: spin ratio {r:.2f}; {rate():.1f} frames/s")
hp = torch.cuda.Stream(device=DEV, priority=-1)
pipe.streams[1] = hp
print(f"   slot 1 on a high-priority stream: spin ratio {spin_ratio(*pipe.streams):.2f}; {rate():.1f} frames/s")
pipe.streams[0] = torch.cuda.Stream(device=DEV, priority=-1)
print(f"   both slots on high-priority streams: spin ratio {spin_ratio(*pipe.streams):.2f}; {rate():.1f} frames/s")
pipe = StepsInFlight(m, depth=2)
tab = pipe.calibrate(xs[0], coords, ts)
print("StepsInFlight.calibrate (steps/s):", tab)
print(f"   after calibration: depth {pipe.depth}, {rate(20):.1f} frames/s")
