"""Probe: two independent steps in flight (two model instances, each with its own captured graph, replayed alternately on two
streams) against one step at a time -- does the latency-bound flow estimator of one batch fill under the power-bound synthesis of
the other?  usage: python tools/overlap_probe.py [r|f] [steps]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

os.environ["GIMMVFI_STATIC_OUTPUTS"] = "1"
from gimmvfi_hip.model import GIMMVFI_F, GIMMVFI_R  # noqa: E402
from gimmvfi_hip.params import random_state_dict, random_state_dict_f  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402

DEV = torch.device("cuda:0")
mdl = sys.argv[1] if len(sys.argv) > 1 else "r"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
B, H, W = 8, 256, 448
sd = random_state_dict_f(0) if mdl == "f" else random_state_dict(0)
models = []
for i in range(2):
    m = (GIMMVFI_F if mdl == "f" else GIMMVFI_R)(precision="bf16")
    m.load_state_dict(sd, strict=True)
    models.append(m.to(DEV).eval())
xs = [synthetic_pairs(B, H, W, seed=100 + i).to(DEV) for i in range(2)]
coords = [(models[0].sample_coord_input(B, (H, W), [0.5], device=DEV), None)]
ts = [0.5 * torch.ones(B, device=DEV)]
streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
outs = [None, None]


def step(i):
    s = i % 2
    with torch.cuda.stream(streams[s]):
        o = models[s](xs[s], coords, t=ts)
        outs[s] = models[s].engine(DEV).rt.frames_to_u8(o["imgt_pred"][0])


for i in range(6):
    step(i)
torch.cuda.synchronize()
ref = [o.clone() for o in outs]
for mode in ("one at a time", "two in flight", "one at a time", "two in flight"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
        if mode == "one at a time":
            streams[i % 2].synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(a, b) for a, b in zip(outs, ref))
    print(f"{mdl} {mode}: {K} steps in {dt * 1e3:.1f} ms = {dt / K * 1e3:.3f} ms/step = {B * K / dt:.1f} frames/s; outputs reproduce: {same}")
