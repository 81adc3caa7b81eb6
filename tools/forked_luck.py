"""How much does the one-at-a-time rate of a FORKED captured forward depend on the instance (i.e. on which hardware queues the HIP runtime
put the graph's internal branch streams)?  Builds the model several times in one process, times each instance alone, forked and linear.
usage: python tools/forked_luck.py [r|f] [instances]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gimmvfi_hip.model import GIMMVFI_F, GIMMVFI_R  # noqa: E402
from gimmvfi_hip.params import random_state_dict, random_state_dict_f  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402

DEV = torch.device("cuda:0")
mdl = sys.argv[1] if len(sys.argv) > 1 else "r"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B, H, W = 8, 256, 448
sd = random_state_dict_f(0) if mdl == "f" else random_state_dict(0)
x = synthetic_pairs(B, H, W, seed=100).to(DEV)
keep = []
for i in range(n):
    row = []
    for serial in (False, True):
        m = (GIMMVFI_F if mdl == "f" else GIMMVFI_R)(precision="bf16")
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV).eval()
        m.static_outputs, m.serial_launch = True, serial
        coords = [(m.sample_coord_input(B, (H, W), [0.5], device=DEV), None)]
        ts = [0.5 * torch.ones(B, device=DEV)]
        st = torch.cuda.Stream(device=DEV) if i % 2 else torch.cuda.current_stream(DEV)     # (odd instances: launched from a fresh stream)
        with torch.cuda.stream(st):
            for _ in range(3):
                m(x, coords, t=ts)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(12):
                m(x, coords, t=ts)
            torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 12 * 1e3)
        keep.append(m)          # (instances stay alive: their streams / graphs keep their queues)
    print(f"{mdl} instance {i} ({'fresh stream' if i % 2 else 'default stream'}): forked {row[0]:.2f} ms/step, linear {row[1]:.2f} ms/step")
