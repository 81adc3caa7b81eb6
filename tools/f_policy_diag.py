"""GIMM-VFI-F: which stages of the flow estimator have to run in float for the bf16 path to match the REFERENCE fixtures?
For every hi-res reference fixture (tests/golden/hr_f_*.npz) and every precision policy of the flow estimator
(GIMMVFI_F(precision="bf16", flow_precision=...)): distance from the fixture + eager time of the forward.
usage: python tools/f_policy_diag.py [case ...] [--policies p1;p2;...]     (GPU only; test infrastructure is imported for
the fixture loader / metrics, nothing of it is product code)"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in ("gimm-vfi_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import torch  # noqa: E402

import test_gpu_hires as TH  # noqa: E402
from gimmvfi_hip.model import GIMMVFI_F  # noqa: E402
from gimmvfi_hip.params import random_state_dict_f  # noqa: E402

POLICIES = ["bf16", "tok", "upd", "dec", "fp32", "FULL-FP32"]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pol = POLICIES
    for a in sys.argv[1:]:
        if a.startswith("--policies="):
            pol = a.split("=", 1)[1].split(";")
    cases = args or TH.HR_CASES
    for name in cases:
        path = os.path.join(TH.GOLDEN, f"hr_f_{name}.npz")
        if not os.path.isfile(path):
            print("missing", path)
            continue
        import json

        import numpy as np

        z = np.load(path)
        meta = json.loads(str(z["meta"]))
        x = TH.hr_inputs(meta, z)
        # (fixtures *_fhNNN: the reference ran with the decoder's flow head damped -- small, smooth flows)
        sd = random_state_dict_f(0, flow_head_scale=meta.get("flow_head_scale", 1.0))
        for fp in pol:
            if fp == "FULL-FP32":
                m = GIMMVFI_F(precision="fp32")
            else:
                m = GIMMVFI_F(precision="bf16", flow_precision=fp)
            m.load_state_dict(sd, strict=True)
            m = m.to(TH.DEV).eval()
            m.use_graph = False
            out = TH.run_hr(m, meta, x)       # warm-up (weight packing, allocator)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = TH.run_hr(m, meta, x)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            mt = TH.hr_metrics(out, meta, z)
            print(f"{name:14s} flow_precision={fp:10s} {dt * 1e3:8.1f} ms/forward (eager)  " + TH.fmt_metrics(mt, meta), flush=True)
            del m, out
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
