"""TEST INFRASTRUCTURE.  Generates tests/golden/f448_fh<100*S>.npz: outputs of the REAL reference GIMM-VFI-F (imported from
/root/reference through oracle/ref_harness.py, CPU fp32) on the first samples of bench.py's 448x256 batch (BASELINE.json
configs[3]: seed 100, t = 0.5), with the FlowFormer decoder's flow head of the seeded weights multiplied by S
(params.random_state_dict_f(0, flow_head_scale=S)).

Why a family: no trained checkpoint exists offline, and the seeded random weights give the 32-iteration recurrence 30-50 px
flows full of fold-overs -- far rougher than a trained estimator's.  S in {0.15, 0.4, 0.7, 1.0} walks the maximum flow from a
few pixels to that extreme: the curve "fidelity of the bf16 path against max |flow|" (profiles/r4_f_flow_scale_curve.md) is
what a user with a real checkpoint sits on (VERDICT r3 #5b).

Per sample: the predicted frame as uint8 (round(x * 255)), the INR flow every 2nd pixel as fp16, max |flow|.

    python oracle/make_golden_f448.py [S ...]        (default: 0.15 0.4 0.7 1.0; 4 samples each)
"""
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
warnings.filterwarnings("ignore")

import ref_harness as rh  # noqa: E402
from gimmvfi_hip.params import random_state_dict_f  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402

NS = 4


def main():
    scales = [float(a) for a in sys.argv[1:]] or [0.15, 0.4, 0.7, 1.0]
    x = synthetic_pairs(8, 256, 448, seed=100)[:NS]          # bench.py's rank-0 batch
    torch.set_num_threads(os.cpu_count())
    for s in scales:
        model = rh.build_reference_model_f(random_state_dict_f(0, flow_head_scale=s))
        arrs, fmax = {}, []
        t0 = time.time()
        for b in range(NS):
            o = rh.reference_forward(model, x[b:b + 1], [0.5], None)
            img = o["imgt_pred"][0][0].float()
            ft = o["flowt"][0]
            ft = ft if ft.dim() == 3 else ft[0]
            arrs[f"img_{b}"] = torch.round(img.clamp(0, 1) * 255.0).to(torch.uint8).numpy()
            arrs[f"flowt_{b}"] = ft[:, ::2, ::2].numpy().astype(np.float16)
            fmax.append(float(ft.abs().max()))
        meta = {"model": "f", "flow_head_scale": s, "samples": NS, "seed": 100, "H": 256, "W": 448, "t": 0.5,
                "in_sum": int(torch.round(x * 255.0).to(torch.int64).sum()), "flow_absmax": fmax,
                "ref_cpu_seconds": round(time.time() - t0, 1), "ref_cpu_threads": torch.get_num_threads()}
        arrs["meta"] = np.array(json.dumps(meta))
        path = os.path.join(ROOT, "tests", "golden", f"f448_fh{int(round(s * 100)):03d}.npz")
        np.savez_compressed(path, **arrs)
        print(f"S={s}: {meta['ref_cpu_seconds']} s, {os.path.getsize(path) // 1024} KiB, max |flow| {[round(v, 1) for v in fmax]}", flush=True)


if __name__ == "__main__":
    main()
