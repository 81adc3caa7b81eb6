"""Generates tests/golden/*.npz by running the REAL reference (imported from /root/reference via
oracle/ref_harness.py) on seeded synthetic inputs with the seeded weights of
gimmvfi_hip.params.random_state_dict(0).  Run in the dev container only:

    python oracle/make_golden.py

The fixtures let the GPU box (which has no /root/reference) check both the oracle restatement and
the HIP path against outputs of the reference itself."""
import json
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
warnings.filterwarnings("ignore")

import ref_harness as rh  # noqa: E402
from gimmvfi_hip.params import random_state_dict  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402

CASES = {
    # name: (B, H, W, seed, ds_factor, t list)
    "r_128x192_t050": (1, 128, 192, 3, None, [0.5]),
    "r_b2_128x128_t025_075": (2, 128, 128, 4, None, [0.25, 0.75]),
    "r_256x256_ds050_t050": (1, 256, 256, 5, 0.5, [0.5]),
}


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    sd = random_state_dict(0)
    model = rh.build_reference_model(sd)
    keys = {k: list(v.shape) for k, v in model.state_dict().items()}
    json.dump(keys, open(os.path.join(out_dir, "state_dict_keys_r.json"), "w"), indent=0)
    for name, (B, H, W, seed, ds, tl) in CASES.items():
        x = synthetic_pairs(B, H, W, seed)
        o = rh.reference_forward(model, x, tl, ds)
        arrs = {"raft_flow": o["raft_flow"].numpy(), "nflow": o["nflow"].numpy()}
        for i in range(len(tl)):
            arrs[f"imgt_pred_{i}"] = o["imgt_pred"][i].numpy()
            arrs[f"flowt_{i}"] = o["flowt"][i].numpy()
            arrs[f"flowt0_4_{i}"] = o["flowt0_pred"][i][1].numpy()
        arrs["meta"] = np.array(json.dumps({"B": B, "H": H, "W": W, "seed": seed, "ds": ds, "t": tl}))
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **arrs)
        print(name, {k: v.shape for k, v in arrs.items() if k != "meta"})


if __name__ == "__main__":
    main()
