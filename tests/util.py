"""Shared helpers for the test-suite (test infrastructure)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}


def golden_inputs(meta):
    from gimmvfi_hip.synth import synthetic_pairs
    import gimmvfi_r_oracle as orc

    x = synthetic_pairs(meta["B"], meta["H"], meta["W"], meta["seed"])
    ratio = 1.0 if meta["ds"] is None else meta["ds"]
    coords = [(orc.sample_coord_input(meta["B"], x.shape[-2:], [t], ratio), None) for t in meta["t"]]
    ts = [t * torch.ones(meta["B"]) for t in meta["t"]]
    return x, coords, ts


def psnr(a, b):
    import gimmvfi_r_oracle as orc

    return orc.psnr(a.float().cpu(), b.float().cpu())


def maxabs(a, b):
    return float((a.float().cpu() - b.float().cpu()).abs().max())


def nchw(t):
    return t.float().permute(0, 3, 1, 2)
