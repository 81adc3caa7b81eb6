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


def gimm_inputs(meta):
    """Seeded flow pair for the motion-only model (what src/VTF.py:88-139 feeds GIMM): smooth forward flow f01, a
    roughly consistent backward flow f10, xs = normalize(cat(f01, -f10)), ori_flow = cat(f01, f10)."""
    import gimmvfi_r_oracle as orc

    B, H, W = meta["B"], meta["H"], meta["W"]
    g = torch.Generator().manual_seed(meta["seed"])
    low = torch.randn(B, 2, H // 8, W // 8, generator=g) * 4.0
    f01 = torch.nn.functional.interpolate(low, size=(H, W), mode="bicubic", align_corners=False)
    f10 = -f01 + torch.nn.functional.interpolate(torch.randn(B, 2, H // 8, W // 8, generator=g) * 0.5, size=(H, W),
                                                 mode="bicubic", align_corners=False)
    raw = torch.stack([f01, -f10], 2)                       # VTF.py:90
    scaler = raw.abs().max().reshape(1, 1)                  # VTF.py:124-129 (one scaler per call)
    xs = (raw / scaler + 1.0) / 2.0
    ori = torch.stack([f01, f10], 2)                        # VTF.py:137
    if meta["single"]:
        coord = orc.sample_coord_input(B, (H, W), meta["t"][:1], 1.0)
        ts = torch.tensor(meta["t"][:1])                    # VTF.py:139: one scalar time for the batch
    else:
        coord = [orc.sample_coord_input(B, (H, W), [t], 1.0) for t in meta["t"]]
        ts = [t * torch.ones(B) for t in meta["t"]]
    return xs, ori, coord, ts
