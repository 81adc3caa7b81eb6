"""The oracle restatement must reproduce the reference: (a) against the reference itself when the
checkout is present (dev container), (b) against the committed golden fixtures that were produced
by the reference (everywhere, incl. the GPU box)."""
import pytest
import torch

import gimmvfi_r_oracle as orc
import ref_harness as rh
from util import golden_inputs, load_golden, maxabs

CASES = ["r_128x192_t050", "r_b2_128x128_t025_075", "r_256x256_ds050_t050"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_golden(name, sd):
    meta, gold = load_golden(name)
    x, coords, ts = golden_inputs(meta)
    with torch.no_grad():
        o = orc.forward(sd, x, coords, ts, meta["ds"])
    # same torch build => the restatement is bit-exact; allow 1e-5 for other BLAS/oneDNN builds
    tol = 1e-5
    assert maxabs(o["raft_flow"], gold["raft_flow"]) <= tol * 10
    assert maxabs(o["nflow"], gold["nflow"]) <= tol
    for i in range(len(meta["t"])):
        assert maxabs(o["imgt_pred"][i], gold[f"imgt_pred_{i}"]) <= tol
        assert maxabs(o["flowt"][i], gold[f"flowt_{i}"]) <= tol * 10
        assert maxabs(o["flowt0_pred"][i][1], gold[f"flowt0_4_{i}"]) <= tol * 10
        assert tuple(o["flowt"][i].shape) == tuple(gold[f"flowt_{i}"].shape)  # B==1 squeeze quirk


@pytest.mark.skipif(not rh.reference_available(), reason="reference checkout only exists in the dev container")
def test_oracle_matches_reference_live(sd):
    from gimmvfi_hip.synth import synthetic_pairs

    ref = rh.build_reference_model(sd)
    x = synthetic_pairs(1, 128, 160, 11)
    tl = [0.3, 0.8]
    ro = rh.reference_forward(ref, x, tl, None)
    coords = [(orc.sample_coord_input(1, x.shape[-2:], [t], 1.0), None) for t in tl]
    with torch.no_grad():
        oo = orc.forward(sd, x, coords, [t * torch.ones(1) for t in tl], None)
    for k in ("raft_flow", "nflow"):
        assert maxabs(ro[k], oo[k]) == 0.0
    for i in range(2):
        for k in ("imgt_pred", "flowt", "ninrflow"):
            assert maxabs(ro[k][i], oo[k][i]) == 0.0
        assert maxabs(ro["other_pred"][i][0], oo["other_pred"][i][0]) == 0.0
        for j in range(2):
            assert maxabs(ro["flowt0_pred"][i][j], oo["flowt0_pred"][i][j]) == 0.0
            assert maxabs(ro["flowt1_pred"][i][j], oo["flowt1_pred"][i][j]) == 0.0


@pytest.mark.skipif(not rh.reference_available(), reason="reference checkout only exists in the dev container")
def test_reference_accepts_our_state_dict_strict(sd):
    ref = rh.build_reference_model()
    ref.load_state_dict(sd, strict=True)
    assert list(ref.state_dict().keys()) == list(sd.keys())


# ---- GIMM-VFI-F (FlowFormer flow estimator): oracle/gimmvfi_f_oracle.py
F_CASES = ["f_128x192_t050", "f_b2_128x128_t025_075", "f_136x152_t040"]


@pytest.mark.parametrize("name", F_CASES)
def test_f_oracle_matches_golden(name, sd_f):
    import gimmvfi_f_oracle as forc

    meta, gold = load_golden(name)
    x, coords, ts = golden_inputs(meta)
    with torch.no_grad():
        o = forc.forward(sd_f, x, coords, ts, meta["ds"])
    tol = 1e-5  # bit-exact on the same torch build; slack for other BLAS/oneDNN builds (32 recurrent iterations)
    assert maxabs(o["raft_flow"], gold["raft_flow"]) <= tol * 100
    assert maxabs(o["nflow"], gold["nflow"]) <= tol * 10
    for i in range(len(meta["t"])):
        assert maxabs(o["imgt_pred"][i], gold[f"imgt_pred_{i}"]) <= tol * 10
        assert maxabs(o["flowt"][i], gold[f"flowt_{i}"]) <= tol * 100
        assert tuple(o["flowt"][i].shape) == tuple(gold[f"flowt_{i}"].shape)


@pytest.mark.skipif(not rh.reference_available(), reason="reference checkout only exists in the dev container")
def test_f_oracle_matches_reference_live(sd_f):
    import gimmvfi_f_oracle as forc
    from gimmvfi_hip.synth import synthetic_pairs

    ref = rh.build_reference_model_f(sd_f)
    assert list(ref.state_dict().keys()) == list(sd_f.keys())  # strict=True load + same order as the reference
    x = synthetic_pairs(1, 136, 160, 11)     # 17 x 20 grid at 1/8: the padded-window / zero-extension branches
    tl = [0.3, 0.8]
    ro = rh.reference_forward(ref, x, tl, None)
    coords = [(forc.sample_coord_input(1, x.shape[-2:], [t], 1.0), None) for t in tl]
    with torch.no_grad():
        oo = forc.forward(sd_f, x, coords, [t * torch.ones(1) for t in tl], None)
    for k in ("raft_flow", "nflow"):
        assert maxabs(ro[k], oo[k]) == 0.0
    for i in range(2):
        for k in ("imgt_pred", "flowt", "ninrflow"):
            assert maxabs(ro[k][i], oo[k][i]) == 0.0
        for j in range(2):
            assert maxabs(ro["flowt0_pred"][i][j], oo["flowt0_pred"][i][j]) == 0.0


def test_f_param_spec_matches_recorded_reference_keys(sd_f):
    import json
    import os

    from util import GOLDEN

    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys_f.json")))
    assert list(keys.keys()) == list(sd_f.keys())
    assert all(list(sd_f[k].shape) == v for k, v in keys.items())


def _splat_thread_loop(ten_in, flow):
    """Independent statement of the reference's ONE native kernel (CuPy `softsplat_out`, modules/softsplat.py:371-421), written
    the way the kernel is: one "thread" per INPUT element intIndex = ((n * C + c) * H + y) * W + x, floor / +1 corners, the four
    bilinear weights as products of corner distances, one guarded add per corner -- no tensor operation, no scatter, float32
    arithmetic through numpy scalars.  ("cross-check it against an independent O(P) reference loop".)"""
    import numpy as np

    a = ten_in.numpy()
    f = flow.numpy()
    N, C, H, W = a.shape
    out = np.zeros_like(a)
    f32 = np.float32
    for n in range(N):
        for c in range(C):
            for y in range(H):
                for x in range(W):
                    flt_x = f32(x) + f[n, 0, y, x]
                    flt_y = f32(y) + f[n, 1, y, x]
                    if not (np.isfinite(flt_x) and np.isfinite(flt_y)):
                        continue
                    v = a[n, c, y, x]
                    nw_x = int(np.floor(flt_x))
                    nw_y = int(np.floor(flt_y))
                    ne_x, ne_y = nw_x + 1, nw_y
                    sw_x, sw_y = nw_x, nw_y + 1
                    se_x, se_y = nw_x + 1, nw_y + 1
                    w_nw = (f32(se_x) - flt_x) * (f32(se_y) - flt_y)
                    w_ne = (flt_x - f32(sw_x)) * (f32(sw_y) - flt_y)
                    w_sw = (f32(ne_x) - flt_x) * (flt_y - f32(ne_y))
                    w_se = (flt_x - f32(nw_x)) * (flt_y - f32(nw_y))
                    for tx, ty, w in ((nw_x, nw_y, w_nw), (ne_x, ne_y, w_ne), (sw_x, sw_y, w_sw), (se_x, se_y, w_se)):
                        if 0 <= tx < W and 0 <= ty < H:
                            out[n, c, ty, tx] += v * w
    return torch.from_numpy(out)


def test_splat_restatements_match_an_independent_per_element_loop():
    """Both vectorised statements of `softsplat_out` -- the oracle's (gimmvfi_r_oracle.splat_sum) and the one the reference is
    run with on the CPU (ref_harness._cpu_softsplat_out) -- against the thread-per-element loop above: sub-pixel, large,
    out-of-range, exactly integral and non-finite flows, several sources landing in one cell.  The sums are float adds in
    different orders, hence 1e-5 (atomics order => 1e-5 abs)."""
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 2, 3, 9, 11
    x = torch.randn(N, C, H, W, generator=g)
    flow = torch.randn(N, 2, H, W, generator=g) * 3.0
    flow[0, :, 0, 0] = torch.tensor([2.0, 1.0])          # integral target: one corner takes everything
    flow[0, :, 1, 1] = torch.tensor([-40.0, 3.0])        # far outside
    flow[0, :, 2, 2] = torch.tensor([float("nan"), 0.5])
    flow[1, :, 3, 3] = torch.tensor([0.25, float("inf")])
    flow[1, 0, 4, :] = 10.5 - torch.arange(W, dtype=torch.float32)     # a whole row lands in one column
    flow[1, 1, 4, :] = 0.5
    flow[0, :, H - 1, W - 1] = torch.tensor([0.5, 0.5])  # two of four corners outside
    ref = _splat_thread_loop(x, flow)
    assert float(ref.abs().max()) > 1.0
    for name, fn in (("oracle", orc.splat_sum), ("ref_harness", rh._cpu_softsplat_out)):
        got = fn(x, flow)
        assert torch.isfinite(got).all(), name
        assert maxabs(got, ref) <= 1e-5, (name, maxabs(got, ref))
