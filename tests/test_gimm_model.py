"""Motion-only model GIMM (reference generalizable_INR/gimm.py, driver src/VTF.py).

* oracle/gimm_oracle.py is pinned bit-exact against the real reference class (live, dev container) and against
  tests/golden/gimm_*.npz (outputs of the reference, everywhere);
* the engine's motion path is driven on CPU through the emulator runtime and on the GPU through the model API.
Tolerances on the normalised flow (values in [0,1]): fp32 mode 2e-4 max; bf16 mode mean 2e-3, p99.9 0.05.
"""
import json
import os

import pytest
import torch

import gimm_oracle as go
from util import GOLDEN, gimm_inputs, load_golden

CASES = ["gimm_b2_96x160_t050", "gimm_b1_128x128_t025_075"]


def _sd():
    from gimmvfi_hip.params import gimm_state_dict, random_state_dict

    return gimm_state_dict(random_state_dict(0))


def _outs(o):
    return o if isinstance(o, list) else [o]


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_golden_bit_exact(name):
    meta, gold = load_golden(name)
    xs, ori, coord, ts = gimm_inputs(meta)
    with torch.no_grad():
        out = _outs(go.forward(_sd(), xs, coord, ori, ts))
    for i, o in enumerate(out):
        assert torch.equal(o, gold[f"out_{i}"]), float((o - gold[f"out_{i}"]).abs().max())


def test_oracle_matches_reference_live_and_key_contract():
    import ref_harness as rh

    if not rh.reference_available():
        pytest.skip("reference checkout not present (GPU box)")
    import importlib

    rh.load_reference_modules()
    gm = importlib.import_module(rh._PKG + ".generalizable_INR.gimm")
    cfg = rh.default_arch_config()
    cfg["type"] = "gimm"
    model = gm.GIMM(cfg).eval()
    sd = _sd()
    model.load_state_dict(sd, strict=True)           # identical key set / shapes
    meta = dict(B=1, H=64, W=96, seed=21, t=[0.3], single=True)
    xs, ori, coord, ts = gimm_inputs(meta)
    with torch.no_grad():
        assert torch.equal(model(xs, coord, ori_flow=ori, timesteps=ts), go.forward(sd, xs, coord, ori, ts))


def test_state_dict_keys_and_factory():
    import sys

    from util import ROOT

    sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd", "src"))
    from models import create_model

    model, ema = create_model({"type": "gimm", "coord_range": [-1.0, 1.0]})
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys_gimm.json")))
    assert ema is None and {k: list(v.shape) for k, v in model.state_dict().items()} == keys
    with pytest.raises(RuntimeError):                 # no CPU fallback
        model(torch.zeros(1, 2, 2, 32, 32), torch.zeros(1, 1, 32, 32, 3), ori_flow=torch.zeros(1, 2, 2, 32, 32),
              timesteps=torch.tensor([0.5]))


def _check(out, gold, precision):
    for i, o in enumerate(out):
        d = (o.float().cpu() - gold[f"out_{i}"]).abs().flatten()
        assert o.shape == gold[f"out_{i}"].shape
        if precision == "fp32":
            assert float(d.max()) <= 2e-4, float(d.max())
        else:
            assert float(d.mean()) <= 2e-3 and float(d.kthvalue(int(d.numel() * 0.999))[0]) <= 0.05, (float(d.mean()), float(d.max()))


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_engine_motion_path_on_emulator(precision):
    from gimmvfi_hip.engine import Engine
    from sim_runtime import SimRuntime

    meta, gold = load_golden("gimm_b2_96x160_t050")
    xs, ori, coord, ts = gimm_inputs(meta)
    eng = Engine(SimRuntime(precision), _sd(), motion_only=True)
    _check(_outs(eng.forward_motion(xs, coord, ori, ts)), gold, precision)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", CASES)
def test_gpu_model_matches_reference_golden(name, precision):
    from gimmvfi_hip.model import GIMM

    meta, gold = load_golden(name)
    xs, ori, coord, ts = gimm_inputs(meta)
    m = GIMM(precision=precision)
    m.load_state_dict(_sd(), strict=True)
    m = m.to("cuda").eval()
    cu = lambda t: [x.cuda() for x in t] if isinstance(t, list) else t.cuda()
    out = _outs(m(xs.cuda(), cu(coord), ori_flow=ori.cuda(), timesteps=cu(ts)))
    _check(out, gold, precision)
    loss = m.compute_loss(out[0], gold["out_0"].cuda(), reduction="mean")
    assert float(loss["psnr"]) > (60.0 if precision == "fp32" else 30.0)
