"""the product's ``sample_coord_input`` (gimmvfi_hip/model.py -- what the CLI, the evaluators and
bench.py call) against the oracle restatement and, in the dev container, against the reference's own
``GIMMVFI_R.sample_coord_input`` / ``CoordSampler3D`` (gimmvfi_r.py:428-442, modules/coord_sampler.py:15-91), for the
up-sampling ratios the reference's settings use (DS_SCALE 1 / 0.5 / 0.25, README.md:87-96)."""
import pytest
import torch

import gimmvfi_r_oracle as orc
import ref_harness as rh

SHAPES = [(256, 448), (736, 864), (1088, 2048), (150, 200)]
RATIOS = [1.0, 0.5, 0.25]


def _models():
    from gimmvfi_hip.model import GIMM, GIMMVFI_F, GIMMVFI_R

    return [GIMMVFI_R(), GIMMVFI_F(), GIMM()]


@pytest.mark.parametrize("ratio", RATIOS)
@pytest.mark.parametrize("shape", SHAPES)
def test_product_sampler_equals_oracle(shape, ratio):
    for m in _models():
        for B, t in ((1, 0.5), (3, 0.125)):
            got = m.sample_coord_input(B, shape, [t], device="cpu", upsample_ratio=ratio)
            want = orc.sample_coord_input(B, shape, [t], ratio)
            assert got.shape == want.shape == (B, 1, int(shape[0] * ratio), int(shape[1] * ratio), 3)
            assert got.dtype == torch.float32
            assert float((got - want).abs().max()) == 0.0
            assert float(got[0, 0, 0, 0, 0]) == pytest.approx(t)       # forward() asserts this (gimmvfi_r.py:351)


@pytest.mark.skipif(not rh.reference_available(), reason="reference checkout only exists in the dev container")
@pytest.mark.parametrize("ratio", RATIOS)
def test_product_sampler_equals_reference_live(ratio):
    ref = rh.build_reference_model()
    from gimmvfi_hip.model import GIMMVFI_R

    m = GIMMVFI_R()
    for shape in SHAPES:
        for B, t in ((1, 0.5), (2, 0.875)):
            want = ref.sample_coord_input(B, shape, [t], device=torch.device("cpu"), upsample_ratio=ratio)
            got = m.sample_coord_input(B, shape, [t], device="cpu", upsample_ratio=ratio)
            assert got.shape == want.shape
            assert float((got - want).abs().max()) == 0.0


def test_multiple_time_ids():
    from gimmvfi_hip.model import GIMMVFI_R

    m = GIMMVFI_R()
    got = m.sample_coord_input(2, (64, 96), [0.25, 0.75], device="cpu")
    want = orc.sample_coord_input(2, (64, 96), [0.25, 0.75], 1.0)
    assert got.shape == (2, 2, 64, 96, 3) and float((got - want).abs().max()) == 0.0
