"""C ABI surface + state_dict contract (no compute on the product library without a GPU)."""
import json
import os

import pytest
import torch

from gimmvfi_hip import lib as L
from gimmvfi_hip.params import param_spec, random_state_dict
from util import GOLDEN, ROOT


def test_header_parses_and_declares_everything():
    protos = L.parse_header()
    assert len(protos) >= 30
    for need in ("gvfi_conv2d", "gvfi_softsplat_accum", "gvfi_corr_lookup", "gvfi_warp_nhwc", "gvfi_combine_warps"):
        assert need in protos


def test_product_library_exports_every_declared_symbol():
    # __graft_entry__.build() produces it; hipcc cross-compiles without a GPU
    if not os.path.isfile(L.LIB_PATH):
        import __graft_entry__ as ge

        ge.build()
    lib = L.HipLib(L.LIB_PATH)  # raises AttributeError on a missing symbol
    assert lib.version().startswith(b"gimmvfi-hip gfx950")
    for name in L.parse_header():
        assert hasattr(lib.dll, name)


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gimmvfi_hip.model import GIMMVFI_R

    m = GIMMVFI_R()
    x = torch.rand(1, 3, 2, 128, 128)
    c = [(m.sample_coord_input(1, (128, 128), [0.5], device="cpu"), None)]
    with pytest.raises(RuntimeError):
        m(x, c, t=[0.5 * torch.ones(1)])
    with pytest.raises(RuntimeError):
        L._LIB = None
        L.get()


def test_conv_params_struct_matches_c_layout(simlib):
    # the emulator build is compiled from the same header; a layout mismatch makes conv fail its checks
    import ctypes as C

    p = L.ConvParams()
    p.c0 = 3  # not a multiple of the vector -> must be rejected with -2
    assert simlib.conv2d(C.byref(p), None) == -2


def test_state_dict_contract():
    spec = param_spec()
    gold = json.load(open(os.path.join(GOLDEN, "state_dict_keys_r.json")))
    assert list(spec.keys()) == list(gold.keys())
    for k, v in spec.items():
        assert list(v) == gold[k], k
    assert sum(int(torch.tensor(v).prod()) if len(v) else 1 for v in spec.values()) == 19789980
    from gimmvfi_hip.model import GIMMVFI_R

    m = GIMMVFI_R()
    assert list(m.state_dict().keys()) == list(gold.keys())
    sd = random_state_dict(7)
    m.load_state_dict(sd, strict=True)
    bad = dict(sd)
    bad.pop("alpha_v")
    with pytest.raises(RuntimeError):
        m.load_state_dict(bad, strict=True)
