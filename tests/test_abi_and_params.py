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


def _conv_params(N, H, W, c0, Cout, k=(3, 3), c1=0, w_layout=0, stride=1, pad_mode=L.PAD_ZEROS, res=False, res_f32=False,
                 y_f32=False, dtype=L.BF16, ld0=None, stats=False):
    """A parameter block with fake (aligned, never dereferenced) pointers: gvfi_conv2d_plan is pure host logic."""
    p = L.ConvParams()
    p.dtype = dtype
    p.x0, p.ld0, p.c0 = 0x10000, c0 if ld0 is None else ld0, c0
    p.x1, p.ld1, p.c1 = (0x20000, c1, c1) if c1 else (None, 0, 0)
    p.N, p.H, p.W = N, H, W
    p.w, p.w_group_stride, p.groups, p.bias = 0x30000, 0, 1, 0x40000
    p.Cout, p.KH, p.KW, p.stride = Cout, k[0], k[1], stride
    p.pad_h, p.pad_w, p.pad_mode = k[0] // 2, k[1] // 2, pad_mode
    p.Ho, p.Wo = (H + 2 * (k[0] // 2) - k[0]) // stride + 1, (W + 2 * (k[1] // 2) - k[1]) // stride + 1
    p.epi_mode, p.act1, p.act2, p.out_scale = L.EPI_STD, L.ACT_RELU, L.ACT_NONE, 1.0
    if res:
        p.res, p.ldr, p.res_f32 = 0x50000, Cout, int(res_f32)
    p.y, p.ldy, p.y_f32 = 0x60000, (Cout + 7) // 8 * 8, int(y_f32)
    p.stats = 0x70000 if stats else None
    p.tile_hint, p.w_layout, p.algo = 0, w_layout, 0
    return p


def test_convolution_routing_table_of_the_product_library():
    """Which kernel gvfi_conv2d picks for the path's representative layers (host logic of the real gfx950 library, no
    launch): a silent change of this table is a performance regression no parity test would see."""
    lib = L.HipLib(L.LIB_PATH)
    import ctypes as C

    def plan(p):
        out = (C.c_int * 5)()
        assert lib.conv2d_plan(C.byref(p), out) == 0
        return list(out)

    # decoder ResBlocks (fi_components.py:97-154): chunked weight image -> halo-staged 3x3 kernel, also with two sources / residual
    assert plan(_conv_params(8, 256, 448, 256, 256, w_layout=1))[0] == 4
    assert plan(_conv_params(1, 544, 1024, 192, 256, c1=64, w_layout=1, res=True))[0] == 4
    assert plan(_conv_params(8, 64, 112, 256, 256, w_layout=1))[:3] == [2, 128, 128]          # < 65536 pixels: LDS-DMA kernel
    assert plan(_conv_params(8, 256, 448, 256, 192, w_layout=1))[:3] == [2, 256, 256]         # Cout not a multiple of 256
    # mid-channel full-resolution layers: plain weight image -> conv_p3x3s (also with fused statistics / bf16 residual)
    assert plan(_conv_params(8, 256, 448, 64, 64))[:3] == [5, 256, 64]
    assert plan(_conv_params(16, 256, 448, 32, 32, res=True))[:3] == [5, 256, 32]
    assert plan(_conv_params(16, 128, 224, 64, 64, stats=True))[0] == 5
    assert lib.conv2d_stats_ok(C.byref(_conv_params(16, 128, 224, 64, 64, stats=True))) == 1
    assert plan(_conv_params(8, 256, 448, 64, 64, res=True, res_f32=True))[0] == 2             # float residual: LDS-DMA kernel
    assert plan(_conv_params(8, 256, 448, 64, 64, ld0=256))[0] == 5                            # a channel slice of a wider tensor
    assert plan(_conv_params(8, 256, 448, 96, 96))[0] == 2
    # RAFT update block at 1/8 resolution: 64-row LDS-DMA tiles on grids that would otherwise under-fill the chip
    assert plan(_conv_params(8, 32, 56, 256, 256, k=(1, 5), w_layout=1))[:3] == [2, 64, 128]
    assert plan(_conv_params(16, 32, 56, 256, 256, k=(1, 5), w_layout=1))[:3] == [2, 128, 128]
    # few-channel layers at full resolution and reflect padding: patch kernel; everything else: generic kernel
    assert plan(_conv_params(1, 2176, 4096, 16, 18, k=(7, 7)))[0] == 3                          # (no pad16 contract: patch kernel)
    # combination block (gimmvfi_r.py:60-64): with the pad16 store contract -> column kernel (conv_col7.hip), both layers
    def comb(N, H, W, c0, Cout, **kw):
        q = _conv_params(N, H, W, c0, Cout, k=(7, 7), **kw)
        q.algo = 16
        return q
    assert plan(comb(1, 2176, 4096, 16, 18))[:3] == [7, 1024, 32]
    q = comb(1, 2176, 4096, 24, 3, res=True, res_f32=True, y_f32=True)
    q.act1, q.ldy, q.ldr = L.ACT_NONE, 4, 4
    assert plan(q)[:3] == [7, 1024, 16]
    assert plan(comb(8, 256, 448, 16, 18))[0] == 7
    assert plan(comb(1, 160, 200, 16, 18))[0] == 3                                             # < 65536 pixels: patch kernel
    assert plan(comb(1, 2176, 4096, 16, 18, dtype=L.F32))[0] in (1, 3)                          # float validation mode
    q = comb(1, 2176, 4096, 32, 32)
    assert plan(q)[0] != 7                                                                     # patch + weight fragments + staging > 160 KB
    assert plan(_conv_params(1, 544, 1024, 64, 32, pad_mode=L.PAD_REFLECT))[0] == 3
    assert plan(_conv_params(8, 256, 448, 64, 64, dtype=L.F32))[0] in (1, 2)                    # float validation mode never takes 4 / 5
