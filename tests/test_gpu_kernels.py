"""GPU parity tests of every HIP kernel through the C ABI of libgimmvfi_hip.so (real MI355X)."""
import pytest
import torch

import kernel_cases as kc
from gimmvfi_hip import lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fp32", "bf16"])
def rt(request):
    from gimmvfi_hip.ops import Runtime

    return Runtime(L.get(), request.param, "cuda:0")


CONV_SHAPES = [
    (1, 8, 12, 3, 5, 3, 3, {}),
    (2, 9, 7, 20, 70, 3, 3, dict(stride=2, act1=L.ACT_RELU)),
    (1, 6, 10, 40, 130, 1, 5, dict(act1=L.ACT_TANH)),
    (1, 6, 10, 12, 33, 7, 7, dict(act1=L.ACT_PRELU, with_res=True)),
    (1, 10, 10, 16, 16, 3, 3, dict(reflect=True, with_res=True, act2=L.ACT_LRELU)),
    (1, 7, 9, 48, 24, 3, 3, dict(split=32, with_res=True, act2=L.ACT_PRELU, out_f32=True)),
    (1, 5, 6, 35, 2, 1, 1, dict(act1=L.ACT_SIN, out_f32=True, out_scale=0.25)),
    (1, 12, 12, 8, 64, 5, 5, dict(act1=L.ACT_SIGMOID, tile=128)),
    # production shapes
    (2, 64, 112, 256, 256, 3, 3, dict(act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU)),   # final decoder ResBlock
    (2, 64, 112, 256, 256, 3, 3, dict(split=192)),                                          # two-source conv3/conv5
    (4, 32, 56, 324, 256, 1, 1, dict(act1=L.ACT_RELU)),                                       # RAFT convc1 (324 -> pad)
    (4, 32, 56, 384, 128, 5, 1, dict(split=128)),                                            # SepConvGRU geometry
    (2, 128, 224, 3, 64, 7, 7, dict(stride=2, act1=L.ACT_RELU)),                              # encoder stem
    (2, 64, 112, 273, 24, 3, 3, dict(out_f32=True)),                                         # 273-channel concat
    (1, 40, 40, 648, 256, 1, 1, dict(act1=L.ACT_LRELU)),
    (1, 33, 47, 18, 3, 7, 7, dict(out_f32=True, with_res=True)),                              # comb block, ragged size
    (1, 17, 19, 64, 200, 3, 3, dict(tile=256, act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU)),
    (1, 17, 19, 64, 256, 3, 3, dict(tile=256, act1=L.ACT_PRELU)),   # 8-wave tile, bf16-staged activation epilogue
    (1, 9, 19, 64, 256, 1, 1, dict(tile=256, act1=L.ACT_LRELU, out_scale=0.5)),  # 8-wave tile, ragged
    # the 8-wave tile with its DMA pieces spread over the MFMA groups (algo bit 5; the default issues 4 per group)
    (1, 17, 19, 64, 256, 3, 3, dict(tile=256, algo=2 + 32, act1=L.ACT_PRELU, bf16_only=True)),
    (4, 128, 224, 256, 256, 3, 3, dict(algo=2 + 32, split=192, bf16_only=True)),
    (4, 128, 224, 256, 256, 3, 3, dict(act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU)),   # auto -> 256x256 tile
    (4, 128, 224, 256, 256, 3, 3, dict(split=192, tile=128)),                                # same, 128-wide tile
    (2, 64, 112, 128, 128, 3, 3, dict(algo=1, act1=L.ACT_LRELU)),                             # generic kernel on an aligned shape
    (2, 64, 112, 256, 24, 3, 3, dict(out_f32=True)),                                         # decoder head, 128x32 LDS-DMA tile
    (4, 32, 56, 256, 2, 3, 3, dict(out_f32=True, with_res=True)),                             # RAFT flow head
    (2, 256, 448, 32, 32, 3, 3, dict(act1=L.ACT_LRELU, with_res=True)),                       # auto -> tall 256x32 tile (cnn encoder)
    (2, 64, 112, 64, 24, 3, 3, dict(tile=32 | (256 << 10), out_f32=True)),                    # tall 256x32 tile, 128-byte chunks
    (4, 32, 56, 384, 128, 1, 5, dict(split=128, act1=L.ACT_RELU)),                            # auto -> 64-row tiles (224 workgroups)
    (4, 32, 56, 384, 128, 1, 5, dict(split=128, act1=L.ACT_RELU, tile=128 | (64 << 10) | (3 << 20))),   # 3-deep ring
    (4, 32, 56, 256, 256, 3, 3, dict(act1=L.ACT_RELU, tile=128 | (64 << 10) | (4 << 20))),            # 4-deep ring
    (4, 32, 56, 128, 64, 3, 3, dict(act1=L.ACT_RELU, tile=64 | (4 << 20))),
    # weights-direct variant on the recurrence shapes; ragged 126-channel output on the slim store loop (aligned slice)
    (8, 32, 56, 256, 126, 3, 3, dict(algo=6, act1=L.ACT_RELU, coff=8, bf16_only=True)),
    (8, 32, 56, 256, 126, 3, 3, dict(act1=L.ACT_RELU, with_res=True, act2=L.ACT_LRELU, coff=0, tile=128 | (64 << 10), bf16_only=True)),
    (8, 32, 56, 256, 256, 1, 5, dict(algo=6, split=128, act1=L.ACT_RELU, coff=0, bf16_only=True)),
    (8, 32, 56, 384, 256, 1, 1, dict(algo=6, act1=L.ACT_RELU, bf16_only=True)),
    (2, 64, 112, 256, 24, 3, 3, dict(out_f32=True, coff=0, bf16_only=True)),          # float output on the slim store loop
    (8, 32, 56, 256, 18, 1, 1, dict(out_f32=True, coff=0, bf16_only=True)),
    (1, 1, 14336, 192, 128, 1, 1, dict(algo=6, split=64, act1=L.ACT_GELU, with_res=True, coff=0, bf16_only=True)),
    (1, 40, 56, 96, 96, 3, 3, dict(act1=L.ACT_RELU)),                                         # 64-byte chunks, 4-deep ring (counted vmcnt)
    # halo-staged 3x3 kernel (conv_p3x3.hip)
    (1, 17, 19, 64, 256, 3, 3, dict(algo=4, act1=L.ACT_PRELU, pad16=True, bf16_only=True)),
    (1, 17, 19, 128, 256, 3, 3, dict(algo=4, split=64, act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU, pad16=True, bf16_only=True)),
    (2, 9, 17, 64, 512, 3, 3, dict(algo=4, act1=L.ACT_LRELU, out_scale=0.5, pad16=True, bf16_only=True)),
    (4, 128, 224, 256, 256, 3, 3, dict(act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU, pad16=True, bf16_only=True)),   # auto -> p3x3
    # patch kernel (conv_patch.hip): few channels, halo patch + all weights in LDS; 8 x 64-pixel output blocks, ragged
    # tiles, 16-byte channel groups that are not a power of two per tap, odd step counts, stride 2, reflect padding
    (1, 10, 70, 9, 18, 7, 7, dict(algo=3, act1=L.ACT_PRELU)),
    (1, 9, 40, 18, 3, 7, 7, dict(algo=3, out_f32=True, with_res=True, bf16_only=True)),   # (f32: 178 KB of LDS)
    (2, 20, 70, 3, 64, 7, 7, dict(algo=3, stride=2, act1=L.ACT_RELU)),
    (1, 12, 66, 32, 16, 3, 3, dict(algo=3, reflect=True, with_res=True, act2=L.ACT_LRELU)),
    (1, 9, 33, 64, 32, 3, 3, dict(algo=3, reflect=True, bf16_only=True)),
    (1, 11, 65, 8, 32, 5, 5, dict(algo=3, act1=L.ACT_PRELU)),
    (1, 9, 64, 2, 16, 3, 3, dict(algo=3)),
    (1, 10, 70, 9, 18, 7, 7, dict(algo=3, act1=L.ACT_PRELU, pad16=True)),                     # whole 16-byte stores incl. pad channels
    (1, 9, 40, 18, 3, 7, 7, dict(algo=3, out_f32=True, with_res=True, pad16=True, bf16_only=True)),
    # column kernel of the 7x7 few-channel layers (conv_col7.hip, algo 7): 32 x 32 tiles (ragged, several tiles, two images),
    # every (channel blocks, 16-byte groups per pixel) instantiation the combination block uses and the generic ones,
    # PReLU / LeakyReLU / none, output scale, float output + float residual, pad channels written as zeros
    (1, 10, 70, 9, 18, 7, 7, dict(algo=7, act1=L.ACT_PRELU, pad16=True, bf16_only=True)),
    (1, 9, 40, 18, 3, 7, 7, dict(algo=7, out_f32=True, with_res=True, res_f32=True, pad16=True, bf16_only=True)),
    (2, 37, 45, 9, 18, 7, 7, dict(algo=7, act1=L.ACT_PRELU, pad16=True, bf16_only=True, seed=3)),
    (1, 33, 34, 18, 3, 7, 7, dict(algo=7, out_f32=True, with_res=True, res_f32=True, pad16=True, bf16_only=True, seed=4)),
    (1, 12, 40, 3, 16, 7, 7, dict(algo=7, act1=L.ACT_LRELU, out_scale=0.5, pad16=True, bf16_only=True)),
    (1, 100, 104, 9, 18, 7, 7, dict(algo=7, act1=L.ACT_PRELU, pad16=True, bf16_only=True, seed=5)),       # interior tiles: tile-independent patch offsets
    (2, 97, 70, 18, 3, 7, 7, dict(algo=7, out_f32=True, with_res=True, res_f32=True, pad16=True, bf16_only=True, seed=6)),
    (1, 8, 33, 8, 32, 7, 7, dict(algo=7, act1=L.ACT_RELU, pad16=True, bf16_only=True)),
    (1, 6, 36, 24, 12, 7, 7, dict(algo=7, pad16=True, bf16_only=True)),
    (1, 256, 448, 9, 18, 7, 7, dict(act1=L.ACT_PRELU)),                                       # auto -> patch kernel (comb block)
    (1, 256, 448, 18, 3, 7, 7, dict(out_f32=True, with_res=True)),
    (2, 256, 448, 3, 64, 7, 7, dict(stride=2, act1=L.ACT_RELU)),                              # encoder stem
    (1, 256, 448, 64, 32, 3, 3, dict(reflect=True, with_res=True)),                           # latent refiner output
    (2, 256, 448, 8, 32, 5, 5, dict(act1=L.ACT_PRELU)),
    # FlowFormer (GIMM-VFI-F): patch / sub-sampling convolutions with stride == kernel and no padding, the 6x6 stride-2
    # cost-map convolutions, GELU epilogues on both kernels, token-matrix linears ([1,1,rows,C])
    (2, 16, 24, 3, 128, 4, 4, dict(stride=4, pad=0)),
    (1, 8, 12, 128, 40, 2, 2, dict(stride=2, pad=0)),
    (1, 16, 16, 64, 24, 8, 8, dict(stride=8, pad=0)),
    (2, 8, 12, 96, 32, 4, 4, dict(stride=4, pad=0)),
    (3, 8, 12, 16, 32, 6, 6, dict(stride=2, pad=2, act1=L.ACT_RELU)),
    (1, 1, 200, 64, 72, 1, 1, dict(act1=L.ACT_GELU)),
    (1, 1, 300, 128, 256, 1, 1, dict(act1=L.ACT_GELU, tile=256, bf16_only=True)),     # GELU in the 8-wave tile's accumulator-layout epilogue
    (1, 1, 300, 128, 128, 1, 1, dict(act1=L.ACT_GELU, with_res=True, pad16=True, bf16_only=True)),   # ... and in the slim store loop
    (1, 1, 200, 24, 40, 1, 1, dict(act1=L.ACT_GELU, with_res=True, out_f32=True)),
    (1, 1, 8, 64, 64, 1, 1, {}),
]


@pytest.mark.parametrize("shape", CONV_SHAPES)
def test_conv(rt, shape):
    *a, kw = shape
    kw = dict(kw)
    if kw.pop("bf16_only", False) and rt.precision != "bf16":
        pytest.skip("bf16-only kernel path")
    kc.conv_case(rt, *a, **kw)
    torch.cuda.synchronize()


def test_p3x3_conv_is_bit_identical_to_the_lds_dma_kernel(rt):
    if rt.precision != "bf16":
        pytest.skip("bf16-only kernel")
    kc.p3x3_equals_glds_case(rt, 1, 18, 17, 128, 256, split=64, with_res=True, act2=L.ACT_PRELU)
    kc.p3x3_equals_glds_case(rt, 2, 250, 443, 256, 256, split=192, seed=1)                       # ragged tiles, conv3 / conv5 geometry
    kc.p3x3_equals_glds_case(rt, 4, 128, 224, 256, 256, with_res=True, act2=L.ACT_PRELU, seed=2)  # final ResBlock
    kc.p3x3_equals_glds_case(rt, 1, 136, 256, 320, 512, split=256, act1=L.ACT_LRELU, out_scale=0.5, seed=3)
    # launch form 1: the round-2 kernel (workgroup-wide staging tile)
    kc.p3x3_equals_glds_case(rt, 1, 18, 17, 128, 256, split=64, with_res=True, act2=L.ACT_PRELU, variant=1 << 13)
    kc.p3x3_equals_glds_case(rt, 4, 128, 224, 256, 256, with_res=True, act2=L.ACT_PRELU, seed=2, variant=1 << 13)
    torch.cuda.synchronize()


def test_p3x3_stream_kernel_is_bit_identical_to_the_lds_dma_kernel(rt):
    """conv_p3x3.hip's persistent form (one workgroup per CU, its tiles as one stream of channel chunks; see the emulated twin
    of this test): the library's own choice on the production shapes (>= 2 tiles per CU), the forced form on grids with 1-2
    tiles per workgroup, ragged borders, two sources, residual, both activation paths -- and five launches of the hot layer
    in a row, bit-equal (a race between the epilogue's staging and the next tile's DMA would show as a rare difference)."""
    if rt.precision != "bf16":
        pytest.skip("bf16-only kernel")
    V = 3 << 13
    kc.p3x3_equals_glds_case(rt, 2, 250, 443, 256, 256, split=192, seed=1)                                   # 896 ragged tiles: auto = stream
    kc.p3x3_equals_glds_case(rt, 8, 256, 448, 256, 256, with_res=True, act2=L.ACT_PRELU, seed=2)             # the hot layer, 14 tiles per CU
    kc.p3x3_equals_glds_case(rt, 4, 128, 224, 256, 256, with_res=True, act2=L.ACT_PRELU, seed=3, variant=V)  # 448 tiles: 1-2 per workgroup
    kc.p3x3_equals_glds_case(rt, 1, 136, 256, 320, 256, split=256, act1=L.ACT_LRELU, seed=4, variant=V)      # five chunks
    kc.p3x3_equals_glds_case(rt, 2, 544, 1024, 256, 256, act1=L.ACT_PRELU, seed=5, slope_hi=2.0)             # 2K / 4K grid, slopes > 1
    kc.p3x3_equals_glds_case(rt, 1, 200, 330, 64, 256, with_res=True, act2=L.ACT_LRELU, seed=6, variant=V)   # one chunk per tile
    kc.p3x3_repeat_case(rt, 8, 256, 448, 256, 256, reps=5)
    torch.cuda.synchronize()


def test_p3x3s_conv_is_bit_identical_to_the_lds_dma_kernel(rt):
    """conv_p3x3s.hip (mid-channel sibling, algo 5): all four (Cin, Cout tile) instantiations, ragged tiles, Cout below
    the tile width, residual + second activation, output scale, output slice of a wider tensor."""
    if rt.precision != "bf16":
        pytest.skip("bf16-only kernel")
    kc.p3x3_equals_glds_case(rt, 1, 18, 17, 64, 64, algo_new=5)
    kc.p3x3_equals_glds_case(rt, 1, 17, 33, 64, 24, with_res=True, act2=L.ACT_PRELU, algo_new=5, seed=1)
    kc.p3x3_equals_glds_case(rt, 2, 9, 20, 32, 64, act1=L.ACT_LRELU, out_scale=0.5, algo_new=5, seed=2, ld_extra=24)
    kc.p3x3_equals_glds_case(rt, 1, 20, 18, 32, 32, with_res=True, act2=L.ACT_LRELU, algo_new=5, seed=3)
    kc.p3x3_equals_glds_case(rt, 8, 256, 448, 64, 64, algo_new=5, seed=4)                                    # ResBlock side branch
    kc.p3x3_equals_glds_case(rt, 4, 250, 443, 32, 32, with_res=True, act2=L.ACT_LRELU, algo_new=5, seed=5)   # CNN encoder, ragged
    kc.p3x3_equals_glds_case(rt, 2, 256, 448, 32, 64, algo_new=5, seed=6)
    kc.p3x3_equals_glds_case(rt, 2, 256, 448, 64, 32, algo_new=5, seed=7)
    torch.cuda.synchronize()


def test_gru_epilogues(rt):
    kc.gru_case(rt, kh=1, kw=5)
    kc.gru_case(rt, kh=5, kw=1, seed=1)
    kc.gru_case(rt, N=1, H=5, W=7, C=64, kh=1, kw=5, seed=2)   # 64-multiples -> LDS-DMA kernel and its GRU store loops
    kc.gru_case(rt, N=1, H=5, W=7, C=64, kh=5, kw=1, seed=3, ctx_split=True)   # hoisted context term (LDS-DMA kernel)
    kc.gru_case(rt, kh=1, kw=5, seed=4, ctx_split=True)                          # same on the generic kernel
    kc.gru_case(rt, N=2, H=32, W=56, C=128, kh=1, kw=5, seed=2)
    if rt.precision == "bf16":
        # float recurrent state (h, z) beside the bf16 operand copy -- slim store loops of the LDS-DMA kernel, its
        # weights-direct variant (both column tiles come up: 2C = 256 / C = 128 outputs), and the generic kernel
        kc.gru_case(rt, N=2, H=32, W=56, C=128, kh=1, kw=5, seed=5, state_f32=True, ctx_split=True)
        kc.gru_case(rt, N=2, H=32, W=56, C=128, kh=5, kw=1, seed=6, state_f32=True, ctx_split=True, wdir=True)
        kc.gru_case(rt, N=1, H=5, W=7, C=64, kh=1, kw=5, seed=7, state_f32=True, wdir=True)
        kc.gru_case(rt, kh=1, kw=5, seed=8, state_f32=True)


def test_conv_pair_launch_equals_two_launches(rt):
    if rt.precision != "bf16":
        pytest.skip("the pair launch exists for the 16-bit weights-direct variant")
    kc.conv_pair_case(rt)
    kc.conv_pair_case(rt, N=8, H=32, W=56, shapes=((384, 256, 1, 1), (128, 128, 1, 1)), seed=1)      # convc1 || convf1 of a RAFT lane
    kc.conv_pair_case(rt, N=8, H=32, W=56, shapes=((256, 192, 3, 3), (128, 64, 3, 3)), seed=2)       # convc2 || convf2
    kc.conv_pair_case(rt, N=1, H=68, W=128, shapes=((256, 192, 3, 3), (128, 64, 3, 3)), seed=3)      # one 2K / 4K lane
    kc.conv_pair_case(rt, shapes=((32, 40, 1, 1), (64, 24, 3, 3)), seed=4, expect_pair=False)


def test_corr_volume_grouped_gemm(rt):
    kc.corr_volume_case(rt)
    kc.corr_volume_case(rt, B=2, h=32, w=56, C=256, seed=3)


def test_conv_fused_instnorm_stats(rt):
    if rt.precision != "bf16":
        pytest.skip("fused statistics live in the bf16 store loop; fp32 runs gvfi_instnorm_stats")
    kc.conv_stats_case(rt)
    kc.conv_stats_case(rt, N=1, H=8, W=16, Cin=64, Cout=96)
    kc.conv_stats_case(rt, N=2, H=17, W=19, Cin=64, Cout=64, algo=5)      # mid-channel halo-staged kernel, ragged tiles
    kc.conv_stats_case(rt, N=1, H=16, W=33, Cin=32, Cout=24, algo=5)


def test_tap_split_conv(rt):
    kc.tap_split_conv_case(rt)


def test_patch_conv(rt):
    kc.patch_conv_case(rt)
    kc.patch_conv_case(rt, N=1, H=9, W=8, Cin=4, Cout=16)


def test_inr_mlp(rt):
    if rt.precision != "bf16":
        pytest.skip("fused hypo-network is the bf16 path; fp32 runs layer by layer")
    kc.inr_mlp_case(rt)
    kc.inr_mlp_case(rt, B=2, H=16, W=40)


def test_instnorm(rt):
    kc.instnorm_case(rt)
    kc.instnorm_case(rt, N=2, H=64, W=112, C=96)


def test_resize_warp_shuffle(rt):
    kc.resize_warp_shuffle_case(rt)


def test_corr_lookup(rt):
    kc.corr_lookup_case(rt)


def test_flow_step_equals_tap_sum_flow_pack_im2col(rt):
    kc.flow_step_case(rt)
    kc.flow_step_case(rt, N=1, h=8, w=16, first=True)
    kc.flow_step_case(rt, N=3, h=5, w=7)
    kc.flow_step_case(rt, N=8, h=32, w=56)


def test_space_to_depth_form_of_the_filter_equals_stride_convolutions(rt):
    kc.s2d_case(rt, 2, 16, 24, 3, 128, 4)        # Twins patch embedding (twins.py:720-745): 4 taps of 4 x 8 values
    kc.s2d_case(rt, 1, 16, 24, 128, 128, 8, 1)   # sub-sampling convolution (twins.py:870-925): 8 taps of 1024
    kc.s2d_case(rt, 2, 8, 12, 64, 40, 4, 2)


def test_convex_upsample(rt):
    kc.convex_upsample_case(rt)


def test_softsplat_edge_cases(rt):
    kc.splat_case(rt)
    kc.splat_case(rt, B=2, H=64, W=96)


def test_compose_side_by_side_frames_like_the_reference_cli(rt):
    kc.compose_sbs_case(rt)
    kc.compose_sbs_case(rt, b=1, N=2, H0=8, W0=9, pad=(0, 0, 0, 0))


def test_col7_folded_finalisation_equals_finalize_image(rt):
    if rt.precision != "bf16":
        pytest.skip("the column kernel is bf16 only")
    kc.col7_planar_case(rt)
    kc.col7_planar_case(rt, N=1, H=70, W=71, seed=9)


def test_softsplat_gather_is_deterministic_and_matches_the_oracle(rt):
    kc.splat_gather_case(rt)
    kc.splat_gather_case(rt, converge=True)
    kc.splat_gather_case(rt, converge=9)
    kc.splat_gather_case(rt, B=2, H=64, W=96)


def test_combine_warps_up_equals_separate_passes(rt):
    for scale in (1, 2, 4):
        kc.combine_warps_up_case(rt, scale=scale)
    # full-resolution frames made of whole 64 x 4 tiles: the staged form (decoder taps through LDS, one tile row per wave)
    kc.combine_warps_up_case(rt, B=2, H=6, W=64, scale=4)
    kc.combine_warps_up_case(rt, B=1, H=6, W=128, scale=2)
    kc.combine_warps_up_case(rt, B=1, H=7, W=48, scale=4)


def test_softsplat_native_op_contract(rt):
    kc.splat_nchw_case(rt)


def test_splat_weights_and_flow_norm(rt, sd):
    kc.splat_weights_and_norm_case(rt, sd)


def test_flow_to_image_matches_the_cli_colour_coding(rt):
    if rt.precision != "fp32":
        pytest.skip("type-independent kernel: once is enough")
    kc.flow_to_image_case(rt)


F16_CONVS = [
    # IEEE-half operands (GVFI_F16: the decoder of GIMM-VFI-F's flow estimator): LDS-DMA 4-wave tiles, weights-direct
    # variant (both column tiles, every tail case of its ring), generic kernel, two sources, float output, residual
    (1, 9, 12, 64, 130, 3, 3, dict(act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU)),
    (1, 8, 10, 128, 130, 1, 5, dict(tile=128 | (64 << 10), split=64, act1=L.ACT_RELU)),
    (1, 9, 11, 128, 24, 3, 3, dict(out_f32=True)),
    (1, 9, 12, 64, 40, 3, 3, dict(act1=L.ACT_LRELU)),
    (1, 1, 300, 128, 256, 1, 1, dict(act1=L.ACT_GELU, tile=256)),            # 8-wave 256 x 256 tile on half operands (round 5: "enc:f16")
    (1, 9, 19, 64, 200, 3, 3, dict(tile=256, act1=L.ACT_PRELU, with_res=True, res_f32=True, coff=8)),
    (1, 8, 10, 128, 130, 1, 5, dict(algo=6, split=64, act1=L.ACT_RELU)),
    (1, 6, 20, 128, 96, 1, 1, dict(algo=6, act1=L.ACT_GELU)),
    (1, 7, 9, 64, 200, 1, 5, dict(algo=6, tile=256, act1=L.ACT_RELU)),
    (1, 7, 9, 64, 160, 3, 3, dict(algo=6, tile=256, with_res=True)),
    (1, 6, 10, 40, 130, 1, 5, dict(act1=L.ACT_TANH)),                 # generic kernel (40 channels)
    (1, 1, 200, 24, 40, 1, 1, dict(act1=L.ACT_GELU, with_res=True, out_f32=True)),
]


@pytest.mark.gpu
def test_fp16_convolutions_and_gru():
    from gimmvfi_hip.ops import Runtime

    rt16 = Runtime(L.get(), "fp16", "cuda:0")
    for *a, kw in F16_CONVS:
        kc.conv_case(rt16, *a, **kw)
    kc.gru_case(rt16, kh=1, kw=5)                                                            # generic kernel
    kc.gru_case(rt16, N=1, H=5, W=7, C=64, kh=1, kw=5, seed=2)                                # LDS-DMA kernel, slim GRU loops
    kc.gru_case(rt16, N=1, H=5, W=7, C=64, kh=5, kw=1, seed=3, ctx_split=True, wdir=True)     # weights-direct + context term
    kc.tap_split_conv_case(rt16)
    kc.patch_conv_case(rt16)
    kc.corr_volume_case(rt16)
    kc.gru_case(rt16, N=2, H=32, W=56, C=128, kh=1, kw=5, seed=6, ctx_split=True, wdir=True)
    kc.conv_case(rt16, 4, 32, 56, 256, 256, 3, 3, algo=6, act1=L.ACT_RELU)
    torch.cuda.synchronize()
