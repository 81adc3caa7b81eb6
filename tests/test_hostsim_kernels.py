"""CPU checks of the HIP kernel sources through the host emulator build (tests/hostsim): index math,
LDS addressing, barriers, epilogues and edge cases, before any GPU minute is spent.  The same cases
run against the real library in tests/test_gpu_kernels.py."""
import os

import pytest

import kernel_cases as kc
from gimmvfi_hip import lib as L
from sim_runtime import SimRuntime


@pytest.fixture(scope="module", params=["fp32", "bf16"])
def rt(request):
    return SimRuntime(request.param, emulate_conv=True)


CONV_SHAPES = [
    # N, H, W, Cin, Cout, KH, KW, kwargs
    (1, 8, 12, 3, 5, 3, 3, {}),
    (2, 9, 7, 20, 70, 3, 3, dict(stride=2, act1=L.ACT_RELU)),
    (1, 6, 10, 40, 130, 1, 5, dict(act1=L.ACT_TANH)),
    (1, 6, 10, 12, 33, 7, 7, dict(act1=L.ACT_PRELU, with_res=True)),
    (1, 10, 10, 16, 16, 3, 3, dict(reflect=True, with_res=True, act2=L.ACT_LRELU)),
    (1, 7, 9, 48, 24, 3, 3, dict(split=32, with_res=True, act2=L.ACT_PRELU, out_f32=True)),
    (1, 5, 6, 35, 2, 1, 1, dict(act1=L.ACT_SIN, out_f32=True, out_scale=0.25)),
    (1, 12, 12, 8, 64, 5, 5, dict(stride=1, act1=L.ACT_SIGMOID, tile=128)),
    # channel counts that are multiples of a K chunk -> LDS-DMA kernel (conv_igemm_glds.hip)
    (1, 6, 10, 64, 70, 3, 3, dict(act1=L.ACT_RELU, with_res=True)),
    (1, 5, 9, 128, 48, 1, 5, dict(split=64, act1=L.ACT_PRELU)),
    (2, 7, 7, 64, 130, 3, 3, dict(stride=2, out_f32=True)),
    (1, 9, 9, 64, 40, 3, 3, dict(reflect=True, with_res=True, act2=L.ACT_LRELU)),
    (1, 6, 10, 64, 70, 3, 3, dict(act1=L.ACT_RELU, with_res=True, algo=1)),   # same shape, generic kernel
    (1, 17, 19, 64, 200, 3, 3, dict(tile=256, act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU)),
    # (bf16_only: these paths only exist for bf16; the fp32 emulation of a 256x256 tile costs ~20 s each)
    (1, 17, 19, 64, 256, 3, 3, dict(tile=256, act1=L.ACT_PRELU, bf16_only=True)),   # 8-wave tile, bf16-staged activation epilogue
    (1, 9, 19, 64, 256, 1, 1, dict(tile=256, act1=L.ACT_LRELU, out_scale=0.5, bf16_only=True)),  # 8-wave 256x256 tile
    # the 8-wave tile with its DMA pieces spread over the MFMA groups (algo bit 5; the default issues 4 per group)
    (1, 17, 19, 64, 256, 3, 3, dict(tile=256, algo=2 + 32, act1=L.ACT_PRELU, bf16_only=True)),
    (1, 9, 11, 128, 24, 3, 3, dict(out_f32=True)),        # Cout <= 32 on the LDS-DMA kernel (128x32 tile)
    (2, 6, 7, 64, 2, 3, 3, dict(out_f32=True, with_res=True)),
    # selectable LDS-DMA variants: 64-row tiles, 64-byte chunks with the 4-deep ring (counted vmcnt), tall 256-row tiles
    (1, 9, 40, 64, 70, 3, 3, dict(tile=128 | (64 << 10), act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU)),
    (1, 8, 10, 128, 130, 1, 5, dict(tile=128 | (64 << 10), split=64, act1=L.ACT_RELU)),
    (1, 9, 12, 64, 130, 3, 3, dict(algo=2 + 128, tile=128)),
    # ragged last channel group with an even count on the slim store loop (dword stores): 126 = 15 x 8 + 6, with residual
    (1, 9, 12, 64, 126, 3, 3, dict(act1=L.ACT_RELU, coff=0, bf16_only=True)),
    (1, 9, 12, 64, 126, 3, 3, dict(algo=6, act1=L.ACT_PRELU, with_res=True, act2=L.ACT_LRELU, coff=8, bf16_only=True)),
    (1, 6, 10, 64, 68, 1, 1, dict(tile=128 | (64 << 10), out_scale=0.5, coff=0, bf16_only=True)),        # 4 valid channels
    (1, 6, 10, 64, 66, 1, 1, dict(algo=6, act1=L.ACT_GELU, coff=0, bf16_only=True)),                      # 2 valid channels
    (1, 6, 10, 64, 128, 1, 1, dict(algo=6, act1=L.ACT_RELU, with_res=True, coff=8, bf16_only=True)),      # aligned, full groups
    # float outputs on the slim store loop (decoder head 256->24, per-tap sums of the flow head 256->18)
    (1, 9, 11, 128, 24, 3, 3, dict(out_f32=True, coff=0, bf16_only=True)),
    (1, 6, 10, 64, 18, 1, 1, dict(out_f32=True, coff=0, act1=L.ACT_RELU, bf16_only=True)),
    (1, 6, 10, 128, 130, 1, 1, dict(algo=6, out_f32=True, coff=8, bf16_only=True)),
    # ... with a float residual (the transformer blocks' residual streams) and a 16-bit residual into a float output
    (1, 1, 200, 128, 128, 1, 1, dict(out_f32=True, with_res=True, res_f32=True, coff=0, bf16_only=True)),
    (1, 1, 200, 64, 70, 1, 1, dict(algo=6, act1=L.ACT_GELU, out_f32=True, with_res=True, res_f32=True, coff=8, bf16_only=True)),
    (1, 1, 200, 64, 64, 1, 1, dict(out_f32=True, with_res=True, coff=0, bf16_only=True)),
    # deeper rings at 128-byte chunks (tile_hint bits 20..23): 3 / 4 stages in flight, counted vmcnt, KT < / > ring depth
    (1, 8, 10, 128, 130, 1, 5, dict(tile=128 | (64 << 10) | (3 << 20), split=64, act1=L.ACT_RELU)),
    (1, 9, 12, 64, 130, 3, 3, dict(tile=128 | (64 << 10) | (4 << 20), act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU)),
    (1, 6, 20, 64, 128, 1, 1, dict(tile=128 | (64 << 10) | (4 << 20))),        # a single K chunk in a 4-deep ring
    (1, 9, 12, 128, 130, 3, 3, dict(tile=128 | (128 << 10) | (3 << 20), act1=L.ACT_LRELU)),
    (1, 9, 12, 64, 40, 3, 3, dict(tile=64 | (4 << 20), act1=L.ACT_LRELU)),
    (1, 9, 11, 128, 24, 3, 3, dict(tile=32 | (128 << 10) | (3 << 20), out_f32=True)),
    # weights-direct variant (w_layout 2: fragment-ordered weights loaded straight into MFMA operand registers, ring of 4):
    # K loops longer / shorter than the ring (steady state, every tail case, phantom chunks), two sources, ragged Cout
    # (column blocks beyond the image), ragged rows, both column tiles, residual + second activation, float output
    (1, 8, 10, 128, 130, 1, 5, dict(algo=6, split=64, act1=L.ACT_RELU, bf16_only=True)),                 # KT = 10
    (1, 9, 12, 64, 130, 3, 3, dict(algo=6, act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU, bf16_only=True)),   # KT = 9
    (1, 6, 20, 64, 128, 1, 1, dict(algo=6, bf16_only=True)),                                           # KT = 1
    (1, 6, 20, 128, 96, 1, 1, dict(algo=6, act1=L.ACT_GELU, bf16_only=True)),                          # KT = 2
    (1, 5, 9, 192, 40, 1, 1, dict(algo=6, split=64, out_f32=True, bf16_only=True)),                    # KT = 3
    (1, 7, 9, 64, 256, 2, 2, dict(algo=6, tile=256, act1=L.ACT_LRELU, pad=1, bf16_only=True)),         # KT = 4, 256 columns
    (1, 7, 9, 64, 200, 1, 5, dict(algo=6, tile=256, act1=L.ACT_RELU, bf16_only=True)),                 # KT = 5, ragged 256 tile
    (1, 7, 9, 128, 64, 3, 1, dict(algo=6, bf16_only=True)),                                            # KT = 6
    (1, 7, 9, 64, 32, 7, 1, dict(algo=6, bf16_only=True)),                                             # KT = 7
    (1, 7, 9, 64, 160, 3, 3, dict(algo=6, tile=256, with_res=True, bf16_only=True)),                   # KT = 9
    (2, 9, 12, 64, 130, 3, 3, dict(algo=6, tile=128 | (128 << 10), act1=L.ACT_RELU, with_res=True, bf16_only=True)),   # 128-row tiles, ragged
    (1, 8, 20, 128, 64, 1, 5, dict(algo=6, tile=128 | (128 << 10), split=64, out_f32=True, bf16_only=True)),
    (1, 9, 12, 96, 40, 3, 3, dict(act1=L.ACT_LRELU)),
    (1, 18, 20, 32, 32, 3, 3, dict(tile=32 | (256 << 10), act1=L.ACT_LRELU, with_res=True)),
    (1, 18, 20, 64, 24, 3, 3, dict(tile=32 | (256 << 10), out_f32=True)),
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
    # halo-staged 3x3 kernel (conv_p3x3.hip): ragged 16 x 16 tiles, two sources / two channel chunks, both epilogues, two Cout tiles
    (1, 17, 19, 64, 256, 3, 3, dict(algo=4, act1=L.ACT_PRELU, pad16=True, bf16_only=True)),
    (1, 17, 19, 128, 256, 3, 3, dict(algo=4, split=64, act1=L.ACT_PRELU, with_res=True, act2=L.ACT_PRELU, pad16=True, bf16_only=True)),
    (2, 9, 17, 64, 512, 3, 3, dict(algo=4, act1=L.ACT_LRELU, out_scale=0.5, pad16=True, bf16_only=True)),
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


def test_p3x3_conv_is_bit_identical_to_the_lds_dma_kernel(rt):
    if rt.precision != "bf16":
        pytest.skip("bf16-only kernel")
    kc.p3x3_equals_glds_case(rt, 1, 18, 17, 128, 256, split=64, with_res=True, act2=L.ACT_PRELU)
    kc.p3x3_equals_glds_case(rt, 1, 9, 33, 64, 256, act1=L.ACT_LRELU, out_scale=0.5, seed=1)
    kc.p3x3_equals_glds_case(rt, 1, 7, 40, 192, 256, split=128, seed=11)                          # odd chunk count, image lower than a tile
    kc.p3x3_equals_glds_case(rt, 1, 33, 5, 320, 256, split=256, act1=L.ACT_NONE, with_res=True, act2=L.ACT_PRELU, seed=13)   # image narrower than a tile
    kc.p3x3_equals_glds_case(rt, 2, 16, 16, 64, 512, with_res=True, act2=L.ACT_LRELU, seed=12)   # two Cout tiles, residual
    # launch form 1: the round-2 kernel (workgroup-wide staging tile); the calls above take the wave-private epilogue
    kc.p3x3_equals_glds_case(rt, 1, 18, 17, 128, 256, split=64, with_res=True, act2=L.ACT_PRELU, variant=1 << 13)
    kc.p3x3_equals_glds_case(rt, 1, 9, 33, 64, 256, act1=L.ACT_LRELU, out_scale=0.5, seed=1, variant=1 << 13)


def test_p3x3_stream_kernel_is_bit_identical_to_the_lds_dma_kernel(rt):
    """conv_p3x3.hip's persistent form: one workgroup per compute unit (the emulator launches 8) walks its tiles as ONE stream of
    channel chunks -- the next tile's first patch / weight stage arrive during the current tile's last chunk, wave-private
    epilogue in the LDS the last step read.  Several tiles per workgroup, one / odd / even chunk counts (the patch and weight
    parities differ from tile to tile), two sources, a workgroup's tiles crossing image rows and images, ragged borders, the
    residual by LDS-DMA and through registers, both activation paths (all slopes <= 1 / some beyond), and the library's own
    choice of the form (algo bits 13, 14 = 0) -- all under the emulator's adversarial LDS-DMA timing."""
    if rt.precision != "bf16":
        pytest.skip("bf16-only kernel")
    V = 3 << 13
    kc.p3x3_equals_glds_case(rt, 1, 40, 72, 64, 256, variant=V, seed=21)                                                   # one chunk per tile
    kc.p3x3_equals_glds_case(rt, 2, 33, 40, 192, 256, split=128, with_res=True, act2=L.ACT_PRELU, seed=22, variant=V)       # three chunks, two images
    kc.p3x3_equals_glds_case(rt, 1, 50, 50, 128, 256, act1=L.ACT_LRELU, with_res=True, act2=L.ACT_LRELU, seed=23, variant=V)
    kc.p3x3_equals_glds_case(rt, 1, 33, 65, 128, 256, with_res=True, act2=L.ACT_PRELU, seed=24, variant=V, slope_hi=2.0)    # slopes > 1
    kc.p3x3_equals_glds_case(rt, 1, 48, 80, 64, 256, act1=L.ACT_NONE, seed=25, slope_hi=2.0)                               # 15 tiles: the library picks the stream form


def test_p3x3s_conv_is_bit_identical_to_the_lds_dma_kernel(rt):
    """conv_p3x3s.hip (mid-channel sibling, algo 5): all four (Cin, Cout tile) instantiations, ragged tiles, Cout below
    the tile width, residual + second activation, output scale, output slice of a wider tensor."""
    if rt.precision != "bf16":
        pytest.skip("bf16-only kernel")
    kc.p3x3_equals_glds_case(rt, 1, 18, 17, 64, 64, algo_new=5)
    kc.p3x3_equals_glds_case(rt, 1, 17, 33, 64, 24, with_res=True, act2=L.ACT_PRELU, algo_new=5, seed=1)
    kc.p3x3_equals_glds_case(rt, 2, 9, 20, 32, 64, act1=L.ACT_LRELU, out_scale=0.5, algo_new=5, seed=2, ld_extra=24)
    kc.p3x3_equals_glds_case(rt, 1, 20, 18, 32, 32, with_res=True, act2=L.ACT_LRELU, algo_new=5, seed=3)
    kc.p3x3_equals_glds_case(rt, 3, 16, 3, 32, 8, algo_new=5, with_res=True, act2=L.ACT_PRELU, seed=15)   # narrower than a tile, one channel group
    kc.p3x3_equals_glds_case(rt, 1, 31, 31, 64, 40, algo_new=5, act1=L.ACT_NONE, out_scale=2.0, seed=16)


def test_gru_epilogues(rt):
    kc.gru_case(rt, kh=1, kw=5)
    kc.gru_case(rt, kh=5, kw=1, seed=1)
    if rt.precision == "bf16":   # (the slim GRU store loops are bf16-only; fp32 covers the kernel with the next case)
        kc.gru_case(rt, N=1, H=5, W=7, C=64, kh=1, kw=5, seed=2)   # 64-multiples -> LDS-DMA kernel and its GRU store loops
    kc.gru_case(rt, N=1, H=5, W=7, C=64, kh=5, kw=1, seed=3, ctx_split=True)   # hoisted context term (LDS-DMA kernel)
    kc.gru_case(rt, kh=1, kw=5, seed=4, ctx_split=True)                          # same on the generic kernel
    if rt.precision == "bf16":
        # float recurrent state (h, z) beside the bf16 operand copy: LDS-DMA kernel, its weights-direct variant, generic kernel
        kc.gru_case(rt, N=1, H=5, W=7, C=64, kh=1, kw=5, seed=5, state_f32=True, ctx_split=True)
        kc.gru_case(rt, N=1, H=5, W=7, C=64, kh=5, kw=1, seed=6, state_f32=True, ctx_split=True, wdir=True)
        kc.gru_case(rt, N=1, H=5, W=7, C=64, kh=1, kw=5, seed=7, state_f32=True, wdir=True)
        kc.gru_case(rt, kh=1, kw=5, seed=8, state_f32=True)


def test_conv_pair_launch_equals_two_launches(rt):
    if rt.precision != "bf16":
        kc.conv_pair_case(rt, shapes=((32, 40, 1, 1), (32, 24, 3, 3)), expect_pair=False)     # float: never a pair -> two launches
        return
    kc.conv_pair_case(rt)                                                                # 1x1 || 1x1: convc1 || convf1
    kc.conv_pair_case(rt, N=2, H=7, W=10, shapes=((128, 192, 3, 3), (64, 64, 3, 3)), seed=1)   # 3x3 || 3x3, ragged tiles: convc2 || convf2
    kc.conv_pair_case(rt, N=1, H=5, W=200, shapes=((64, 126, 1, 1), (64, 130, 3, 3)), seed=2)  # 16 + 32 workgroups: XCD order of both grids
    kc.conv_pair_case(rt, shapes=((32, 40, 1, 1), (64, 24, 3, 3)), seed=3, expect_pair=False)  # 32 channels: not the weights-direct variant


def test_corr_volume_grouped_gemm(rt):
    kc.corr_volume_case(rt)


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


def test_resize_warp_shuffle(rt):
    kc.resize_warp_shuffle_case(rt)


def test_corr_lookup(rt):
    kc.corr_lookup_case(rt)


def test_flow_step_equals_tap_sum_flow_pack_im2col(rt):
    kc.flow_step_case(rt)
    kc.flow_step_case(rt, N=1, h=8, w=16, first=True)
    kc.flow_step_case(rt, N=3, h=5, w=7)


def test_space_to_depth_form_of_the_filter_equals_stride_convolutions(rt):
    kc.s2d_case(rt, 2, 8, 12, 3, 64, 4)          # Twins patch embedding (twins.py:720-745): 4 taps of 4 x 8 values
    kc.s2d_case(rt, 1, 8, 16, 32, 32, 8, 1)      # sub-sampling convolution (twins.py:870-925; there 8 taps of 1024): 8 taps of 256


def test_convex_upsample(rt):
    kc.convex_upsample_case(rt)


def test_softsplat_edge_cases(rt):
    kc.splat_case(rt)


def test_compose_side_by_side_frames_like_the_reference_cli(rt):
    kc.compose_sbs_case(rt)
    kc.compose_sbs_case(rt, b=1, N=2, H0=8, W0=9, pad=(0, 0, 0, 0))


def test_col7_folded_finalisation_equals_finalize_image(rt):
    if rt.precision != "bf16":
        pytest.skip("the column kernel is bf16 only")
    kc.col7_planar_case(rt)
    kc.col7_planar_case(rt, N=1, H=70, W=71, seed=9)      # (an interior tile)


def test_softsplat_gather_is_deterministic_and_matches_the_oracle(rt):
    kc.splat_gather_case(rt)
    kc.splat_gather_case(rt, converge=True)
    kc.splat_gather_case(rt, converge=9)


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


def test_fp16_convolutions_and_gru():
    rt16 = SimRuntime("fp16", emulate_conv=True)
    for *a, kw in F16_CONVS:
        kc.conv_case(rt16, *a, **kw)
    kc.gru_case(rt16, kh=1, kw=5)                                                            # generic kernel
    kc.gru_case(rt16, N=1, H=5, W=7, C=64, kh=1, kw=5, seed=2)                                # LDS-DMA kernel, slim GRU loops
    kc.gru_case(rt16, N=1, H=5, W=7, C=64, kh=5, kw=1, seed=3, ctx_split=True, wdir=True)     # weights-direct + context term
    kc.tap_split_conv_case(rt16)
    kc.patch_conv_case(rt16)
    kc.corr_volume_case(rt16)


@pytest.mark.skipif(bool(os.environ.get("GVFI_EMU_SCHED")), reason="already inside the adversarial run")
def test_kernel_cases_under_adversarial_lds_dma_timing():
    """Every emulated kernel case again with GVFI_EMU_DMA=1 (tests/hostsim/hip_emu.h): an LDS-DMA poisons its destination at
    issue and delivers the data only at the `s_waitcnt vmcnt(N)` that covers it -- the latest and the earliest the hardware may
    land it, at once.  A missing or too lenient counted wait, a read in front of the barrier that publishes a chunk, a ring slot
    re-filled while a slower wave still reads it, a DMA landing in the epilogue's staging area: all read NaNs here, on every run,
    where the GPU would fail now and then.  Together with it the emulator's other wave schedules (GVFI_EMU_SCHED: reversed wave
    order, depth first, both): a kernel with all its barriers computes the same under every interleaving of its waves.
    Negative controls: with the counted waits disabled (DMA mode 2) the same cases must FAIL; tests/test_emulator_selftest.py."""
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    files = [os.path.join(root, "tests", f) for f in ("test_hostsim_kernels.py", "test_kernels_f.py", "test_alt_corr.py")]
    base = [sys.executable, "-m", "pytest", "-q", "-m", "not gpu", "-p", "no:cacheprovider"]
    for sched in ("3", "1", "2", "4"):          # reverse + depth first, reverse, depth first, random (seed 1)
        r = subprocess.run(base + files, capture_output=True, text=True, env=dict(os.environ, GVFI_EMU_DMA="1", GVFI_EMU_SCHED=sched), cwd=root)
        assert r.returncode == 0, (sched, r.stdout[-3000:])
        print(f"adversarial LDS-DMA timing, schedule {sched}:", r.stdout.strip().splitlines()[-1])
    r2 = subprocess.run(base + files[:1] + ["-k", "p3x3s_conv_is_bit_identical or conv_pair_launch"],
                        capture_output=True, text=True, env=dict(os.environ, GVFI_EMU_DMA="2"), cwd=root)
    assert r2.returncode != 0 and " failed" in r2.stdout, r2.stdout[-2000:]
    print("negative control (counted waits retire nothing):", r2.stdout.strip().splitlines()[-1])
