"""FlowFormer glue kernels (csrc/flowformer_ops.hip): CPU emulator build of the real sources (`-m "not gpu"`) and the
gfx950 library on the GPU (`-m gpu`), same cases (tests/kernel_cases_f.py)."""
import os

import pytest

import kernel_cases_f as kf
from gimmvfi_hip import lib as L


@pytest.fixture(scope="module", params=["fp32", "bf16", "fp16"])
def rt_sim(request):
    from sim_runtime import SimRuntime

    return SimRuntime(request.param, emulate_conv=True)


@pytest.fixture(scope="module", params=["fp32", "bf16", "fp16"])
def rt_gpu(request):
    from gimmvfi_hip.ops import Runtime

    return Runtime(L.get(), request.param, "cuda:0")


def _all(rt):
    kf.layernorm_case(rt)
    kf.layernorm_case(rt, rows=300, C=256, x_f32=True, eps=1e-5)     # > emulator's vector limit: scalar path on CPU
    kf.layernorm_case(rt, rows=5, C=64)
    kf.layernorm_case(rt, rows=70, C=256, x_f32=True, eps=1e-5)      # vector kernel, float stream, 32 / 64 lanes per row
    kf.layernorm_case(rt, rows=100, C=128, x_f32=True)
    kf.layernorm_case(rt, rows=9, C=20)                               # not a power-of-two lane count: scalar kernel
    if rt.on_gpu:                                                     # >= 4096 rows: four rows per lane group, ragged tail
        kf.layernorm_case(rt, rows=4096 + 37, C=128, x_f32=True)
        kf.layernorm_case(rt, rows=5000, C=256, x_f32=False, eps=1e-5)
    kf.dwconv_case(rt)
    kf.dwconv_case(rt, f32=True)
    kf.pos_embed_case(rt)
    kf.cost_embed_lookup_case(rt)
    kf.cost_embed_lookup_case(rt, maps=5, h=8, w=12)                   # whole float4 rows: the 16-byte copy of the LDS-staged form
    kf.attn_window_case(rt)
    kf.attn_window_case(rt, B=1, H=7, W=14, C=128, heads=4)          # head_dim 32, no padding
    kf.attn_window_case(rt, B=1, H=15, W=8, C=128, heads=8)          # head_dim 16, ragged in both directions
    kf.attn_global_case(rt)
    kf.attn_global_mfma_case(rt, hd=16)                               # bf16: MFMA attention (attn_mfma.hip); fp32: scalar kernel
    kf.attn_global_mfma_case(rt, hd=32)
    kf.xqk_case(rt)
    kf.tile_softmax_case(rt)
    kf.token_chain_case(rt)
    kf.token_chain_case(rt, rows=32)


def test_flowformer_kernels_emulated(rt_sim, monkeypatch):
    monkeypatch.setenv("GVFI_ATTN_MFMA", "1")     # the emulator build takes the MFMA attention kernels only on request
    _all(rt_sim)


@pytest.mark.gpu
def test_flowformer_kernels_gpu(rt_gpu):
    import torch

    _all(rt_gpu)
    torch.cuda.synchronize()
