"""GIMM-VFI-F (FlowFormer flow estimator) on the HIP engine.

CPU (`-m "not gpu"`): the whole EngineF launch list in the host emulator (glue kernels = the real .hip sources,
contractions through the independent torch statement of sim_runtime) against the reference golden and, stage by
stage, the oracle -- batch 2, so the reference's context.repeat() tiling over (batch, latent token) is exercised.
GPU (`-m gpu`): libgimmvfi_hip.so through the drop-in model API against the same goldens.

Tolerances: fp32 mode PSNR >= 80 dB, flows within 5e-3 px (32 recurrent iterations of lookup -> GRU; measured on
MI355X: 139 dB, 3e-5 px); bf16 mode PSNR >= 42 dB and mean flow error < 0.25 px on flows of up to 30 px (measured:
43.7-49.0 dB, 0.15-0.21 px with the seeded random weights -- bf16 operands through two Twins encoders, 6 context-aware
blocks and 32 decoder iterations; flows / cost volume / coordinates / residual streams stay fp32)."""
import os

import pytest
import torch

import gimmvfi_f_oracle as forc
from util import golden_inputs, load_golden, maxabs, nchw, psnr

DEV = "cuda:0"


def _lat(t, B, P8):
    # image-major latent layout [(image, token k), pixel p, 128] -> the reference's (B*P8, 8, 128)
    return t.float().reshape(-1, 8, P8, 128)[:B].permute(0, 2, 1, 3).reshape(B * P8, 8, 128)


def _check_taps(taps, otaps, B, tol):
    def rel(a, b):
        return maxabs(a, b) / (float(b.abs().max()) + 1e-12)

    P8 = otaps["f01_cost_memory"].shape[0] // B
    assert rel(nchw(taps["f01_context"]), otaps["f01_context"]) < tol
    assert rel(nchw(taps["f01_cfeat4"]), otaps["f01_cfeat4"]) < tol
    assert rel(nchw(taps["f01_ffeat"]), otaps["f01_ffeat"]) < tol
    assert rel(taps["f01_cost_tokens"], otaps["f01_cost_tokens"]) < tol
    assert rel(_lat(taps["f01_latent_in"], B, P8), otaps["f01_latent_in"]) < tol
    assert rel(_lat(taps["f01_latent_l0"], B, P8), otaps["f01_latent_l0"]) < tol
    mem = taps["f01_cost_memory"].float().permute(0, 2, 1, 3).reshape(B * P8, 8, 128)
    assert rel(mem, otaps["f01_cost_memory"]) < tol
    assert rel(nchw(taps["f01_cost_fwd_it0"]), otaps["f01_cost_fwd_it0"]) < tol
    assert rel(nchw(taps["f01_cost_global_it0"]), otaps["f01_cost_global_it0"]) < tol
    assert rel(nchw(taps["f01_net_it0"]), otaps["f01_net_it0"]) < tol
    assert rel(nchw(taps["f01_net_it31"]), otaps["f01_net_it31"]) < 10 * tol


def test_engine_f_sim_fp32_matches_golden_and_oracle(sd_f):
    from gimmvfi_hip.engine_f import EngineF
    from sim_runtime import SimRuntime

    meta, gold = load_golden("f_b2_128x128_t025_075")
    x, coords, ts = golden_inputs(meta)
    eng = EngineF(SimRuntime("fp32"), sd_f)
    taps = {}
    out = eng.forward(x, coords, ts, iters=None, taps=taps)
    assert maxabs(out["raft_flow"], gold["raft_flow"]) < 2e-3
    for i in range(2):
        assert psnr(out["imgt_pred"][i], gold[f"imgt_pred_{i}"]) > 100.0
        assert maxabs(out["flowt"][i], gold[f"flowt_{i}"]) < 5e-3
        assert tuple(out["flowt"][i].shape) == tuple(gold[f"flowt_{i}"].shape)
    otaps = {}
    with torch.no_grad():
        forc.forward(sd_f, x, coords, ts, None, taps=otaps)
    _check_taps(taps, otaps, meta["B"], 1e-4)


def test_engine_f_sim_bf16(sd_f):
    """bf16 mode in the emulator: bf16 contraction operands, float residual streams / cost volume / coordinates."""
    from gimmvfi_hip.engine_f import EngineF
    from sim_runtime import SimRuntime

    meta, gold = load_golden("f_128x192_t050")
    x, coords, ts = golden_inputs(meta)
    out = EngineF(SimRuntime("bf16"), sd_f).forward(x, coords, ts, iters=None)
    assert psnr(out["imgt_pred"][0], gold["imgt_pred_0"]) > 40.0
    d = (out["flowt"][0].float() - gold["flowt_0"]).abs().flatten()
    assert float(d.mean()) < 0.5


def test_engine_f_sim_token_chains_equal_the_separate_launches(sd_f, monkeypatch):
    """The fused flow-token chains (gvfi_token_chain, default) against the 5 + 4 separate launches they replace
    (GVFI_F_TOKCHAIN=0), same emulated engine, three decoder iterations: the flows agree to the rounding of a few bf16
    operands (both paths round to the activation type at the same points; only the LayerNorm reductions differ in order)."""
    from gimmvfi_hip.engine_f import EngineF
    from sim_runtime import SimRuntime

    meta, _ = load_golden("f_128x192_t050")
    x, coords, ts = golden_inputs(meta)
    outs = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("GVFI_F_TOKCHAIN", sw)
        eng = EngineF(SimRuntime("bf16"), sd_f)
        assert (eng.chain_a is not None) == (sw == "1")
        outs[sw] = eng.forward(x, coords, ts, iters=3)
    d = (outs["1"]["raft_flow"].float() - outs["0"]["raft_flow"].float()).abs()
    assert float(d.max()) < 5e-2 and float(d.mean()) < 2e-3, (float(d.max()), float(d.mean()))
    assert psnr(outs["1"]["imgt_pred"][0], outs["0"]["imgt_pred"][0]) > 55.0


def test_engine_f_sim_flow_precision_policy(sd_f):
    """bf16 engine with the flow estimator's stages in float (GIMMVFI_F(flow_precision=...)): with all three stages in
    float the flows are those of the fp32 engine, the frames those of bf16 synthesis; a single float stage still runs the
    whole launch list (stage-boundary conversions)."""
    from gimmvfi_hip.engine_f import EngineF, parse_flow_policy
    from sim_runtime import SimRuntime

    assert parse_flow_policy("bf16") == {} and parse_flow_policy("fp32") == {s: "fp32" for s in ("enc", "cost", "tok", "upd")}
    assert parse_flow_policy("cost, dec") == {"cost": "fp32", "tok": "fp32", "upd": "fp32"}
    assert parse_flow_policy("f16") == {st: "fp16" for st in ("enc", "cost", "tok", "upd")}
    assert parse_flow_policy("dec:f16") == {"tok": "fp16", "upd": "fp16"} and parse_flow_policy("upd:f16,enc") == {"upd": "fp16", "enc": "fp32"}
    for bad in ("decoder", "dec:f8"):
        with pytest.raises(ValueError):
            parse_flow_policy(bad)
    meta, gold = load_golden("f_128x192_t050")
    x, coords, ts = golden_inputs(meta)
    mixed = EngineF(SimRuntime("bf16"), sd_f, flow_precision="fp32").forward(x, coords, ts, iters=None)
    assert maxabs(mixed["raft_flow"], gold["raft_flow"]) < 2e-3          # the float flow estimator
    p_mixed = psnr(mixed["imgt_pred"][0], gold["imgt_pred_0"])
    assert p_mixed > 50.0, p_mixed                                       # bf16 synthesis on exact flows
    # every stage-boundary conversion of the launch list: float encoder, the default (half decoder), a float token path feeding a
    # half update block
    for pol in ("enc", "dec:f16", "upd:f16,tok", "f16"):        # ("f16": the model default since round 5)
        part = EngineF(SimRuntime("bf16"), sd_f, flow_precision=pol).forward(x, coords, ts, iters=None)
        assert psnr(part["imgt_pred"][0], gold["imgt_pred_0"]) > 40.0, pol
    # an fp32 engine ignores the policy (everything is float already)
    assert EngineF(SimRuntime("fp32"), sd_f, flow_precision="fp32").side == {}


def test_engine_f_sim_ragged_grid(sd_f):
    """136 x 152 frames -> 17 x 19 grid at 1/8: ragged 7x7 windows (bias / positional-code keys), zero-extended
    sub-sampling convolutions and cost-map patches, odd P8 (padded pitch of the GMA attention matrix)."""
    from gimmvfi_hip.engine_f import EngineF
    from sim_runtime import SimRuntime

    meta, gold = load_golden("f_136x152_t040")
    x, coords, ts = golden_inputs(meta)
    out = EngineF(SimRuntime("fp32"), sd_f).forward(x, coords, ts, iters=None)
    assert maxabs(out["raft_flow"], gold["raft_flow"]) < 2e-3
    assert psnr(out["imgt_pred"][0], gold["imgt_pred_0"]) > 100.0


def test_create_model_f_contract(sd_f):
    """create_model('gimmvfi_f') returns the drop-in module: reference key set, strict load, no CPU fallback."""
    import json
    import os

    from src.models import create_model
    from util import GOLDEN

    m, ema = create_model({"type": "gimmvfi_f", "coord_range": [-1.0, 1.0]})
    assert ema is None
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys_f.json")))
    got = m.state_dict()
    assert set(got.keys()) == set(keys.keys())
    assert all(list(got[k].shape) == v for k, v in keys.items())
    m.load_state_dict(sd_f, strict=True)
    x = torch.rand(1, 3, 2, 128, 128)
    c = m.sample_coord_input(1, (128, 128), [0.5], device=x.device)
    with pytest.raises(RuntimeError):
        m(x, [(c, None)], t=[0.5 * torch.ones(1)])     # CPU tensors: the product path has no fallback


# ---------------------------------------------------------------------------------------------- GPU
def _model(sd, precision):
    from gimmvfi_hip.model import GIMMVFI_F

    m = GIMMVFI_F(precision=precision)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def _run(m, x, coords, ts, ds=None):
    out = m(x.to(DEV), [(c[0].to(DEV), None) for c in coords], t=[t.to(DEV) for t in ts], ds_factor=ds)
    torch.cuda.synchronize()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["f_128x192_t050", "f_b2_128x128_t025_075", "f_136x152_t040"])
def test_gpu_f_fp32_matches_reference_golden(name, sd_f):
    meta, gold = load_golden(name)
    x, coords, ts = golden_inputs(meta)
    out = _run(_model(sd_f, "fp32"), x, coords, ts, meta["ds"])
    assert maxabs(out["raft_flow"], gold["raft_flow"]) < 5e-3
    for i in range(len(meta["t"])):
        p = psnr(out["imgt_pred"][i], gold[f"imgt_pred_{i}"])
        assert p >= 80.0, p
        assert tuple(out["flowt"][i].shape) == tuple(gold[f"flowt_{i}"].shape)
        d = (out["flowt"][i].cpu() - gold[f"flowt_{i}"]).abs().flatten()
        assert float(d.kthvalue(int(d.numel() * 0.999))[0]) < 5e-3


@pytest.mark.gpu
def test_gpu_f_fp32_stage_taps_vs_oracle(sd_f):
    meta, _ = load_golden("f_b2_128x128_t025_075")
    x, coords, ts = golden_inputs(meta)
    otaps = {}
    with torch.no_grad():
        forc.forward(sd_f, x, coords, ts, None, taps=otaps)
    m = _model(sd_f, "fp32")
    taps = {}
    m.engine(DEV).forward(x.to(DEV), [(c[0].to(DEV), None) for c in coords], [t.to(DEV) for t in ts], iters=None,
                          taps=taps)
    torch.cuda.synchronize()
    _check_taps({k: v.cpu() for k, v in taps.items()}, otaps, meta["B"], 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["f_128x192_t050", "f_b2_128x128_t025_075", "f_136x152_t040"])
def test_gpu_f_bf16_matches_reference_golden(name, sd_f):
    meta, gold = load_golden(name)
    x, coords, ts = golden_inputs(meta)
    out = _run(_model(sd_f, "bf16"), x, coords, ts, meta["ds"])
    for i in range(len(meta["t"])):
        assert torch.isfinite(out["imgt_pred"][i]).all()
        p = psnr(out["imgt_pred"][i], gold[f"imgt_pred_{i}"])
        d = (out["flowt"][i].cpu().float() - gold[f"flowt_{i}"]).abs().flatten()
        print(f"\n[gimmvfi_f bf16 {name} t{i}] PSNR {p:.2f} dB, mean |flow err| {float(d.mean()):.3f} px")
        assert p >= 42.0, p
        assert float(d.mean()) < 0.25


@pytest.mark.gpu
def test_gpu_f_benchmark_size_bf16_vs_fp32(sd_f):
    """BASELINE.json configs[3] shape (448x256, batch 8): finite output, bf16 vs fp32-mode PSNR, graph replay equals
    the first (captured) call."""
    from gimmvfi_hip.synth import synthetic_pairs

    B, H, W = 8, 256, 448
    x = synthetic_pairs(B, H, W, seed=42)
    coords = [(forc.sample_coord_input(B, (H, W), [0.5], 1.0), None)]
    ts = [0.5 * torch.ones(B)]
    m16 = _model(sd_f, "bf16")
    o16 = _run(m16, x, coords, ts)
    o16b = _run(m16, x, coords, ts)
    assert torch.isfinite(o16["imgt_pred"][0]).all()
    assert psnr(o16b["imgt_pred"][0], o16["imgt_pred"][0]) >= 60.0
    # informational: graph-replay time of the whole forward (not a pass criterion)
    xd, cd, td = x.to(DEV), [(coords[0][0].to(DEV), None)], [ts[0].to(DEV)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        m16(xd, cd, t=td)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"\n[gimmvfi_f bf16 448x256 B=8] {ms:.1f} ms/step = {B / ms * 1e3:.1f} interpolated frames/s")
    del m16
    torch.cuda.empty_cache()
    o32 = _run(_model(sd_f, "fp32"), x, coords, ts)
    p = psnr(o16["imgt_pred"][0], o32["imgt_pred"][0])
    assert p >= 35.0, p


def test_whole_f_model_on_the_emulated_kernels(sd_f, monkeypatch):
    """GIMM-VFI-F in its default precision policy (bf16 synthesis, half-precision flow estimator) with every launch on the host
    build of the real kernels -- convolutions, the MFMA attentions, token chains: the GPU suite's golden test without a GPU."""
    from gimmvfi_hip.engine_f import EngineF
    from sim_runtime import SimRuntime

    monkeypatch.setenv("GVFI_ATTN_MFMA", "1")        # (the emulator build keeps the scalar attention kernels unless asked)
    meta, gold = load_golden("f_128x192_t050")
    x, coords, ts = golden_inputs(meta)
    rt = SimRuntime("bf16", emulate_conv=True)
    rt.lib.dll.gvfi_emu_set_dma_mode(1)       # (adversarial LDS-DMA timing, tests/hostsim/hip_emu.h)
    rt.lib.dll.gvfi_emu_set_sched(3)          # (... and the waves in reverse order, depth first)
    try:
        out = EngineF(rt, sd_f, flow_precision="f16").forward(x, coords, ts, iters=None)
    finally:
        rt.lib.dll.gvfi_emu_set_sched(0)
    p = psnr(out["imgt_pred"][0], gold["imgt_pred_0"])
    print(f"WHOLE-MODEL EMULATION gimmvfi_f bf16 / f16: PSNR(imgt_pred vs reference golden) = {p:.2f} dB, "
          f"max|raft_flow err| = {maxabs(out['raft_flow'], gold['raft_flow']):.3e}")
    assert p >= 50.0, p
    assert maxabs(out["raft_flow"], gold["raft_flow"]) < 0.1
