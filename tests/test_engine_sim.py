"""End-to-end check of the host pipeline (gimmvfi_hip/engine.py) on the CPU: every glue kernel runs
in the emulator build, convolutions through an independent torch statement of the launch arguments
(tests/hostsim/sim_runtime.py).  Catches layout / channel-order / folding mistakes without a GPU."""
import pytest
import torch

import gimmvfi_r_oracle as orc
from gimmvfi_hip.engine import Engine
from sim_runtime import SimRuntime
from util import golden_inputs, load_golden, maxabs, nchw, psnr


@pytest.mark.parametrize("name", ["r_128x192_t050", "r_256x256_ds050_t050"])
def test_engine_fp32_matches_golden_and_oracle(name, sd):
    meta, gold = load_golden(name)
    x, coords, ts = golden_inputs(meta)
    eng = Engine(SimRuntime("fp32"), sd)
    taps = {}
    out = eng.forward(x, coords, ts, ds_factor=meta["ds"], taps=taps)
    assert psnr(out["imgt_pred"][0], gold["imgt_pred_0"]) > 100.0
    assert maxabs(out["raft_flow"], gold["raft_flow"]) < 1e-3
    assert maxabs(out["flowt"][0], gold["flowt_0"]) < 1e-3
    assert maxabs(out["flowt0_pred"][0][1], gold["flowt0_4_0"]) < 1e-3
    assert tuple(out["flowt"][0].shape) == tuple(gold["flowt_0"].shape)
    otaps = {}
    with torch.no_grad():
        orc.forward(sd, x, coords, ts, meta["ds"], taps=otaps)
    for k in ("pl0", "feat0_4", "feat0_8"):
        assert maxabs(nchw(taps[k]), otaps[k]) < 1e-4, k
    assert maxabs(nchw(taps["t0_latent"]), otaps["t0_latent"]) < 1e-4
    assert maxabs(nchw(taps["t0_upd_ft_4"]), otaps["t0_upd_ft_4"]) < 1e-4


def test_engine_bf16_batch2_two_timesteps(sd):
    meta, gold = load_golden("r_b2_128x128_t025_075")
    x, coords, ts = golden_inputs(meta)
    out = Engine(SimRuntime("bf16"), sd).forward(x, coords, ts)
    for i in range(2):
        assert psnr(out["imgt_pred"][i], gold[f"imgt_pred_{i}"]) > 45.0   # bf16 storage, fp32 accumulation
        assert tuple(out["flowt"][i].shape) == tuple(gold[f"flowt_{i}"].shape)
        # splat holes (0/0 -> 1, softsplat.py:333-334) are discontinuous: judge flow by mean / p99, not max
        d = (out["flowt"][i].float() - gold[f"flowt_{i}"]).abs().flatten()
        assert float(d.mean()) < 0.03 and float(d.kthvalue(int(d.numel() * 0.99))[0]) < 0.1


def test_sequence_mode_shares_encoder_work_and_changes_nothing(sd):
    """Engine.forward(seq=True): B consecutive pairs of one frame sequence -- the per-frame encoders run on the B+1
    distinct frames; the flow estimator's outputs equal the pair-by-pair batch bit for bit, everything
    behind the softmax splat up to the order of its float atomics."""
    import torch

    from gimmvfi_hip.engine import Engine
    from gimmvfi_hip.synth import synthetic_pairs
    from sim_runtime import SimRuntime

    p = synthetic_pairs(2, 128, 128, seed=31)
    frames = torch.stack([p[0, :, 0], p[0, :, 1], p[1, :, 1]], 0)          # 3 consecutive frames
    x = torch.stack([frames[:-1], frames[1:]], dim=2)                       # (2,3,2,H,W): pairs (0,1), (1,2)
    coords = [(orc.sample_coord_input(2, (128, 128), [t], 1.0), None) for t in (0.25, 0.5)]
    ts = [t * torch.ones(2) for t in (0.25, 0.5)]
    eng = Engine(SimRuntime("fp32"), sd)
    calls = []
    enc = eng._enc
    eng._enc = lambda x_, *a, **k: (calls.append(x_.shape[0]), enc(x_, *a, **k))[1]
    a = eng.forward(x, coords, ts)
    assert calls == [4, 4]                                                  # fnet, cnet on 2B images
    calls.clear()
    b = eng.forward(x, coords, ts, seq=True)
    assert calls == [3, 3]                                                  # ... on the 3 distinct frames
    for k in ("raft_flow", "nflow"):
        assert torch.equal(a[k], b[k])
    for i in range(2):
        assert maxabs(a["imgt_pred"][i], b["imgt_pred"][i]) < 1e-5
        assert maxabs(a["flowt"][i], b["flowt"][i]) < 1e-4


def test_timestep_batched_synthesis_equals_one_by_one(sd):
    """Engine.forward runs frame synthesis of several timesteps as ONE batch [t][b] whose t-independent sources (images,
    context features, correlation pyramids) are read modulo the pair batch by the warp / copy / look-up / combine kernels
    (gvfi_*'s src_N): same frames, flows and stage taps as the reference's one-timestep-at-a-time loop
    (gimmvfi_r.py:376-396), here with T = 3, B = 2 and a group bound that splits the timesteps 2 + 1."""
    from gimmvfi_hip.synth import synthetic_pairs

    x = synthetic_pairs(2, 128, 128, seed=41)
    tl = (0.25, 0.5, 0.875)
    coords = [(orc.sample_coord_input(2, (128, 128), [t], 1.0), None) for t in tl]
    ts = [t * torch.ones(2) for t in tl]
    eng = Engine(SimRuntime("fp32"), sd)
    outs, taps = [], []
    for pix in (0, 2 * 2 * 128 * 128, 10 ** 9):            # one by one | groups of 2 + 1 | all three together
        eng.t_batch_pix = pix
        tp = {}
        outs.append(eng.forward(x, coords, ts, taps=tp))
        taps.append(tp)
    for o, tp in zip(outs[1:], taps[1:]):
        for i in range(3):
            assert maxabs(o["imgt_pred"][i], outs[0]["imgt_pred"][i]) < 1e-5
            assert tuple(o["imgt_pred"][i].shape) == (2, 3, 128, 128)
            for k in ("flowt0_pred", "flowt1_pred"):
                for a, b in zip(o[k][i], outs[0][k][i]):
                    assert a.shape == b.shape and maxabs(a, b) < 1e-4
            assert maxabs(o["other_pred"][i][0], outs[0]["other_pred"][i][0]) < 1e-5
            assert maxabs(o["flowt"][i], outs[0]["flowt"][i]) < 1e-4     # (same launches; the splat's float atomics reorder)
            for name in ("init_ft_4", "upd_ft_4", "upd_flow0_4", "final_res"):
                assert maxabs(tp[f"t{i}_{name}"], taps[0][f"t{i}_{name}"]) < 1e-4, (i, name)


@pytest.mark.parametrize("precision,name", [("fp32", "r_128x192_t050"), ("bf16", "r_b2_128x128_t025_075"), ("bf16", "r_256x256_ds050_t050")])
def test_whole_model_on_the_emulated_kernels(precision, name, sd):
    """The forward with EVERY launch -- the convolution kernels included (MFMA fragments, LDS-DMA rings, halo staging, fused
    epilogues) -- on the host build of the real kernel sources, against the goldens the reference generated: the GPU suite's
    `test_*_matches_reference_golden` without a GPU (fp32 lands where the MI355X does, 141 dB; bf16 on the batch-2 / two-timestep
    and on the DS_SCALE 0.5 fixture).  Possible since the emulator runs lanes as fibers: 0.4 - 1.5 minutes per forward (the
    thread-per-lane form needed hours).  The LDS-DMAs run under the emulator's adversarial timing."""
    meta, gold = load_golden(name)
    x, coords, ts = golden_inputs(meta)
    rt = SimRuntime(precision, emulate_conv=True)
    rt.lib.dll.gvfi_emu_set_dma_mode(1)       # (adversarial LDS-DMA timing, tests/hostsim/hip_emu.h: same results, or NaNs)
    rt.lib.dll.gvfi_emu_set_sched(3)          # (... and the waves in reverse order, depth first)
    try:
        out = Engine(rt, sd).forward(x, coords, ts, ds_factor=meta["ds"])
    finally:
        rt.lib.dll.gvfi_emu_set_sched(0)
    assert maxabs(out["raft_flow"], gold["raft_flow"]) < (1e-4 if precision == "fp32" else 0.1)
    for i in range(len(meta["t"])):
        p = psnr(out["imgt_pred"][i], gold[f"imgt_pred_{i}"])
        d = (out["flowt"][i].float() - gold[f"flowt_{i}"]).abs().flatten()
        print(f"WHOLE-MODEL EMULATION {precision} {name} t[{i}]: {rt.n_launch} launches, PSNR(imgt_pred vs reference golden) = {p:.2f} dB, "
              f"mean|flowt err| = {float(d.mean()):.3e}")
        if precision == "fp32":
            assert p >= 120.0, p
            assert float(d.kthvalue(int(d.numel() * 0.999))[0]) < 2e-3
            assert maxabs(out["flowt0_pred"][i][1], gold[f"flowt0_4_{i}"]) < 2e-3
        else:
            assert p >= 60.0, p          # (the GPU runs of these fixtures: 71 - 76 dB; the gate of the GPU suite is 40)
            assert float(d.mean()) < 0.05


def test_zero_once_buffers_are_released_with_the_signature_that_used_them():
    """Runtime's persistent zero-once buffers (the largest activations of a forward) used to pile up under shape churn.
    They are now booked per input signature (`once_scope`, set by the models around a forward) and dropped by release_once()
    when that signature's graph is evicted -- unless another live signature uses the same buffer."""
    rt = SimRuntime("bf16")
    rt.once_scope = "sig A"
    a1 = rt.act(1, 4, 4, 3, once="x")            # 3 channels: padded pitch, zero state needed once
    a2 = rt.f32(2, 5, zero=True, once="y")
    assert rt.act(1, 4, 4, 3, once="x") is a1    # handed out again, no new tensor
    rt.once_scope = "sig B"
    assert rt.act(1, 4, 4, 3, once="x") is a1    # same (name, shape): shared between the signatures
    b1 = rt.act(1, 8, 8, 3, once="x")
    n_all = rt.once_bytes()
    assert n_all == sum(t.numel() * t.element_size() for t in (a1, a2, b1))
    rt.release_once("sig A")                     # `y` goes, `x` 4x4 stays (sig B uses it)
    assert rt.once_bytes() == n_all - a2.numel() * a2.element_size()
    assert rt.act(1, 4, 4, 3, once="x") is a1
    rt.release_once("sig B")
    assert rt.once_bytes() == 0
    rt.release_once("never seen")                # harmless
    rt.once_scope = None
    c = rt.act(1, 4, 4, 3, once="x")             # outside any signature: kept for the runtime's life, as before
    assert c is not a1 and rt.once_bytes() > 0


def test_serial_launch_switch_turns_every_parallel_launch_sequence_off_and_back(sd):
    """Engine.set_serial_launch (the linear-graph slot kind of StepsInFlight): all five lane switches off, and restored exactly."""
    from hostsim.sim_runtime import SimRuntime
    from gimmvfi_hip.engine import Engine

    eng = Engine(SimRuntime("fp32"), sd)
    names = ("raft_lanes", "synth_lanes", "misc_lanes", "enc_lanes", "post_lanes")
    before = {n: getattr(eng, n) for n in names}
    assert before["raft_lanes"] == 2 and before["enc_lanes"] and before["post_lanes"] and before["synth_lanes"]
    eng.set_serial_launch(True)
    assert eng.raft_lanes == 1 and not any(getattr(eng, n) for n in names[1:])
    eng.set_serial_launch(True)
    eng.set_serial_launch(False)
    assert {n: getattr(eng, n) for n in names} == before
