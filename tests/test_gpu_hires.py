"""Parity at the configurations BASELINE.json's metric is quoted on.

(a) 448x256 (configs[0]/[1]): the HIP path against the CPU oracle run LIVE on the GPU box's host cores on the bench
    input (B=1, t=0.5) -- fp32 mode PSNR >= 80 dB / flows p99.9 < 2e-3 px, bf16 mode PSNR >= 40 dB / mean flow error
    < 0.05 px; and the B=8 bench batch in bf16 against the same oracle sample.
(b) 2K DS 0.5 and 4K DS 0.25, 8x (configs[2]/[4], reference README.md:87-96) on a seeded synthetic pair, and the
    reference's own demo frames (844x720 padded to 864x736 at DS 1; 2048x1080 padded to 2048x1088 at DS 0.5), against
    fixtures produced by the REFERENCE ITSELF (oracle/make_golden_hires.py; tests/golden/hr_r_*.npz): uint8 crops
    within 1 LSB (fp32 mode) / PSNR >= 40 dB (bf16), 16x16 block means of the whole frame, INR flows.
"""
import json
import os

import numpy as np
import pytest
import torch

import gimmvfi_r_oracle as orc
from util import GOLDEN, psnr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(sd, precision, kind="r"):
    from gimmvfi_hip.model import GIMMVFI_F, GIMMVFI_R

    m = (GIMMVFI_F if kind == "f" else GIMMVFI_R)(precision=precision)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


# ------------------------------------------------------------------------------------------------ (a) 448x256, live oracle
@pytest.fixture(scope="module")
def bench_oracle(sd):
    from gimmvfi_hip.synth import synthetic_pairs

    x = synthetic_pairs(8, 256, 448, seed=100)           # bench.py's rank-0 batch
    coords = [(orc.sample_coord_input(1, (256, 448), [0.5], 1.0), None)]
    ts = [0.5 * torch.ones(1)]
    with torch.no_grad():
        ref = orc.forward(sd, x[:1], coords, ts, None)
    return x, ref


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_448x256_b1_vs_live_oracle(sd, bench_oracle, prec):
    x, ref = bench_oracle
    m = _model(sd, prec)
    c = [(m.sample_coord_input(1, (256, 448), [0.5], device=DEV), None)]
    out = m(x[:1].to(DEV), c, t=[0.5 * torch.ones(1, device=DEV)])
    torch.cuda.synchronize()
    p = psnr(out["imgt_pred"][0], ref["imgt_pred"][0])
    d = (out["flowt"][0].cpu().float() - ref["flowt"][0]).abs().flatten()
    dr = (out["raft_flow"].cpu().float() - ref["raft_flow"]).abs().flatten()
    print(f"448x256 B=1 {prec}: PSNR {p:.2f} dB, flowt mean {float(d.mean()):.2e} p99.9 "
          f"{float(d.kthvalue(int(d.numel() * 0.999))[0]):.2e}, raft_flow max {float(dr.max()):.2e}")
    if prec == "fp32":
        assert p >= 80.0, p
        assert float(d.kthvalue(int(d.numel() * 0.999))[0]) < 2e-3
        assert float(dr.max()) < 2e-3
    else:
        assert p >= 40.0, p
        assert float(d.mean()) < 0.05


def _oracle_batch(orc_mod, sd_, x):
    """The CPU oracle on every pair of a batch, one pair at a time (a few seconds each on the GPU box's host cores)."""
    coords = [(orc_mod.sample_coord_input(1, tuple(x.shape[-2:]), [0.5], 1.0), None)]
    ts = [0.5 * torch.ones(1)]
    refs = []
    with torch.no_grad():
        for b in range(x.shape[0]):
            refs.append(orc_mod.forward(sd_, x[b:b + 1], coords, ts, None))
    return refs


def _check_batch(out, refs, tag, min_psnr, max_flow_mean):
    worst_p, worst_f = 1e9, 0.0
    for b, ref in enumerate(refs):
        assert torch.isfinite(out["imgt_pred"][0][b]).all()
        p = psnr(out["imgt_pred"][0][b:b + 1], ref["imgt_pred"][0])
        d = (out["flowt"][0][b].cpu().float() - ref["flowt"][0]).abs().flatten()
        worst_p, worst_f = min(worst_p, p), max(worst_f, float(d.mean()))
        print(f"{tag} sample {b}: PSNR {p:.2f} dB, flowt mean err {float(d.mean()):.2e} p99.9 "
              f"{float(d.kthvalue(int(d.numel() * 0.999))[0]):.2e} px (max |flow| {float(ref['flowt'][0].abs().max()):.1f})")
    assert worst_p >= min_psnr, worst_p
    assert worst_f <= max_flow_mean, worst_f


def test_448x256_b8_bench_batch_bf16_all_samples_vs_live_oracle(sd):
    """The very forward bench.py times (BASELINE.json configs[1]: 8 pairs, bf16, hipGraph replay): EVERY sample of the
    batch against the CPU oracle run live on this box."""
    from gimmvfi_hip.synth import synthetic_pairs

    x = synthetic_pairs(8, 256, 448, seed=100)           # bench.py's rank-0 batch
    refs = _oracle_batch(orc, sd, x)
    m = _model(sd, "bf16")
    c = [(m.sample_coord_input(8, (256, 448), [0.5], device=DEV), None)]
    t = [0.5 * torch.ones(8, device=DEV)]
    for _ in range(2):                                   # second call = graph replay
        out = m(x.to(DEV), c, t=t)
    torch.cuda.synchronize()
    _check_batch(out, refs, "R 448x256 B=8 bf16", 40.0, 0.05)


@pytest.fixture(scope="module")
def bench_oracle_f(sd_f):
    import gimmvfi_f_oracle as forc
    from gimmvfi_hip.synth import synthetic_pairs

    x = synthetic_pairs(8, 256, 448, seed=100)           # bench.py --model f, rank-0 batch
    return x, _oracle_batch(forc, sd_f, x)


# GIMM-VFI-F at the size its metric is quoted on (BASELINE.json configs[3]); gates per mode: (min PSNR, max mean flow error)
# measured (round 3): float 77.6 dB / 7e-5 px (31 px flows of an un-trained recurrence: a 1e-4 px difference already flips
# fold-overs); all-bf16 45.3-47.6 dB / 0.16-0.25 px over the 8 samples; decoder in float 58.5 dB / 0.067 px
F448 = {"fp32": (70.0, 2e-3), "bf16-fast": (40.0, 0.30), "bf16": (50.0, 0.12), "bf16+dec": (50.0, 0.12)}


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16-fast", "bf16+dec"])
def test_448x256_f_b1_vs_live_oracle(sd_f, bench_oracle_f, mode):
    """GIMM-VFI-F 448x256, one pair, against the CPU oracle of the FlowFormer model (bit-exact with the reference,
    tests/test_oracle_pin.py) run live: float mode, bf16 with the default precision policy of the flow estimator (decoder
    on IEEE-half operands), the all-bf16 fast mode, and bf16 with the decoder in float."""
    from gimmvfi_hip.model import GIMMVFI_F

    x, refs = bench_oracle_f
    m = GIMMVFI_F(precision="fp32" if mode == "fp32" else "bf16",
                  flow_precision={"bf16+dec": "dec", "bf16-fast": "bf16"}.get(mode))     # None = the default policy (f16)
    m.load_state_dict(sd_f, strict=True)
    m = m.to(DEV).eval()
    c = [(m.sample_coord_input(1, (256, 448), [0.5], device=DEV), None)]
    out = m(x[:1].to(DEV), c, t=[0.5 * torch.ones(1, device=DEV)])
    torch.cuda.synchronize()
    out = {k: ([v_.reshape(1, *v_.shape) if k == "flowt" and v_.dim() == 3 else v_ for v_ in v] if isinstance(v, list) else v)
           for k, v in out.items()}
    _check_batch(out, refs[:1], f"F 448x256 B=1 {mode}", *F448[mode])


def test_448x256_f_b8_bench_batch_bf16_all_samples_vs_live_oracle(sd_f, bench_oracle_f):
    """The forward `bench.py --model f` times (8 pairs, bf16, hipGraph replay): every sample against the live oracle."""
    x, refs = bench_oracle_f
    m = _model(sd_f, "bf16", "f")
    c = [(m.sample_coord_input(8, (256, 448), [0.5], device=DEV), None)]
    t = [0.5 * torch.ones(8, device=DEV)]
    for _ in range(2):
        out = m(x.to(DEV), c, t=t)
    torch.cuda.synchronize()
    assert m.flow_precision == "f16"
    _check_batch(out, refs, "F 448x256 B=8 bf16 (f16)", *F448["bf16"])


# ------------------------------------------------------------------------------------------------ (b) reference fixtures
HR_CASES = ["demo_864x736", "2k_ds050", "demo2k_ds050", "4k_ds025"]


def load_hr(name, model="r"):
    path = os.path.join(GOLDEN, f"hr_{model}_{name}.npz")
    if not os.path.isfile(path):
        pytest.skip(f"{path} not generated")
    z = np.load(path)
    return json.loads(str(z["meta"])), z


def hr_inputs(meta, z):
    """The padded (1,3,2,Hp,Wp) input of a fixture: synthetic pairs are re-generated from the seed (guarded by the
    stored pixel sum), demo frames come from the fixture and go through the CLI's InputPadder (src/video_Nx.py:155-157)."""
    if meta["kind"] == "synthetic":
        from gimmvfi_hip.synth import synthetic_pairs

        x = synthetic_pairs(1, meta["H"], meta["W"], meta["seed"])
    else:
        import sys

        from util import ROOT

        sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd", "src"))
        from utils.utils import InputPadder

        frames = z["frames_u8"] if "frames_u8" in z.files else np.load(os.path.join(GOLDEN, meta["frames_from"] + ".npz"))["frames_u8"]
        fr = [torch.from_numpy(np.ascontiguousarray(f)).permute(2, 0, 1).float().div(255.0).unsqueeze(0)
              for f in frames]
        i0, i2 = InputPadder(fr[0].shape, 32).pad(fr[0], fr[1])
        x = torch.stack([i0, i2], 2)
    assert tuple(x.shape[-2:]) == (meta["Hp"], meta["Wp"])
    got = int(torch.round(x * 255.0).to(torch.int64).sum())
    if got != meta["in_sum"]:
        # a different host CPU may round a few of the 10^7 synthetic pixels the other way (bicubic / bilinear kernels
        # of another ISA level); a handful of LSBs does not matter for the gates below, anything more fails them
        print(f"note: re-generated input differs from the fixture's by {got - meta['in_sum']} LSB in total")
        assert abs(got - meta["in_sum"]) < 1000, "synthetic input re-generation differs"
    return x


def run_hr(m, meta, x):
    N, ds = meta["N"], meta["ds"]
    B, (Hp, Wp) = 1, x.shape[-2:]
    ratio = 1.0 if ds is None else ds
    coords = [(m.sample_coord_input(B, (Hp, Wp), [i / N], device=DEV, upsample_ratio=ratio), None) for i in range(1, N)]
    ts = [(i / N) * torch.ones(B, device=DEV) for i in range(1, N)]
    out = m(x.to(DEV), coords, t=ts, ds_factor=ds)
    torch.cuda.synchronize()
    return out


def hr_metrics(out, meta, z):
    """Worst-case distances of a forward's outputs from a reference fixture (over the kept timesteps / crops)."""
    T = meta["N"] - 1
    assert len(out["imgt_pred"]) == T
    cyx = z["crop_yx"]
    m = dict(psnr=1e9, lsb=0, bm=0.0, flow999=0.0, frac=0.0, bm999=0.0, fmed=0.0, fmean=0.0)
    for i in range(T):
        img = out["imgt_pred"][i][0].float().cpu()
        assert tuple(img.shape) == (3, meta["Hp"], meta["Wp"]) and torch.isfinite(img).all()
        bm = img.reshape(3, meta["Hp"] // 16, 16, meta["Wp"] // 16, 16).mean(dim=(2, 4))
        dbm = (bm - torch.from_numpy(z[f"bm_{i}"])).abs().flatten()
        m["bm"] = max(m["bm"], float(dbm.max()))
        m["bm999"] = max(m["bm999"], float(dbm.kthvalue(int(dbm.numel() * 0.999))[0]))
        if i not in meta["keep"]:
            continue
        u8 = torch.round(img.clamp(0, 1) * 255.0)
        ref = torch.from_numpy(z[f"crops_{i}"]).float()
        got = torch.stack([u8[:, y:y + ref.shape[-2], x_:x_ + ref.shape[-1]] for y, x_ in cyx])
        dl = (got - ref).abs()
        m["lsb"] = max(m["lsb"], int(dl.max()))
        m["frac"] = max(m["frac"], float((dl > 1).float().mean()))
        mse = float(((got - ref) / 255.0).pow(2).mean())
        m["psnr"] = min(m["psnr"], 99.0 if mse == 0 else -10.0 * np.log10(mse))
        ft = out["flowt"][i].float().cpu()
        ft = ft if ft.dim() == 3 else ft[0]
        rf = torch.from_numpy(z[f"flowt_{i}"].astype(np.float32))
        assert tuple(ft[:, ::2, ::2].shape) == tuple(rf.shape)
        d = ((ft[:, ::2, ::2] - rf).abs() - 1.5e-3 * rf.abs()).clamp_min(0).flatten()   # fp16 storage of the fixture: 2^-11 relative
        m["flow999"] = max(m["flow999"], float(d.kthvalue(int(d.numel() * 0.999))[0]))
        m["fmed"] = max(m["fmed"], float(d.median()))
        m["fmean"] = max(m["fmean"], float(d.mean()))
    return m


def fmt_metrics(m, meta):
    return (f"crops max |d| {m['lsb']} LSB ({m['frac']:.1e} of the pixels > 1 LSB), min PSNR {m['psnr']:.2f} dB, "
            f"block-mean |d| max {m['bm']:.2e} p99.9 {m['bm999']:.2e}, flowt |d| mean {m['fmean']:.2e} median {m['fmed']:.2e} "
            f"p99.9 {m['flow999']:.2e} px (max |flow| {meta['flow_absmax']:.1f})")


def check_hr(out, meta, z, prec, tag, kind="r", bounds=None):
    m = hr_metrics(out, meta, z)
    worst_psnr, worst_lsb, worst_bm, worst_flow, worst_frac, worst_bm999, worst_fmed = (
        m["psnr"], m["lsb"], m["bm"], m["flow999"], m["frac"], m["bm999"], m["fmed"])
    print(f"{tag} {prec}: " + fmt_metrics(m, meta))
    # The reference formula has discontinuities: splat holes (0/0 -> 1, softsplat.py:333-334) and foldovers flip on a
    # 1e-6 flow difference, more of them the rougher the flow (the seeded random weights give GIMM-VFI-F 40-50 px flows
    # full of them).  Hence: R fp32 everything within 1 LSB; F fp32 all but <= 2e-4 of the pixels (measured: 0 on three
    # cases, 7e-5 on the demo pair with 51 px flows); bf16 judged by PSNR, the fraction of crop pixels off by more than
    # 1 LSB and the mean / median / p99.9 flow error.
    if prec == "fp32":
        if kind == "r":
            assert worst_lsb <= 1, worst_lsb                    # +-1 LSB of the 8-bit frame
            assert worst_bm <= 2e-4, worst_bm
        assert worst_frac <= (0.0 if kind == "r" else 2e-4), worst_frac
        assert worst_psnr >= 60.0, worst_psnr
        assert worst_bm999 <= 2e-4, worst_bm999
        assert worst_flow <= (2e-3 if kind == "r" else 5e-3), worst_flow
    elif kind == "r":
        assert worst_psnr >= 40.0, worst_psnr
        assert worst_bm <= 1e-1, worst_bm      # one 16x16 block; sub-pixel flow differences at occlusion edges
        assert worst_flow <= 0.25, worst_flow
    else:
        # GIMM-VFI-F, bf16: `bounds` = (min PSNR, max fraction of crop pixels > 1 LSB, max mean flow error, max p99.9 flow
        # error) -- the default policy (decoder of the flow estimator in float) has one set for all cases, the all-bf16 fast
        # mode is pinned per case to what it measures (profiles/r3_f_policy.md) plus a margin
        min_psnr, max_frac, max_fmean, max_f999 = bounds
        assert worst_psnr >= min_psnr, worst_psnr
        assert worst_frac <= max_frac, worst_frac
        assert m["fmean"] <= max_fmean, m["fmean"]
        assert worst_flow <= max_f999, worst_flow


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("name", HR_CASES)
def test_hires_matches_reference_fixture(sd, name, prec):
    meta, z = load_hr(name)
    x = hr_inputs(meta, z)
    m = _model(sd, prec)
    out = run_hr(m, meta, x)
    check_hr(out, meta, z, prec, name)
    del out, m
    torch.cuda.empty_cache()


# GIMM-VFI-F in bf16 against the reference fixtures.  The REQUIREMENT is the 40 dB of every other bf16 test, on every fixture.
# Default policy (flow_precision = "f16" since round 5: the whole flow estimator on IEEE-half operands; "dec:f16" before): per-
# case regression gates on top of it (VERDICT r3 #5a: one global bound set by the worst fixture let a 2 dB regression on the
# easy cases pass) = measured value (profiles/r5_f_policy_all.txt; two calls agree within 0.2 dB) - 1.5 dB / x1.3 / x1.3 / x1.2.
# Fast mode (flow_precision = "bf16"): per-case pins = measured value + margin (three calls of round 3 agree within 0.3 dB): it
# reaches 40 dB only while the flows are small (the *_fh015 fixtures below).
F_REQUIREMENT_DB = 40.0      # every bf16-mode test of this suite: >= 40 dB against the reference, on every fixture
# measured (PSNR dB, fraction of crop pixels > 1 LSB, mean flow error px, p99.9 flow error px), profiles/r6_gpu_parity_final.log
F_DEFAULT_MEASURED = {
    "demo_864x736": (52.9, 0.0073, 0.096, 6.4),    # (51 px flows; policy "dec:f16" of rounds 3-4: 50.3 dB)
    "2k_ds050": (50.3, 0.028, 0.063, 4.3),
    "demo2k_ds050": (43.6, 0.142, 0.085, 6.6),     # (the hardest case; "dec:f16": 41.6 dB; all-float flow estimator: 46.2 dB)
    "4k_ds025": (46.8, 0.15, 0.064, 4.3),
}
# gate = max(requirement, measured - 1.5 dB); the three error figures: measured x 1.3 / x 1.3 / x 1.2
F_DEFAULT_BOUNDS = {k: (max(F_REQUIREMENT_DB, v[0] - 1.5), v[1] * 1.3, v[2] * 1.3, v[3] * 1.2) for k, v in F_DEFAULT_MEASURED.items()}
F_FAST_BOUNDS = {
    "demo_864x736": (37.5, 0.14, 0.50, 14.5),      # measured 39.0 dB, 0.11, 0.41 px, 12.6 px
    "2k_ds050": (38.3, 0.25, 0.26, 9.6),           # 39.8 dB, 0.20, 0.21 px, 8.3 px
    "demo2k_ds050": (31.0, 0.36, 0.50, 14.4),      # 32.3-32.5 dB, 0.31, 0.42 px, 12.5 px
    "4k_ds025": (33.4, 0.52, 0.27, 9.9),           # 34.6-35.0 dB, 0.45, 0.22 px, 8.6 px
    "demo2k_ds050_fh015": (48.0, 0.06, 0.04, 0.12),    # 52.1 dB, 0.037, 0.028 px, 0.088 px   (flows <= 9.2 px)
    "4k_ds025_fh015": (50.0, 0.01, 0.03, 0.09),        # 53.9 dB, 0.0016, 0.018 px, 0.060 px  (flows <= 6.8 px)
}


def _model_f(sd_, mode):
    from gimmvfi_hip.model import GIMMVFI_F

    m = GIMMVFI_F(precision="fp32" if mode == "fp32" else "bf16", flow_precision="bf16" if mode == "bf16-fast" else None)
    if mode == "bf16":
        assert m.flow_precision == "f16"      # the model's default policy
    m.load_state_dict(sd_, strict=True)
    return m.to(DEV).eval()


def _sd_for(meta, sd_f):
    sc = meta.get("flow_head_scale", 1.0)
    if sc == 1.0:
        return sd_f
    from gimmvfi_hip.params import random_state_dict_f

    return random_state_dict_f(0, flow_head_scale=sc)


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16-fast"])
@pytest.mark.parametrize("name", ["demo_864x736", "2k_ds050", "demo2k_ds050", "4k_ds025"])
def test_hires_f_matches_reference_fixture(sd_f, name, mode):
    """GIMM-VFI-F (FlowFormer flow estimator, BASELINE.json configs[3]/[4]) on the same hi-res cases, against fixtures of
    the reference's GIMMVFI_F (oracle/make_golden_hires.py --model f; Twins from the reference's vendored class, see
    README 'timm'): float mode, bf16 with the default precision policy of the flow estimator, and the all-bf16 fast mode."""
    meta, z = load_hr(name, "f")
    x = hr_inputs(meta, z)
    m = _model_f(sd_f, mode)
    out = run_hr(m, meta, x)
    check_hr(out, meta, z, "fp32" if mode == "fp32" else "bf16", f"F {name} [{mode}]", "f",
             bounds=F_DEFAULT_BOUNDS[name] if mode == "bf16" else F_FAST_BOUNDS[name])
    del out, m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["fp32", "bf16", "bf16-fast"])
@pytest.mark.parametrize("name", ["demo2k_ds050_fh015", "4k_ds025_fh015"])
def test_hires_f_small_flows_separate_conditioning_from_arithmetic(sd_f, name, mode):
    """The same reference model with the decoder's flow head damped by 0.15 (params.random_state_dict_f(0,
    flow_head_scale=0.15); fixtures from the reference run with those weights): flows of <= 9 px instead of 40-50 px of
    folds.  Here the ALL-bf16 path is 52-54 dB from the reference with a p99.9 flow error below 0.1 px -- the 32-40 dB of
    the large-flow fixtures are the conditioning of the un-trained recurrence, not an arithmetic fault of the bf16 path."""
    meta, z = load_hr(name, "f")
    assert meta["flow_head_scale"] == 0.15 and meta["flow_absmax"] < 10.0
    x = hr_inputs(meta, z)
    m = _model_f(_sd_for(meta, sd_f), mode)
    out = run_hr(m, meta, x)
    check_hr(out, meta, z, "fp32" if mode == "fp32" else "bf16", f"F {name} [{mode}]", "f",
             bounds=(50.0, 0.01, 0.03, 0.10) if mode == "bf16" else F_FAST_BOUNDS[name])
    del out, m
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ flow-scale families (VERDICT r3 #5b)
# No trained checkpoint exists offline; the seeded weights give the 32-iteration decoder 30-50 px flows full of fold-overs.  The
# decoder's flow head scaled by S walks max |flow| from a few pixels (what a trained estimator produces on ordinary footage) to
# that extreme.  Fixtures = the reference itself with those weights (oracle/make_golden_f448.py, make_golden_hires.py
# --flow-head-scale); the printed lines are collected into profiles/r4_f_flow_scale_curve.md by tools/f_flow_curve.py.
# bounds per S: (min PSNR dB, max p99.9 flow error px) for the default policy -- measured value - 2 dB / x1.4
# measured (round 4, profiles/r4_f_flow_scale_curve.md): 62.9 / 59.1 / 56.5 / 54.2 dB, p99.9 0.042 / 0.116 / 0.35 / 3.4 px
F448_FAMILY = {0.15: (60.8, 0.06), 0.4: (57.0, 0.165), 0.7: (54.5, 0.49), 1.0: (52.2, 4.8)}


@pytest.mark.parametrize("scale", sorted(F448_FAMILY))
def test_f_448_b8_flow_scale_family(scale):
    path = os.path.join(GOLDEN, f"f448_fh{int(round(scale * 100)):03d}.npz")
    if not os.path.isfile(path):
        pytest.skip(f"{path} not generated")
    from gimmvfi_hip.params import random_state_dict_f
    from gimmvfi_hip.synth import synthetic_pairs

    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    x = synthetic_pairs(8, 256, 448, seed=100)
    m = _model_f(random_state_dict_f(0, flow_head_scale=scale), "bf16")
    c = [(m.sample_coord_input(8, (256, 448), [0.5], device=DEV), None)]
    t = [0.5 * torch.ones(8, device=DEV)]
    for _ in range(2):                                   # second call = graph replay (the forward bench.py times)
        out = m(x.to(DEV), c, t=t)
    torch.cuda.synchronize()
    min_psnr, max_f999 = F448_FAMILY[scale]
    for b in range(meta["samples"]):
        img = out["imgt_pred"][0][b].float().cpu()
        u8 = torch.round(img.clamp(0, 1) * 255.0)
        ref = torch.from_numpy(z[f"img_{b}"]).float()
        mse = float(((u8 - ref) / 255.0).pow(2).mean())
        p = 99.0 if mse == 0 else -10.0 * np.log10(mse)
        rf = torch.from_numpy(z[f"flowt_{b}"].astype(np.float32))
        d = ((out["flowt"][0][b].float().cpu()[:, ::2, ::2] - rf).abs() - 1.5e-3 * rf.abs()).clamp_min(0).flatten()
        f999 = float(d.kthvalue(int(d.numel() * 0.999))[0])
        print(f"FAMILY 448x256 fh={scale:.2f} sample {b}: max |flow| {meta['flow_absmax'][b]:.1f} px, PSNR {p:.2f} dB, "
              f"pixels > 1 LSB {float(((u8 - ref).abs() > 1).float().mean()):.2e}, flowt |d| mean {float(d.mean()):.2e} p99.9 {f999:.2e} px")
        assert p >= min_psnr, (scale, b, p)
        assert f999 <= max_f999, (scale, b, f999)
    del out, m
    torch.cuda.empty_cache()


# demo-2K pair (the reference's own 2048x1080 frames, DS 0.5, 8x): the same family; S = 0.15 and 1.0 are the fixtures above
# (min PSNR, frac > 1 LSB, mean flow err, p99.9); measured 50.0 dB, 0.048, 0.040 px, 0.18 px (max |flow| 22.9 px) and
# 44.1 dB, 0.12, 0.069 px, 7.9 px (38.9 px) -- between the 52-55 dB of S = 0.15 (9 px) and the 41.6 dB of S = 1 (50 px)
F2K_FAMILY = {0.4: (48.4, 0.063, 0.052, 0.24), 0.7: (42.5, 0.16, 0.09, 9.5)}


@pytest.mark.parametrize("scale", sorted(F2K_FAMILY))
def test_hires_f_demo2k_flow_scale_family(sd_f, scale):
    meta, z = load_hr(f"demo2k_ds050_fh{int(round(scale * 100)):03d}", "f")
    assert meta["flow_head_scale"] == scale
    x = hr_inputs(meta, z)
    m = _model_f(_sd_for(meta, sd_f), "bf16")
    out = run_hr(m, meta, x)
    check_hr(out, meta, z, "bf16", f"FAMILY demo2k_ds050 fh={scale:.2f} [bf16]", "f", bounds=F2K_FAMILY[scale])
    del out, m
    torch.cuda.empty_cache()
