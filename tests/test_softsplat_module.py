"""The reference's native-op boundary: ``softsplat_func.apply(tenIn NCHW f32, tenFlow)`` and the
python wrapper ``softsplat(tenIn, tenFlow, tenMetric, strMode)`` (reference modules/softsplat.py:286-446).

* CPU: the reference's OWN ``softsplat()`` python runs on top of this repo's kernel (``gvfi_softsplat_out_nchw``, host
  emulator build of the same source) and must agree with the reference on its CPU restatement of the CuPy kernel
  (oracle/ref_harness.py) -- i.e. the kernel is a drop-in under the reference's wrapper;
* CPU + GPU: ``gimmvfi_hip.softsplat.softsplat`` (our mirror of that wrapper) against the oracle for every mode."""
import sys

import pytest
import torch

import gimmvfi_r_oracle as orc
import ref_harness as rh

MODES = ["sum", "avg", "linear", "linear-addeps", "linear-zeroeps", "linear-clipeps", "softmax", "softmax-zeroeps"]


def _inputs(seed=3, N=2, C=6, H=17, W=23):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g)
    flow = torch.randn(N, 2, H, W, generator=g) * 4
    flow[1, :, :, :5] = 30.0            # a band leaves the image -> holes (zero denominators)
    flow[0, 0, 2, 3] = float("inf")     # skipped source pixel
    metric = torch.rand(N, 1, H, W, generator=g) + 0.25
    return x, flow, metric


def _oracle(x, flow, metric, mode):
    """softsplat.py:286-352 restated on oracle.splat_sum."""
    kind = mode.split("-")[0]
    if kind == "avg":
        x = torch.cat([x, x.new_ones(x.shape[0], 1, *x.shape[2:])], 1)
    elif kind == "linear":
        x = torch.cat([x * metric, metric], 1)
    elif kind == "softmax":
        x = torch.cat([x * metric.exp(), metric.exp()], 1)
    o = orc.splat_sum(x, flow)
    if kind == "sum":
        return o
    nrm = o[:, -1:].clone()
    suffix = mode.split("-")[1] if "-" in mode else "addeps"
    if suffix == "addeps":
        nrm = nrm + 0.0000001
    elif suffix == "zeroeps":
        nrm[nrm == 0.0] = 1.0
    else:
        nrm = nrm.clip(0.0000001, None)
    return o[:, :-1] / nrm


class _HostsimSplat:
    """softsplat_func stand-in that runs gvfi_softsplat_out_nchw of the host emulator build."""

    @staticmethod
    def apply(tenIn, tenFlow):
        from sim_runtime import hostsim_lib

        a, f = tenIn.float().contiguous(), tenFlow.float().contiguous()
        out = a.new_zeros(a.shape)
        n, c, h, w = a.shape
        assert hostsim_lib().softsplat_out_nchw(a.data_ptr(), f.data_ptr(), out.data_ptr(), n, c, h, w, 0) == 0
        return out


@pytest.mark.skipif(not rh.reference_available(), reason="reference checkout only exists in the dev container")
@pytest.mark.parametrize("mode", MODES)
def test_reference_softsplat_python_on_top_of_our_kernel(mode):
    rh.load_reference_modules()
    ref_mod = sys.modules[rh._PKG + ".generalizable_INR.modules.softsplat"]
    x, flow, metric = _inputs()
    met = None if mode in ("sum", "avg") else metric
    want = ref_mod.softsplat(x, flow, met, mode)            # reference wrapper + harness CPU statement of the CuPy kernel
    keep = ref_mod.softsplat_func
    ref_mod.softsplat_func = _HostsimSplat
    try:
        got = ref_mod.softsplat(x, flow, met, mode)         # reference wrapper + THIS repo's kernel
    finally:
        ref_mod.softsplat_func = keep
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= 2e-5 * scale
    assert float((want - _oracle(x, flow, met, mode)).abs().max()) <= 2e-5 * scale    # and the oracle says the same


@pytest.mark.parametrize("mode", MODES)
def test_our_wrapper_matches_oracle_hostsim(mode, monkeypatch):
    from gimmvfi_hip import softsplat as ss

    monkeypatch.setattr(ss, "softsplat_func", _HostsimSplat)
    x, flow, metric = _inputs(seed=4)
    met = None if mode in ("sum", "avg") else metric
    got = ss.softsplat(x, flow, met, mode)
    want = _oracle(x, flow, met, mode)
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    if mode == "linear-zeroeps":
        o, n = ss.softsplat(x, flow, met, mode, return_norm=True)
        assert o.shape[1] == x.shape[1] and n.shape[1] == 1 and float((n == 0).sum()) == 0


def test_wrapper_error_behaviour():
    from gimmvfi_hip import softsplat as ss

    x, flow, metric = _inputs()
    with pytest.raises(AssertionError):
        ss.softsplat(x, flow, metric, "sum")           # softsplat.py:289-290: "sum" takes no metric
    with pytest.raises(AssertionError):
        ss.softsplat(x, flow, None, "linear")
    with pytest.raises(AssertionError):
        ss.softsplat(x, flow, metric, "median")
    with pytest.raises(AssertionError):
        ss.softsplat_func.apply(x, flow)               # CPU tensors: softsplat.py:439-440 `assert False`
    with pytest.raises(AssertionError):                # "avg-<suffix>": the reference divides by the last INPUT channel -- rejected
        ss.softsplat(x, flow, None, "avg-zeroeps")


@pytest.mark.parametrize("metric_given", [False, True])
def test_sum_with_a_suffix_is_the_plain_sum(monkeypatch, metric_given):
    """softsplat.py:286-352: only the bare word "sum" asserts `tenMetric is None`; "sum-<anything>" appends no weight channel and
    is never normalised, i.e. it IS the splatted sum, with or without a metric."""
    from gimmvfi_hip import softsplat as ss

    monkeypatch.setattr(ss, "softsplat_func", _HostsimSplat)
    x, flow, metric = _inputs(seed=6)
    got = ss.softsplat(x, flow, metric if metric_given else None, "sum-addeps")
    want = ss.softsplat(x, flow, None, "sum")
    # (two runs of the float-atomic scatter: same sums up to the order of the additions)
    assert got.shape == x.shape and float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
    if rh.reference_available():
        rh.load_reference_modules()
        ref_mod = sys.modules[rh._PKG + ".generalizable_INR.modules.softsplat"]
        ref = ref_mod.softsplat(x, flow, metric if metric_given else None, "sum-addeps")
        assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_unknown_eps_suffix_leaves_the_normaliser_untouched(monkeypatch):
    """softsplat.py:325-334 is an if / elif chain without an else: "linear-foo" divides by the raw splatted weight (holes
    give 0 / 0 = NaN and trip the NaN check).  On hole-free input the result equals the plain quotient."""
    from gimmvfi_hip import softsplat as ss

    monkeypatch.setattr(ss, "softsplat_func", _HostsimSplat)
    x, _, metric = _inputs(seed=6)
    flow = torch.zeros(x.shape[0], 2, *x.shape[2:]) + 0.25          # every target pixel receives weight
    o, n = ss.softsplat(x, flow, metric, "linear-foo", return_norm=True)
    raw = orc.splat_sum(torch.cat([x * metric, metric], 1), flow)
    assert torch.equal(n, raw[:, -1:]) or float((n - raw[:, -1:]).abs().max()) < 1e-6
    o2, n2 = ss.softsplat(x, flow, metric, "linear", return_norm=True)
    assert float((n2 - n).abs().max()) > 0 and float((n2 - n - 1e-7).abs().max()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("mode", MODES)
def test_gpu_wrapper_matches_oracle(mode):
    from gimmvfi_hip import softsplat as ss

    x, flow, metric = _inputs(seed=5, N=2, C=16, H=64, W=96)
    met = None if mode in ("sum", "avg") else metric
    got = ss.softsplat(x.cuda(), flow.cuda(), None if met is None else met.cuda(), mode)
    want = _oracle(x, flow, met, mode)
    assert got.is_cuda and got.dtype == torch.float32 and got.shape == want.shape
    assert float((got.cpu() - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
