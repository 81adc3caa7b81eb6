"""Drop-in for the reference's native-op module ``modules/softsplat.py`` (its one CUDA/CuPy component on the path).

Same public names and contracts:

* ``softsplat_func.apply(tenIn, tenFlow) -> tenOut``  (reference softsplat.py:358-446): tenIn (N,C,H,W) f32, tenFlow
  (N,2,H,W) f32, both on the GPU; tenOut = ``tenIn.new_zeros(...)`` accumulated by the kernel on torch's current
  stream.  The reference JIT-compiles a CuPy kernel per shape; here it is ``gvfi_softsplat_out_nchw`` of
  libgimmvfi_hip.so.  CPU tensors raise ``AssertionError`` like the reference (``assert False``, softsplat.py:439-440).
  Inference only: there is no backward.
* ``softsplat(tenIn, tenFlow, tenMetric, strMode, return_norm=False)`` (contract of softsplat.py:286-352, written from
  that contract, not from its text): every mode of the reference ("sum", "avg", "linear[-addeps|-zeroeps|-clipeps]",
  "softmax[-...]"), AssertionError on a wrong mode / metric combination and on NaN inputs or outputs.  A caller who keeps
  the reference's own wrapper only swaps ``softsplat_func``.

The GIMM-VFI forward itself does not go through this module (it splats NHWC latents with ``gvfi_softsplat_accum`` / ``_normalize``); this is the
boundary for callers that use the reference's op directly.
"""
import torch

from . import lib as L


class softsplat_func:
    @staticmethod
    def apply(tenIn, tenFlow):
        if not (tenIn.is_cuda and tenFlow.is_cuda):
            assert False, "softsplat_func runs on the GPU only (reference softsplat.py:439-440)"
        assert tenFlow.shape[1] == 2 and tenFlow.shape[0] == tenIn.shape[0] and tenFlow.shape[2:] == tenIn.shape[2:]
        tenIn = tenIn.to(torch.float32).contiguous()       # custom_fwd(cast_inputs=torch.float32), softsplat.py:360
        tenFlow = tenFlow.to(torch.float32).contiguous()
        tenOut = tenIn.new_zeros([tenIn.shape[0], tenIn.shape[1], tenIn.shape[2], tenIn.shape[3]])
        n, c, h, w = tenIn.shape
        with torch.cuda.device(tenIn.device):
            rc = L.get().softsplat_out_nchw(tenIn.data_ptr(), tenFlow.data_ptr(), tenOut.data_ptr(), n, c, h, w,
                                            torch.cuda.current_stream(tenIn.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"gvfi_softsplat_out_nchw failed with code {rc}")
        return tenOut


_KINDS = ("sum", "avg", "linear", "softmax")
# how the normaliser (the splatted weight channel) is made safe to divide by   softsplat.py:322-334
_EPS_POLICY = {
    "": lambda n: n + 0.0000001,
    "addeps": lambda n: n + 0.0000001,
    "zeroeps": lambda n: torch.where(n == 0.0, torch.ones_like(n), n),
    "clipeps": lambda n: n.clip(0.0000001, None),
}


def _finite_or_die(t, where):
    # the reference prints a message and `assert False`s on NaN inputs / outputs (softsplat.py:310-312,317-319,349-351)
    assert not bool(torch.isnan(t).any()), f"softsplat: NaN values in {where}"


def softsplat(tenIn, tenFlow, tenMetric, strMode, return_norm=False):
    """Forward splat of tenIn along tenFlow (contract of reference softsplat.py:286-352).  strMode = kind[-eps policy]:
    kind "sum" (plain), "avg" (weight 1), "linear" (weight = metric), "softmax" (weight = exp(metric)); weighted kinds
    splat [tenIn * weight | weight] in one pass and divide by the splatted weight made non-zero by the eps policy
    (default / "addeps": + 1e-7, "zeroeps": 0 -> 1, "clipeps": clip at 1e-7).  return_norm: (numerator, normaliser)."""
    kind, _, eps = strMode.partition("-")
    assert kind in _KINDS, strMode
    # "sum-<anything>" is the plain splatted sum in the reference too (softsplat.py:289-296,322: no weight channel, no normalisation,
    # the metric is not even looked at), so it is accepted and means "sum".  "avg-<suffix>" is the one odd corner: the reference
    # only appends the ones channel for the bare word, yet normalises for every "avg-*" -- by the LAST INPUT CHANNEL; nothing
    # can rely on that, so it is rejected instead of reproduced
    assert kind != "avg" or eps == "", f"softsplat: mode {strMode!r}: 'avg' takes no eps suffix"
    if kind == "sum" and eps != "":
        tenMetric = None
    assert (tenMetric is None) == (kind in ("sum", "avg")), f"mode {kind!r}: metric {'not ' if tenMetric is None else ''}given"
    if kind == "sum":
        weight = None
    elif kind == "avg":
        weight = tenIn.new_ones([tenIn.shape[0], 1, tenIn.shape[2], tenIn.shape[3]])
    else:
        weight = tenMetric if kind == "linear" else tenMetric.exp()
    src = tenIn if weight is None else torch.cat([tenIn if kind == "avg" else tenIn * weight, weight], 1)
    _finite_or_die(src, "the splat input")
    acc = softsplat_func.apply(src, tenFlow)
    _finite_or_die(acc, "the splatted sums")
    if weight is None:
        return acc
    # an unknown suffix ("linear-foo") leaves the normaliser untouched, as the reference's if / elif chain does (softsplat.py:325-334)
    norm = _EPS_POLICY.get(eps, lambda n: n)(acc[:, -1:, :, :])
    if return_norm:
        return acc[:, :-1, :, :], norm
    out = acc[:, :-1, :, :] / norm
    _finite_or_die(out, "the normalised result")
    return out
