"""Drop-in for the reference's native-op module ``modules/softsplat.py`` (its one CUDA/CuPy component on the path).

Same public names and contracts:

* ``softsplat_func.apply(tenIn, tenFlow) -> tenOut``  (reference softsplat.py:358-446): tenIn (N,C,H,W) f32, tenFlow
  (N,2,H,W) f32, both on the GPU; tenOut = ``tenIn.new_zeros(...)`` accumulated by the kernel on torch's current
  stream.  The reference JIT-compiles a CuPy kernel per shape; here it is ``gvfi_softsplat_out_nchw`` of
  libgimmvfi_hip.so.  CPU tensors raise ``AssertionError`` like the reference (``assert False``, softsplat.py:439-440).
  Inference only: there is no backward.
* ``softsplat(tenIn, tenFlow, tenMetric, strMode, return_norm=False)`` (softsplat.py:286-352): every mode of the
  reference ("sum", "avg", "linear[-addeps|-zeroeps|-clipeps]", "softmax[-...]"), the same asserts, the same NaN guards.

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


def softsplat(tenIn, tenFlow, tenMetric, strMode, return_norm=False):
    assert strMode.split("-")[0] in ["sum", "avg", "linear", "softmax"]
    if strMode == "sum":
        assert tenMetric is None
    if strMode == "avg":
        assert tenMetric is None
    if strMode.split("-")[0] == "linear":
        assert tenMetric is not None
    if strMode.split("-")[0] == "softmax":
        assert tenMetric is not None

    if strMode == "avg":
        tenIn = torch.cat([tenIn, tenIn.new_ones([tenIn.shape[0], 1, tenIn.shape[2], tenIn.shape[3]])], 1)
    elif strMode.split("-")[0] == "linear":
        tenIn = torch.cat([tenIn * tenMetric, tenMetric], 1)
    elif strMode.split("-")[0] == "softmax":
        tenIn = torch.cat([tenIn * tenMetric.exp(), tenMetric.exp()], 1)

    if torch.isnan(tenIn).any():
        print("NaN values detected during training in tenIn. Exiting.")
        assert False

    tenOut = softsplat_func.apply(tenIn, tenFlow)

    if torch.isnan(tenOut).any():
        print("NaN values detected during training in tenOut_1. Exiting.")
        assert False

    if strMode.split("-")[0] in ["avg", "linear", "softmax"]:
        tenNormalize = tenOut[:, -1:, :, :]
        if len(strMode.split("-")) == 1:
            tenNormalize = tenNormalize + 0.0000001
        elif strMode.split("-")[1] == "addeps":
            tenNormalize = tenNormalize + 0.0000001
        elif strMode.split("-")[1] == "zeroeps":
            tenNormalize[tenNormalize == 0.0] = 1.0
        elif strMode.split("-")[1] == "clipeps":
            tenNormalize = tenNormalize.clip(0.0000001, None)
        if return_norm:
            return tenOut[:, :-1, :, :], tenNormalize
        tenOut = tenOut[:, :-1, :, :] / tenNormalize

    if torch.isnan(tenOut).any():
        print("NaN values detected during training in tenOut_2. Exiting.")
        assert False
    return tenOut
