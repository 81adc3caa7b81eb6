"""Volume-free correlation lookup with the interface of the reference's native module ``alt_cuda_corr``
(flowformer/alt_cuda_corr/correlation.cpp:19-54) and its caller ``AlternateCorrBlock`` (raft/corr.py:96-124).

``forward(fmap1, fmap2, coords, radius) -> [corr]`` takes the same tensors as the pybind op: NHWC float32 feature maps
``(B,H1,W1,C)`` / ``(B,H2,W2,C)``, ``coords (B,N,H1,W1,2)`` and returns ``[corr (B,N,(2r+1)^2,H1,W1)]``.  The work is
the HIP kernel ``gvfi_alt_corr_forward``; like the original, inputs must be device tensors and contiguous (it raises
otherwise -- the CUDA op's TORCH_CHECKs), and there is no CPU fallback.
"""
import math

import torch

from . import lib as L


def _check(x, name):
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")        # CHECK_CUDA, correlation.cpp:19
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")           # CHECK_CONTIGUOUS, correlation.cpp:20


def forward(fmap1, fmap2, coords, radius):
    for t, n in ((fmap1, "fmap1"), (fmap2, "fmap2"), (coords, "coords")):
        _check(t, n)
    assert fmap1.dtype == fmap2.dtype and fmap1.dtype in (torch.float32, torch.bfloat16)
    B, N, H, W = coords.shape[:4]
    assert fmap1.shape[:3] == (B, H, W) and fmap2.shape[0] == B and fmap1.shape[3] == fmap2.shape[3]
    rd = 2 * radius + 1
    corr = torch.empty((B, N, rd * rd, H, W), dtype=torch.float32, device=fmap1.device)
    lib = L.get()
    rc = lib.alt_corr_forward(fmap1.data_ptr(), fmap2.data_ptr(), coords.float().contiguous().data_ptr(), corr.data_ptr(),
                              B, N, H, W, fmap2.shape[1], fmap2.shape[2], fmap1.shape[3], radius,
                              L.F32 if fmap1.dtype == torch.float32 else L.BF16,
                              torch.cuda.current_stream(fmap1.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"gvfi_alt_corr_forward failed with code {rc}")
    return [corr]


def avgpool2_nhwc(x):
    """F.avg_pool2d(x_nchw, 2, stride=2) for an NHWC tensor (the pyramid of raft/corr.py:101-105)."""
    _check(x, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, h // 2, w // 2, c), dtype=x.dtype, device=x.device)
    rc = L.get().avgpool2_nhwc(x.data_ptr(), y.data_ptr(), n, h, w, c, L.F32 if x.dtype == torch.float32 else L.BF16,
                               torch.cuda.current_stream(x.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"gvfi_avgpool2_nhwc failed with code {rc}")
    return y


class AlternateCorrBlock:
    """raft/corr.py:96-124 on NHWC maps: lookup of a 4-level pyramid without materialising the all-pairs volume
    (memory per direction: the fmap2 pyramid instead of P8^2 floats)."""

    def __init__(self, fmap1_nhwc, fmap2_nhwc, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        self.fmap1 = fmap1_nhwc.contiguous()
        self.pyramid2 = [fmap2_nhwc.contiguous()]
        for _ in range(num_levels - 1):
            self.pyramid2.append(avgpool2_nhwc(self.pyramid2[-1]))

    def __call__(self, coords_nchw):
        coords = coords_nchw.permute(0, 2, 3, 1)
        B, H, W, _ = coords.shape
        dim = self.fmap1.shape[-1]
        out = []
        for i in range(self.num_levels):
            ci = (coords / 2**i).reshape(B, 1, H, W, 2).contiguous()
            (corr,) = forward(self.fmap1, self.pyramid2[i], ci, self.radius)
            out.append(corr.squeeze(1))
        corr = torch.stack(out, dim=1).reshape(B, -1, H, W)
        return corr / math.sqrt(dim)
