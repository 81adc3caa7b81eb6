"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's native correlation op `alt_cuda_corr`.

`alt_corr_forward` follows flowformer/alt_cuda_corr/correlation_kernel.cu:18-119 (corr_forward_kernel) and the host
wrapper :255-286: for every query pixel the (2r+2)^2 dot products with fmap2 at the integer taps around
floor(coords) - r (zero outside fmap2, :71-77) are spread with bilinear weights dy/dx to the (2r+1)^2 outputs
(:88-109), output channel = iy + rd*ix (:88-91).

Pinning: the CUDA op cannot be compiled here (no nvcc) and the reference ships no vectors for it; the reference itself
documents it as equivalent to `CorrBlock` (raft/corr.py:96-124 vs :127-165), whose restatement in
gimmvfi_r_oracle.py IS pinned bit-exact against the reference.  tests/test_alt_corr.py checks this restatement against
that pinned lookup (same pyramid levels, same window order) up to floating-point order.
"""
import math

import torch
import torch.nn.functional as F


def alt_corr_forward(fmap1, fmap2, coords, r):
    """fmap1 (B,H1,W1,C), fmap2 (B,H2,W2,C), coords (B,N,H1,W1,2) -> corr (B,N,(2r+1)^2,H1,W1)."""
    B, H1, W1, C = fmap1.shape
    H2, W2 = fmap2.shape[1:3]
    N = coords.shape[1]
    rd = 2 * r + 1
    x2, y2 = coords[..., 0], coords[..., 1]                       # (B,N,H1,W1)
    fx, fy = torch.floor(x2), torch.floor(y2)
    dx, dy = x2 - fx, y2 - fy                                     # correlation_kernel.cu:64-65
    f2p = fmap2
    corr = torch.zeros(B, N, rd * rd, H1, W1, dtype=fmap1.dtype)
    bidx = torch.arange(B).view(B, 1, 1, 1).expand(B, N, H1, W1)
    f1 = fmap1.unsqueeze(1).expand(B, N, H1, W1, C)
    s = torch.zeros(B, N, rd + 1, rd + 1, H1, W1, dtype=fmap1.dtype)
    for iy in range(rd + 1):
        for ix in range(rd + 1):
            h2 = fy.long() - r + iy                               # :71-72
            w2 = fx.long() - r + ix
            ok = (h2 >= 0) & (h2 < H2) & (w2 >= 0) & (w2 < W2)     # within_bounds, :13-16
            g = f2p[bidx, h2.clamp(0, H2 - 1), w2.clamp(0, W2 - 1)]   # (B,N,H1,W1,C)
            s[:, :, iy, ix] = (f1 * g).sum(-1) * ok
    for a in range(rd):          # iy index of the output
        for c in range(rd):      # ix index of the output; channel = a + rd*c  (:88-91: ix_se = iy + rd*ix)
            corr[:, :, a + rd * c] = ((1 - dy) * (1 - dx) * s[:, :, a, c] + (1 - dy) * dx * s[:, :, a, c + 1]
                                      + dy * (1 - dx) * s[:, :, a + 1, c] + dy * dx * s[:, :, a + 1, c + 1])
    return corr


def alternate_corr_block(fmap1_nchw, fmap2_nchw, coords_nchw, num_levels=4, r=4):
    """raft/corr.py:96-124 (AlternateCorrBlock) on NCHW maps -> (B, levels*(2r+1)^2, H, W)."""
    pyr = [(fmap1_nchw, fmap2_nchw)]
    f1, f2 = fmap1_nchw, fmap2_nchw
    for _ in range(num_levels - 1):   # (the reference pools once more than it uses, raft/corr.py:102-105)
        f1 = F.avg_pool2d(f1, 2, stride=2)
        f2 = F.avg_pool2d(f2, 2, stride=2)
        pyr.append((f1, f2))
    coords = coords_nchw.permute(0, 2, 3, 1)
    B, H, W, _ = coords.shape
    dim = fmap1_nchw.shape[1]
    out = []
    for i in range(num_levels):
        a = pyr[0][0].permute(0, 2, 3, 1).contiguous()
        b = pyr[i][1].permute(0, 2, 3, 1).contiguous()
        ci = (coords / 2**i).reshape(B, 1, H, W, 2).contiguous()
        out.append(alt_corr_forward(a, b, ci, r).squeeze(1))
    return torch.stack(out, 1).reshape(B, -1, H, W) / math.sqrt(dim)
