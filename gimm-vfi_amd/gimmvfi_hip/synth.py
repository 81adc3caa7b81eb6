"""Seeded synthetic frame pairs for benchmarks and parity tests.

I0 = band-limited RGB noise (uniform noise, blurred, rescaled to [0,1], quantised to uint8/255 like
reference src/video_Nx.py:46-50 load_image); I1 = I0 warped by a smooth seeded displacement field
(|d| <= max_disp px) so the flow estimator and the splat see non-trivial, finite motion.
Data generation only -- CPU torch, not part of the timed path."""
import torch
import torch.nn.functional as F


def _blur(x, k, times):
    for _ in range(times):
        x = F.avg_pool2d(F.pad(x, (k // 2,) * 4, mode="reflect"), k, 1)
    return x


def synthetic_pairs(B, H, W, seed=0, max_disp=8.0):
    g = torch.Generator(device="cpu")
    out = torch.empty(B, 3, 2, H, W)
    for b in range(B):
        g.manual_seed(seed * 1000 + b)
        n = torch.rand(1, 3, H, W, generator=g)
        i0 = _blur(n, 7, 2)
        i0 = (i0 - i0.amin()) / (i0.amax() - i0.amin() + 1e-12)
        d = torch.rand(1, 2, max(H // 32, 2), max(W // 32, 2), generator=g) * 2 - 1
        d = F.interpolate(d, size=(H, W), mode="bicubic", align_corners=True).clamp(-1, 1) * max_disp
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        gx = (xs[None] + d[:, 0]) / (W - 1) * 2 - 1
        gy = (ys[None] + d[:, 1]) / (H - 1) * 2 - 1
        i1 = F.grid_sample(i0, torch.stack([gx, gy], -1), mode="bilinear", padding_mode="border", align_corners=True)
        out[b, :, 0] = i0[0]
        out[b, :, 1] = i1[0]
    return torch.round(out.clamp(0, 1) * 255.0) / 255.0
