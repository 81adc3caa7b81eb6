"""Optical-flow colour coding (Middlebury wheel, Baker et al. ICCV 2007) for the CLI's flow.mp4 side
output -- same mapping as reference src/utils/flow_viz.py:20-136 (`flow_to_image`), written vectorised."""
import numpy as np

_SEGMENTS = (("RY", 15), ("YG", 6), ("GC", 4), ("CB", 11), ("BM", 13), ("MR", 6))


def make_colorwheel():
    n = sum(k for _, k in _SEGMENTS)
    wheel = np.zeros((n, 3))
    # each segment ramps one channel up or down while another stays at 255
    plan = ((0, 1, +1), (1, 0, -1), (1, 2, +1), (2, 1, -1), (2, 0, +1), (0, 2, -1))  # (const ch, ramp ch, dir)
    pos = 0
    for (_, k), (cc, rc, d) in zip(_SEGMENTS, plan):
        ramp = np.floor(255 * np.arange(k) / k)
        wheel[pos:pos + k, cc] = 255
        wheel[pos:pos + k, rc] = ramp if d > 0 else 255 - ramp
        pos += k
    return wheel


def flow_uv_to_colors(u, v, convert_to_bgr=False):
    wheel = make_colorwheel()
    n = wheel.shape[0]
    rad = np.sqrt(u * u + v * v)
    fk = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (n - 1)
    k0 = np.floor(fk).astype(np.int32)
    k1 = (k0 + 1) % n
    f = (fk - k0)[..., None]
    col = (1 - f) * wheel[k0] / 255.0 + f * wheel[k1] / 255.0
    small = (rad <= 1)[..., None]
    col = np.where(small, 1 - rad[..., None] * (1 - col), col * 0.75)
    img = np.floor(255 * col).astype(np.uint8)
    return img[..., ::-1] if convert_to_bgr else img


def flow_to_image(flow_uv, clip_flow=None, convert_to_bgr=False, max_flow=None):
    assert flow_uv.ndim == 3 and flow_uv.shape[2] == 2, "input flow must have shape [H,W,2]"
    if clip_flow is not None:
        flow_uv = np.clip(flow_uv, 0, clip_flow)
    u, v = flow_uv[:, :, 0], flow_uv[:, :, 1]
    rad_max = np.max(np.sqrt(u * u + v * v)) if max_flow is None else max_flow
    eps = 1e-5
    return flow_uv_to_colors(u / (rad_max + eps), v / (rad_max + eps), convert_to_bgr)
