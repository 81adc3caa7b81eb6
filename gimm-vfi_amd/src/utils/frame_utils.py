"""Middlebury .flo optical-flow files (what the reference's utils/frame_utils.py:24-43 `readFlow` reads for the GIMM
flow benchmarks).  Format: float32 magic 202021.25, int32 width, int32 height, then height*width*(u, v) float32,
row-major, little endian."""
import numpy as np

FLO_MAGIC = 202021.25


def readFlow(fn):
    """-> float32 array (H, W, 2), or None when the magic number is wrong (the reference's behaviour)."""
    with open(fn, "rb") as f:
        head = np.frombuffer(f.read(12), dtype=np.dtype("<f4, <i4, <i4"))[0]
        if float(head[0]) != FLO_MAGIC:
            print("Magic number incorrect. Invalid .flo file")
            return None
        w, h = int(head[1]), int(head[2])
        data = np.frombuffer(f.read(8 * w * h), dtype="<f4")
    return np.array(data, dtype=np.float32).reshape(h, w, 2)


def writeFlow(fn, uv):
    """uv: (H, W, 2) float array."""
    uv = np.asarray(uv, dtype="<f4")
    h, w = uv.shape[:2]
    with open(fn, "wb") as f:
        np.array([FLO_MAGIC], dtype="<f4").tofile(f)
        np.array([w, h], dtype="<i4").tofile(f)
        uv.tofile(f)
