"""Host glue of reference src/utils/utils.py:146-185: seeding and the /divisor replicate padder."""
import random

import numpy as np
import torch
import torch.nn.functional as F


def set_seed(seed=None):
    if seed is None:
        seed = random.getrandbits(32)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    return seed


class InputPadder:
    """Replicate-pads the last two dims up to a multiple of `divisor`, centred (utils.py:159-168)."""

    def __init__(self, dims, divisor=16):
        self.ht, self.wd = dims[-2:]
        ph = (-self.ht) % divisor
        pw = (-self.wd) % divisor
        self._pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]

    def pad(self, *inputs):
        out = [F.pad(x, self._pad, mode="replicate") for x in inputs]
        return out[0] if len(out) == 1 else out

    def unpad(self, *inputs):
        out = [self._unpad(x) for x in inputs]
        return out[0] if len(out) == 1 else out

    def _unpad(self, x):
        ht, wd = x.shape[-2:]
        l, r, t, b = self._pad
        return x[..., t: ht - b, l: wd - r]
