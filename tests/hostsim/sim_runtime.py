"""TEST INFRASTRUCTURE ONLY.

``SimRuntime`` drives gimm-vfi_amd/gimmvfi_hip/engine.py on the CPU so that the engine's
orchestration (buffer layouts, channel offsets, BN folding, weight splitting, launch
arguments) and every glue kernel can be checked against the oracle without a GPU:

* all non-convolution entry points run in the host emulator build of the real kernels
  (tests/hostsim/_build/libgimmvfi_hostsim.so);
* convolutions run either in the emulator too (``emulate_conv=True``: the kernel unit cases and the
  whole-model emulation tests -- half a minute to two minutes per forward at 128 x 192) or through an
  independent torch statement of the launch arguments (``_torch_conv``): a second, independent check
  of the launch lists, and seconds per forward.

Nothing here is importable from the product package.
"""
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))

from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import Runtime, V  # noqa: E402

_LIB = None


def hostsim_lib():
    global _LIB
    if _LIB is None:
        sys.path.insert(0, HERE)
        import build as hostsim_build

        _LIB = L.HipLib(hostsim_build.build())
    return _LIB


def _act(v, kind, slope):
    if kind == L.ACT_NONE:
        return v
    if kind == L.ACT_RELU:
        return F.relu(v)
    if kind == L.ACT_LRELU:
        return F.leaky_relu(v, 0.1)
    if kind == L.ACT_PRELU:
        return torch.where(v > 0, v, v * slope.view(1, 1, 1, -1))
    if kind == L.ACT_SIGMOID:
        return torch.sigmoid(v)
    if kind == L.ACT_TANH:
        return torch.tanh(v)
    if kind == L.ACT_SIN:
        return torch.sin(v)
    if kind == L.ACT_GELU:
        return F.gelu(v)
    raise ValueError(kind)


class SimRuntime(Runtime):
    def __init__(self, precision, emulate_conv=False):
        super().__init__(hostsim_lib(), precision, "cpu")
        self.emulate_conv = emulate_conv

    def sibling(self, precision):
        return SimRuntime(precision, self.emulate_conv)

    def gru_half(self, *a, **kw):
        if not self.emulate_conv:
            return False     # (the torch statement of the two gate convolutions runs instead)
        return super().gru_half(*a, **kw)

    def conv_pair(self, a, b):
        if self.emulate_conv:
            return super().conv_pair(a, b)
        self.conv(**a)       # the torch statement of a launch has no notion of a shared grid
        self.conv(**b)

    def conv(self, layer, x0, out, x1=None, act1=L.ACT_NONE, res=None, act2=L.ACT_NONE, out_scale=1.0,
             slope1=None, slope2=None, epi=L.EPI_STD, y2=None, aux0=None, aux1=None, groups=1,
             w_group_stride=0, w_raw=None, cout=None, tile=0, algo=0, stats=None, pad16=False, state_f32=False, planar3=None):
        if self.emulate_conv:
            return super().conv(layer, x0, out, x1, act1, res, act2, out_scale, slope1, slope2, epi, y2, aux0, aux1,
                                groups, w_group_stride, w_raw, cout, tile, algo, stats, pad16, state_f32, planar3)
        self.last_stats_fused = False     # the torch statement leaves the statistics to gvfi_instnorm_stats
        self.last_planar = False          # ... and the planar finalisation to gvfi_finalize_image
        return self._torch_conv(layer, V(x0), V(out), None if x1 is None else V(x1), act1, res, act2, out_scale,
                                slope1, slope2, epi, y2, aux0, aux1, groups, w_raw, cout)

    def inr_mlp(self, mlp, lat, coord, out):
        """Fused hypo-network: emulated kernel for unit shapes, else an independent torch statement of the same
        arithmetic (bf16-rounded weights and hidden activations, fp32 accumulation) from the un-packed layers."""
        if self.emulate_conv:
            return super().inr_mlp(mlp, lat, coord, out)
        lat = V(lat)
        bf = lambda t: t.to(torch.bfloat16).float()
        h = torch.cat([self._sl(lat, 32), bf(coord[:, 0])], -1).reshape(-1, 35)
        for li, (w, b) in enumerate(mlp.layers):
            h = h @ bf(w.float().cpu()).t() + b.float().cpu()
            if li < 4:
                h = bf(torch.sin(h))
        out.copy_(h.reshape(out.shape))
        return out

    def _sl(self, v, c=None):
        c = v.c if c is None else c
        return v.t[..., v.coff:v.coff + c].float()

    def _torch_conv(self, layer, x0, out, x1, act1, res, act2, out_scale, slope1, slope2, epi, y2, aux0, aux1, groups,
                    w_raw, cout):
        xs = [self._sl(x0, self.cp(x0.c))]
        if x1 is not None:
            xs.append(self._sl(x1, self.cp(x1.c)))
        x = torch.cat(xs, -1)
        n = x.shape[0]
        if layer is None:
            # grouped "weights are features" GEMM: out[g, p, q] = sum_k x[g,p,k] * w[g,q,k]
            k = x.shape[-1]
            wf = w_raw.reshape(groups, -1, w_raw.shape[-1])[..., :k].float()
            xf = x.reshape(groups, -1, k)
            v = torch.einsum("gpk,gqk->gpq", xf, wf).reshape(n, x.shape[1], x.shape[2], cout)
        else:
            w = layer.w.float().permute(0, 3, 1, 2)  # [Cout, cin_pad, KH, KW]
            xi = x.permute(0, 3, 1, 2)
            ph, pw = layer.pad
            if layer.pad_mode == L.PAD_REFLECT:
                xi = F.pad(xi, (pw, pw, ph, ph), mode="reflect")
                ph = pw = 0
            v = F.conv2d(xi, w, layer.b, stride=layer.stride, padding=(ph, pw)).permute(0, 2, 3, 1)
            cout = layer.cout
        if epi == L.EPI_STD:
            s1 = slope1 if slope1 is not None else (layer.slope if layer is not None else None)
            v = _act(v, act1, s1)
            if res is not None:
                v = v + self._sl(V(res), cout)
            v = _act(v, act2, slope2) * out_scale
            out.t[..., out.coff:out.coff + cout] = v.to(out.t.dtype)
        elif epi == L.EPI_GRU_ZR:
            if res is not None:
                v = v + self._sl(V(res), cout)
            s = torch.sigmoid(v)
            half = cout // 2
            out.t[..., out.coff:out.coff + half] = s[..., :half].to(out.t.dtype)
            y2 = V(y2)
            y2.t[..., y2.coff:y2.coff + half] = (s[..., half:] * self._sl(V(aux0), half)).to(y2.t.dtype)
        else:
            if res is not None:
                v = v + self._sl(V(res), cout)
            q = torch.tanh(v)
            h = self._sl(V(aux0), cout)
            z = self._sl(V(aux1), cout)
            hn = (1 - z) * h + z * q
            out.t[..., out.coff:out.coff + cout] = hn.to(out.t.dtype)
            if y2 is not None:      # float state beside its operand copy (state_f32; the tensors' dtypes say the rest)
                y2 = V(y2)
                y2.t[..., y2.coff:y2.coff + cout] = hn.to(y2.t.dtype)
        return out
