"""Whole forward beside an adversarial partner: while the captured forward replays on one stream, a second stream runs LDS-DMA
convolutions on unrelated tensors, so that every kernel of the forward shares compute units with LDS-DMA waves at some point.
Outputs must equal the solo replay bit for bit (the forward is bit-reproducible since round 6).  Found in round 6: global
loads under a partial EXEC mask (a bilinear tap behind `if (x + 1 < W)`) return wrong data in a few 16-lane groups when an
LDS-DMA kernel runs on the same CU (tools/concurrency_repro.py); this sweep looks for further kernels with that pattern.
usage: python tools/adversarial_partner.py [r|f] B H W ds T [replays]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.model import GIMMVFI_F, GIMMVFI_R  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, View  # noqa: E402
from gimmvfi_hip.params import random_state_dict, random_state_dict_f  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402

DEV = "cuda:0"
mdl = sys.argv[1] if len(sys.argv) > 1 else "r"
B, H, W, ds, T = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (8, 256, 448, 1.0, 1)
N = int(sys.argv[7]) if len(sys.argv) > 7 else 10
m = (GIMMVFI_F if mdl == "f" else GIMMVFI_R)(precision="bf16")
m.load_state_dict(random_state_dict_f(0) if mdl == "f" else random_state_dict(0), strict=True)
m = m.to(DEV).eval()
rt = m.engine(DEV).rt
x = synthetic_pairs(B, H, W, 3).to(DEV)
ts = [(i + 1) / (T + 1) for i in range(T)]
coords = [(m.sample_coord_input(B, (H, W), [t], device=DEV, upsample_ratio=ds), None) for t in ts]
tt = [t * torch.ones(B, device=DEV) for t in ts]
KEYS = ("imgt_pred", "flowt", "flowt0_pred", "flowt1_pred", "other_pred")


def flat(o):
    out = {"raft_flow": o["raft_flow"].float().clone(), "nflow": o["nflow"].float().clone()}
    for k in KEYS:
        for i, v in enumerate(o[k]):
            vs = v if isinstance(v, (list, tuple)) else [v]
            for j, u in enumerate(vs):
                if isinstance(u, torch.Tensor):
                    out[f"{k}[{i}][{j}]"] = u.float().clone()
    return out


def run():
    o = m(x, coords, t=tt, ds_factor=None if ds == 1.0 else ds)
    return flat(o)


ref = run()
torch.cuda.synchronize()
again = run()
torch.cuda.synchronize()
assert all(torch.equal(ref[k], again[k]) for k in ref), "the solo forward does not reproduce itself"
# partners: 4-wave LDS-DMA tiles that leave room for other waves on their CUs
lay1 = ConvLayer(rt, torch.randn(256, 256, 1, 1) / 16, torch.randn(256))
lay64 = ConvLayer(rt, torch.randn(64, 64, 3, 3) / 24, torch.randn(64))
px = torch.randn(2, 272, 512, 256, device=DEV).to(rt.tdtype)
py = rt.act(2, 272, 512, 256)
qx = torch.randn(2, 544, 1024, 64, device=DEV).to(rt.tdtype)
qy = rt.act(2, 544, 1024, 64)
sb = torch.cuda.Stream()
main = torch.cuda.current_stream()
bad = {}
for rep in range(N):
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        for i in range(int(os.environ.get("NPART", "400" if H * W > 500000 else "150"))):
            if i & 1:
                rt.conv(lay1, View(px, 0, 256), py, algo=2, tile=128)
            else:
                rt.conv(lay64, View(qx, 0, 64), qy)
    got = run()
    torch.cuda.synchronize()
    for k in ref:
        n = int((got[k] != ref[k]).sum())
        if n:
            bad.setdefault(k, []).append((rep, n, float((got[k] - ref[k]).abs().max())))
print(f"{mdl} {B}x{H}x{W} ds {ds} T {T}: {N} replays beside LDS-DMA partners;", "ALL OUTPUTS BIT-IDENTICAL to the solo replay" if not bad else "DIFFERENCES:")
for k, v in bad.items():
    print("  ", k, v[:4])

# ---- localisation: the eager forward with its stage taps, solo against beside-partner runs; first differing tap = the stage
if os.environ.get("TAPS"):
    eng = m.engine(DEV)

    def tapped(with_partner):
        torch.cuda.synchronize()
        if with_partner:
            with torch.cuda.stream(sb):
                for i in range(int(os.environ.get("NPART", "600"))):
                    if i & 1:
                        rt.conv(lay1, View(px, 0, 256), py, algo=2, tile=128)
                    else:
                        rt.conv(lay64, View(qx, 0, 64), qy)
        taps = {}
        eng.forward(x, coords, tt, iters=20, ds_factor=None if ds == 1.0 else ds, taps=taps)
        torch.cuda.synchronize()
        return {k: (v.t if isinstance(v, View) else v).float().clone() for k, v in taps.items() if isinstance(v, (torch.Tensor, View))}

    solo = tapped(False)
    solo2 = tapped(False)
    print("taps:", len(solo), "solo reproduces itself:", all(torch.equal(solo[k], solo2[k]) for k in solo))
    for rep in range(4):
        got = tapped(True)
        diff = [(k, int((got[k] != solo[k]).sum())) for k in solo if got[k].shape == solo[k].shape and not torch.equal(got[k], solo[k])]
        print(f"rep {rep}: differing taps (in forward order):", diff[:12])
