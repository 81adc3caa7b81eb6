"""debug: replicate the hazard test's failing sequence, then localise with eager stage taps"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch
from gimmvfi_hip.model import GIMMVFI_R
from gimmvfi_hip.ops import ConvLayer, View
from gimmvfi_hip.params import random_state_dict
from gimmvfi_hip.synth import synthetic_pairs

DEV = "cuda:0"
B, H, W = 8, 256, 448
sd = random_state_dict(0)
x = synthetic_pairs(B, H, W, 3).to(DEV)


def build():
    m = GIMMVFI_R(precision="bf16")
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def run(m):
    coords = [(m.sample_coord_input(B, (H, W), [0.5], device=DEV), None)]
    o = m(x, coords, t=[0.5 * torch.ones(B, device=DEV)])
    torch.cuda.synchronize()
    return [torch.stack([f.float() for f in o["imgt_pred"]]).clone(), o["raft_flow"].float().clone(), torch.stack([f.float() for f in o["flowt"]]).clone()]


if os.environ.get("WITH_SERIAL", "1") == "1":
    s = build()
    ref0 = run(s)
    del s
m = build()
eng = m.engine(DEV)
rt = eng.rt
keep = {}
LO, HI = int(os.environ.get("SNAP_LO", 0)), int(os.environ.get("SNAP_HI", 10000))
orig_syn = eng._synthesize


def syn(B_, sb_, H_, W_, Hf, Wf, img4, img4_full, flow_t, tv, up8, up4, i0q, i1q, pyr, pyrT, taps, ti0, want_aux):
    for k, v in (("flow_t", flow_t), ("up8", up8), ("up4", up4), ("i0q", i0q), ("i1q", i1q), ("img4", img4)):
        keep["in." + k] = (v.t if isinstance(v, View) else v).clone()
    for i, v in enumerate(pyr):
        keep[f"in.pyr{i}"] = v.clone()
    for i, v in enumerate(pyrT):
        keep[f"in.pyrT{i}"] = v.clone()
    cnt = [0]

    def wrap(name, fn, outpos):
        def f(*a, **k):
            r = fn(*a, **k)
            o = k.get("out") if outpos is None else (a[outpos] if len(a) > outpos else None)
            if o is None:
                o = r
            if o is not None and LO <= cnt[0] < HI:
                if isinstance(o, View):
                    c = o.c
                    if name == "conv" and a[0] is not None:
                        c = min(c, a[0].cout)
                    t = o.t[..., o.coff:o.coff + c]
                else:
                    t = o
                keep[f"s{cnt[0]:03d}.{name}"] = t.clone()
                if os.environ.get("SNAP_LIST"):
                    print(f"op {cnt[0]} {name} out {tuple(t.shape)} {t.dtype}")
            cnt[0] += 1
            return r
        return f

    saved = {n: getattr(rt, n) for n in ("conv", "warp", "resize", "corr_lookup", "copy")}
    rt.conv, rt.warp, rt.resize = wrap("conv", saved["conv"], 2), wrap("warp", saved["warp"], 3), wrap("resize", saved["resize"], None)
    rt.corr_lookup, rt.copy = wrap("corr_lookup", saved["corr_lookup"], 2), wrap("copy", saved["copy"], 1)
    try:
        if LO >= HI:
            print("ops in synthesis so far:", cnt[0])
        return orig_syn(B_, sb_, H_, W_, Hf, Wf, img4, img4_full, flow_t, tv, up8, up4, i0q, i1q, pyr, pyrT, taps, ti0, want_aux)
    finally:
        for n, f in saved.items():
            setattr(rt, n, f)


if os.environ.get("SNAP"):
    eng._synthesize = syn
ref = run(m)
for _ in range(int(os.environ.get("PRE", "2"))):
    assert all(torch.equal(a, b) for a, b in zip(run(m), ref))
g = torch.Generator().manual_seed(0)
lay1 = ConvLayer(rt, torch.randn(256, 256, 1, 1, generator=g) / 16, torch.randn(256, generator=g))
lay64 = ConvLayer(rt, torch.randn(64, 64, 3, 3, generator=g) / 24, torch.randn(64, generator=g))
px, py = torch.randn(2, 272, 512, 256, device=DEV).to(rt.tdtype), rt.act(2, 272, 512, 256)
qx, qy = torch.randn(2, 544, 1024, 64, device=DEV).to(rt.tdtype), rt.act(2, 544, 1024, 64)
sb = torch.cuda.Stream()


def partner():
    torch.cuda.synchronize()
    with torch.cuda.stream(sb):
        for i in range(120):
            if i & 1:
                rt.conv(lay1, View(px, 0, 256), py, algo=2, tile=128)
            else:
                rt.conv(lay64, View(qx, 0, 64), qy)


for rep in range(3):
    partner()
    if rep == 0 and keep:
        snap_ref = {k: v.float().cpu().clone() for k, v in keep.items()}      # (of the last solo replay)
    got = run(m)
    print(f"graph replay beside partner {rep}: differing values", [int((a != b).sum()) for a, b in zip(got, ref)])
    if keep:
        for k, v in keep.items():
            d = v.float().cpu() != snap_ref[k]
            if d.any():
                ch = d.reshape(-1, d.shape[-1]).any(0).nonzero().flatten().tolist()
                print(f"   snapshot {k} shape {tuple(v.shape)} differs in {int(d.sum())} values; last-dim indices {ch[:6]}..{ch[-3:]} ({len(ch)})")
coords = [(m.sample_coord_input(B, (H, W), [0.5], device=DEV), None)]
tt = [0.5 * torch.ones(B, device=DEV)]


def tapped(with_partner):
    if with_partner:
        partner()
    else:
        torch.cuda.synchronize()
    taps = {}
    eng.forward(x, coords, tt, iters=20, taps=taps)
    torch.cuda.synchronize()
    return {k: (v.t if isinstance(v, View) else v).float().clone() for k, v in taps.items() if isinstance(v, (torch.Tensor, View))}


solo = tapped(False)
print("eager taps:", len(solo), "solo reproduces:", all(torch.equal(solo[k], v) for k, v in tapped(False).items()))
for rep in range(3):
    got = tapped(True)
    print(f"eager beside partner {rep}:", [(k, int((got[k] != solo[k]).sum())) for k in solo if not torch.equal(got[k], solo[k])][:10])
