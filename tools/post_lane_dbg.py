"""debug: which intermediate of the main launch sequence differs between replays when the post-recurrence side sequence runs
beside it (GVFI_POST_LANES)?  Wraps Engine._motion_encode / _motion_inr and keeps clones of their results."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

for k in ("GVFI_ENC_LANES", "GVFI_SYNTH_LANES"):
    os.environ[k] = "0"
os.environ["GVFI_POST_LANES"] = os.environ.get("POST", "1")
os.environ["GVFI_RAFT_LANES"] = "1"
from gimmvfi_hip.model import GIMMVFI_R  # noqa: E402
from gimmvfi_hip.params import random_state_dict  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402

DEV = "cuda:0"
B, H, W, ds, T = 1, 1088, 2048, 0.5, 7
m = GIMMVFI_R(precision="bf16")
m.load_state_dict(random_state_dict(0), strict=True)
m = m.to(DEV).eval()
eng = m.engine(DEV)
keep = {}
orig_enc, orig_inr = eng._motion_encode, eng._motion_inr


def enc(*a, **k):
    keep["f01_before"], keep["f10_before"] = a[1].clone(), a[2].clone()
    # the metric kernel twice more, back to back, in front of the engine's own launch (same stream, same inputs)
    for tag in ("x", "y"):
        za, zb = torch.empty(a[1].shape[:3], device=DEV), torch.empty(a[1].shape[:3], device=DEV)
        eng.rt._chk(eng.rt.lib.splat_weights(a[1].data_ptr(), a[2].data_ptr(), eng.g9.data_ptr(), eng.alpha_v, eng.alpha_fe,
                                             za.data_ptr(), zb.data_ptr(), B, a[1].shape[1], a[1].shape[2], eng.rt.stream()), "splat_weights")
        keep["z0" + tag], keep["z1" + tag] = za, zb
    z0, z1, latcat = orig_enc(*a, **k)
    keep["f01_after"], keep["f10_after"] = a[1].clone(), a[2].clone()
    keep["nfA"] = a[0].clone()
    keep["z0"], keep["z1"], keep["latcat_enc"] = z0.clone(), z1.clone(), latcat.clone()
    return z0, z1, latcat


cnt = [0]


def inr(latcat, *a, **k):
    r = orig_inr(latcat, *a, **k)
    keep[f"ninr_{cnt[0] % T}"] = r.clone()
    keep[f"latcat_after_{cnt[0] % T}"] = latcat.clone()
    cnt[0] += 1
    return r


eng._motion_encode, eng._motion_inr = enc, inr
x = synthetic_pairs(B, H, W, 3).to(DEV)
ts = [(i + 1) / (T + 1) for i in range(T)]
coords = [(m.sample_coord_input(B, (H, W), [t], device=DEV, upsample_ratio=ds), None) for t in ts]
tt = [t * torch.ones(B, device=DEV) for t in ts]
runs = []
for r in range(4):
    o = m(x, coords, t=tt, ds_factor=ds)
    torch.cuda.synchronize()
    # z recomputed (eagerly, alone on the GPU) from the flows the captured kernel was given
    zr0, zr1 = torch.empty_like(keep["z0"]), torch.empty_like(keep["z1"])
    eng.rt._chk(eng.rt.lib.splat_weights(keep["f01_before"].data_ptr(), keep["f10_before"].data_ptr(), eng.g9.data_ptr(), eng.alpha_v,
                                         eng.alpha_fe, zr0.data_ptr(), zr1.data_ptr(), B, keep["z0"].shape[1], keep["z0"].shape[2],
                                         eng.rt.stream()), "splat_weights")
    torch.cuda.synchronize()
    for tag in ("x", "y"):
        nzx = (keep["z0" + tag] != zr0).nonzero()
        print(f"  extra launch {tag}: differs from the recomputation at {nzx.shape[0]} pixels", nzx[:2].tolist())
    keep["z0_minus_recomputed"] = keep["z0"] - zr0
    keep["z1_minus_recomputed"] = keep["z1"] - zr1
    nz = (keep["z0_minus_recomputed"] != 0).nonzero()
    if nz.numel():
        b_, y_, x_ = nz[0].tolist()
        print("  z0 got     ", [f"{v:.5f}" for v in keep["z0"][b_, y_, max(0, x_ - 2):x_ + 18].tolist()])
        print("  recomputed ", [f"{v:.5f}" for v in zr0[b_, y_, max(0, x_ - 2):x_ + 18].tolist()])
        fl = keep["f01_before"][b_, y_, x_].tolist()
        print("  flow there ", fl, "-> warp target", x_ + fl[0], y_ + fl[1])
    if r == 0 and nz.numel():
        print("z0 differs from its recomputation at", nz.shape[0], "pixels; y range", int(nz[:, 1].min()), int(nz[:, 1].max()), "x range",
              int(nz[:, 2].min()), int(nz[:, 2].max()), "first", nz[:5].tolist())
    snap = {k: v.float().cpu().clone() for k, v in keep.items()}
    snap["nflow"] = o["nflow"].float().cpu().clone()
    snap["flowt_0"] = o["flowt"][0].float().cpu().clone()
    runs.append(snap)
for r in range(1, 4):
    print(f"replay {r} vs 0:", {k: f"{float((runs[r][k] - runs[0][k]).abs().max()):.2e}" for k in sorted(runs[0])})
for r in range(4):
    print(f"replay {r}: |z0 - recomputed| max {float(runs[r]['z0_minus_recomputed'].abs().max()):.2e}, |z1 - recomputed| max "
          f"{float(runs[r]['z1_minus_recomputed'].abs().max()):.2e}, f01 before/after encode equal: "
          f"{bool(torch.equal(runs[r]['f01_before'], runs[r]['f01_after']))}")
