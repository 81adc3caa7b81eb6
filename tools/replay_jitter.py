"""How much do two replays of the same captured forward differ?  (The InstanceNorm statistics of RAFT's feature encoder are float
atomics: their order is the one run-to-run freedom the path has.)  Prints max |d| of every output over N replays, lanes on / off."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gimmvfi_hip.model import GIMMVFI_F, GIMMVFI_R  # noqa: E402
from gimmvfi_hip.params import random_state_dict, random_state_dict_f  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402

DEV = "cuda:0"
MODEL = os.environ.get("JITTER_MODEL", "r")
sd = random_state_dict_f(0) if MODEL == "f" else random_state_dict(0)
B, H, W, ds, T = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (8, 256, 448, 1.0, 1)
N = int(sys.argv[6]) if len(sys.argv) > 6 else 10
x = synthetic_pairs(B, H, W, 3).to(DEV)
ts = [(i + 1) / (T + 1) for i in range(T)]


ONLY = os.environ.get("JITTER_ONLY")      # e.g. "SYNTH": only that switch on in the "lanes" run (bisection)


def run(lanes):
    for k in ("GVFI_ENC_LANES", "GVFI_POST_LANES", "GVFI_SYNTH_LANES"):
        os.environ[k] = "1" if (lanes and (ONLY is None or ONLY in k)) else "0"
    os.environ["GVFI_RAFT_LANES"] = "2" if (lanes and (ONLY is None or ONLY == "RAFT")) else "1"
    os.environ["GVFI_F_LANES"] = os.environ["GVFI_RAFT_LANES"]
    m = (GIMMVFI_F if MODEL == "f" else GIMMVFI_R)(precision="bf16")
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    hs, ws = int(H * ds), int(W * ds)
    coords = [(m.sample_coord_input(B, (H, W), [t], device=DEV, upsample_ratio=ds), None) for t in ts]
    tt = [t * torch.ones(B, device=DEV) for t in ts]
    outs = []
    for _ in range(N):
        o = m(x, coords, t=tt, ds_factor=None if ds == 1.0 else ds)
        torch.cuda.synchronize()
        outs.append([(f.clamp(0, 1) * 255).round().to(torch.uint8).cpu() for f in o["imgt_pred"]] + [o["raft_flow"].float().cpu()]
                    + [torch.stack([f.float().cpu() for f in o["flowt"]]), torch.stack([f.float().cpu() for f in o["imgt_pred"]]),
                       torch.stack([f[0].float().cpu() for f in o["flowt0_pred"]]) if isinstance(o["flowt0_pred"][0], (list, tuple)) else None])
    return outs


a = run(True)
b = run(False)
for name, runs in (("lanes", a), ("serial", b)):
    for i in range(1, N):
        dimg = max(int((runs[i][k].int() - runs[0][k].int()).abs().max()) for k in range(T))
        npix = sum(int((runs[i][k] != runs[0][k]).sum()) for k in range(T))
        dfl = float((runs[i][T] - runs[0][T]).abs().max())
        dinr = float((runs[i][T + 1] - runs[0][T + 1]).abs().max())
        dpred = (runs[i][T + 2] - runs[0][T + 2]).abs()
        per_t = [float(dpred[k].max()) for k in range(T)]
        print(f"{name} replay {i} vs 0: frames max |d| {dimg} LSB in {npix} values, raft flow max |d| {dfl:.3e}, INR flow max |d| {dinr:.3e}, "
              f"float frames max |d| per t {[f'{v:.1e}' for v in per_t]}")
dimg = max(int((a[0][k].int() - b[0][k].int()).abs().max()) for k in range(T))
npix = sum(int((a[0][k] != b[0][k]).sum()) for k in range(T))
print(f"lanes vs serial: frames max |d| {dimg} LSB in {npix} values, raft flow max |d| {float((a[0][T] - b[0][T]).abs().max()):.3e}")
