"""Where does GIMM-VFI-F's bf16 mode leave its fp32 mode?  Stage taps of both HIP engines on the same input (GPU).
usage: python tools/f_bf16_diag.py [H W seed]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))
import torch  # noqa: E402

from gimmvfi_hip.model import GIMMVFI_F  # noqa: E402
from gimmvfi_hip.params import random_state_dict_f  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402

H, W, seed = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 448, 100)
sd = random_state_dict_f(0)
x = synthetic_pairs(1, H, W, seed, max_disp=float(os.environ.get("DISP", "8"))).cuda()
taps = {}
for prec in ("fp32", "bf16"):
    m = GIMMVFI_F(precision=prec)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    c = [(m.sample_coord_input(1, (H, W), [0.5], device="cuda"), None)]
    t = {}
    out = m.engine("cuda").forward(x, c, [0.5 * torch.ones(1, device="cuda")], iters=None, taps=t)
    torch.cuda.synchronize()
    t["flow_up01"] = out["raft_flow"][:, :, 0]
    t["imgt_pred"] = out["imgt_pred"][0]
    taps[prec] = {k: v.float().cpu() for k, v in t.items() if torch.is_tensor(v)}
    del m
    torch.cuda.empty_cache()
print(f"{'tap':28s} {'max|fp32|':>10s} {'rel max err':>12s} {'rel rms err':>12s}")
for k, a in taps["fp32"].items():
    b = taps["bf16"].get(k)
    if b is None or b.shape != a.shape:
        continue
    s = float(a.abs().max()) + 1e-12
    print(f"{k:28s} {s:10.3g} {float((a - b).abs().max()) / s:12.3e} {float((a - b).pow(2).mean().sqrt()) / (float(a.pow(2).mean().sqrt()) + 1e-12):12.3e}")
