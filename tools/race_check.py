"""Cold-vs-warm determinism check of the motion path (GIMM) on the GPU: in a FRESH process run the fp32 model once,
then the bf16 engine stage by stage (first launch of every bf16 kernel), then the same again warm; stages that are
deterministic by construction (everything before the splat's fp32 atomics) must be bit-identical, the rest within
1e-2.  Prints one line per mismatch; meant to be looped from the shell (`for i in $(seq 20); do ...`).
Test infrastructure (uses tests/util.py for the seeded inputs)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402

from util import gimm_inputs, load_golden  # noqa: E402

DEV = "cuda:0"


def stages(eng, xs, ori, coord, ts):
    """forward_motion of engine.py with every intermediate kept."""
    rt = eng.rt
    xs = xs.to(device=rt.device, dtype=torch.float32)
    ori = ori.to(device=rt.device, dtype=torch.float32)
    B, _, _, H, W = xs.shape
    f01 = ori[:, :, 0].permute(0, 2, 3, 1).contiguous()
    f10 = ori[:, :, 1].permute(0, 2, 3, 1).contiguous()
    nfA = rt.act(2 * B, H, W, 2, zero=True)
    nfA[:B, ..., :2] = xs[:, :, 0].permute(0, 2, 3, 1).to(nfA.dtype)
    nfA[B:, ..., :2] = xs[:, :, 1].permute(0, 2, 3, 1).to(nfA.dtype)
    z0, z1, latcat = eng._motion_encode(nfA, f01, f10, B, H, W)
    out = {"z0": z0.clone(), "z1": z1.clone(), "enc": latcat[..., :32].clone()}
    cg = coord.to(device=rt.device, dtype=torch.float32).contiguous()
    tv = ts.to(device=rt.device, dtype=torch.float32).reshape(-1).contiguous()
    if tv.numel() == 1 and B > 1:
        tv = tv.expand(B).contiguous()
    taps = {}
    ninr = eng._motion_inr(latcat, f01, f10, z0, z1, cg, tv, B, H, W, taps=taps)
    out["splat"] = latcat[..., 32:].clone()
    out["latent"] = taps["latent"].clone()
    out["ninr"] = ninr.clone()
    torch.cuda.synchronize()
    return out


def main():
    from gimmvfi_hip.model import GIMM
    from gimmvfi_hip.params import gimm_state_dict, random_state_dict

    sd = gimm_state_dict(random_state_dict(0))
    meta, gold = load_golden("gimm_b2_96x160_t050")
    xs, ori, coord, ts = gimm_inputs(meta)
    m32 = GIMM(precision="fp32")
    m32.load_state_dict(sd, strict=True)
    m32 = m32.to(DEV).eval()
    m32(xs.cuda(), coord.cuda(), ori_flow=ori.cuda(), timesteps=ts.cuda())
    m = GIMM(precision="bf16")
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    eng = m.engine(torch.device(DEV))
    cold = stages(eng, xs, ori, coord, ts)
    warm = [stages(eng, xs, ori, coord, ts) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8)]
    ref = warm[-1]
    bad = 0
    for tag, run in [("cold", cold)] + [(f"warm{i}", w) for i, w in enumerate(warm[:-1])]:
        for k in ("z0", "z1", "enc", "splat", "latent", "ninr"):
            a, b = run[k].float(), ref[k].float()
            exact = k in ("z0", "z1", "enc")
            d = (a - b).abs()
            nbad = int((d > (0.0 if exact else 1e-2)).sum()) + int(torch.isnan(d).sum())
            if nbad:
                bad += 1
                idx = (d > (0.0 if exact else 1e-2)).nonzero()[:4].tolist()
                print(f"MISMATCH {tag} {k}: {nbad} elements, max {float(d.max()):.3e}, first {idx}", flush=True)
    dg = (ref["ninr"].permute(0, 3, 1, 2).unsqueeze(2).cpu() - gold["out_0"]).abs()
    print(f"RACE_BAD {bad}  (ref vs golden mean {float(dg.mean()):.2e})")


if __name__ == "__main__":
    main()
