"""TEST INFRASTRUCTURE (runnable helper, not collected by pytest): one forward of GIMM-VFI-R with EVERY launch on the host
emulator build of the real kernels (tests/hostsim), against the CPU oracle run live on the same seeded input -- the GPU suite's
`*_vs_live_oracle` cases without a GPU, at any size.
usage: python tests/emulate_whole_model.py [H] [W] [precision] [t]        (default 256 448 bf16 0.5: the bench frame size, one pair)"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "hostsim")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import gimmvfi_r_oracle as orc  # noqa: E402
from gimmvfi_hip.engine import Engine  # noqa: E402
from gimmvfi_hip.params import random_state_dict  # noqa: E402
from gimmvfi_hip.synth import synthetic_pairs  # noqa: E402
from sim_runtime import SimRuntime  # noqa: E402
from util import maxabs, psnr  # noqa: E402


def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 448
    prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
    t = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
    sd = random_state_dict(0)
    x = synthetic_pairs(1, H, W, seed=0)
    coords = [(orc.sample_coord_input(1, (H, W), [t], 1.0), None)]
    ts = [t * torch.ones(1)]
    t0 = time.time()
    with torch.no_grad():
        ref = orc.forward(sd, x, coords, ts, 1.0)
    t1 = time.time()
    rt = SimRuntime(prec, emulate_conv=True)
    rt.lib.dll.gvfi_emu_set_dma_mode(1)
    rt.lib.dll.gvfi_emu_set_sched(3)
    out = Engine(rt, sd).forward(x, coords, ts)
    t2 = time.time()
    d = (out["flowt"][0].float() - ref["flowt"][0].float()).abs().flatten()
    print(f"GIMM-VFI-R {W}x{H} t={t} {prec}: {rt.n_launch} launches on the emulated kernels (adversarial DMA timing, schedule 3) in {t2 - t1:.0f} s "
          f"(oracle {t1 - t0:.0f} s): PSNR(imgt_pred vs live oracle) = {psnr(out['imgt_pred'][0], ref['imgt_pred'][0]):.2f} dB, "
          f"max|raft_flow err| = {maxabs(out['raft_flow'], ref['raft_flow']):.3e}, mean|flowt err| = {float(d.mean()):.3e}")


if __name__ == "__main__":
    main()
