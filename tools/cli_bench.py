"""End-to-end throughput of the CLI (PNG decode -> pad -> H2D -> model -> D2H -> colour-coding), synthetic frames.
usage: python tools/cli_bench.py [n_frames] [W] [H] [N] [DS_SCALE] [GPUS] [dry]
With GPUS > 1 the CLI is started under torch.distributed.run on this node (one rank per GPU, free port), as
scripts/video_Nx.sh does.  `dry` (7th argument): GVFI_CLI_DRY=1 -- the multi-GPU result path rehearsed on ONE GPU: rank 0 is
real, ranks 1.. are CPU stand-ins that decode, compose, ENCODE and gather like real ranks (gloo); prints every rank's host
seconds (GVFI_CLI_TIMING).  Never a throughput figure of the model."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
from PIL import Image

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "gimm-vfi_amd"))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
    W = int(sys.argv[2]) if len(sys.argv) > 2 else 448
    H = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    ds = sys.argv[5] if len(sys.argv) > 5 else "1.0"
    gpus = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    dry = len(sys.argv) > 7 and sys.argv[7] == "dry"
    from gimmvfi_hip.synth import synthetic_pairs

    with tempfile.TemporaryDirectory() as d:
        src, out = os.path.join(d, "in"), os.path.join(d, "out")
        os.makedirs(src)
        x = synthetic_pairs(1, H, W, seed=0)[0]      # (3,2,H,W): reuse the two frames alternately with a shift
        for i in range(n):
            f = np.roll((x[:, i % 2].permute(1, 2, 0).numpy() * 255).astype(np.uint8), 3 * i, axis=1)
            Image.fromarray(f).save(os.path.join(src, f"{i:04d}.png"))
        launcher = [sys.executable]
        if gpus > 1:
            import socket

            import torch

            if torch.cuda.device_count() < gpus and not dry:
                sys.exit(f"cli_bench: {gpus} GPUs requested, {torch.cuda.device_count()} visible")
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port)]
        cmd = launcher + [os.path.join(ROOT, "gimm-vfi_amd", "src", "video_Nx.py"), "--source-path", src,
               "--output-path", out, "--N", str(N), "--ds-factor", ds, "-m",
               os.path.join(ROOT, "gimm-vfi_amd", "configs", "gimmvfi", "gimmvfi_r_arb.yaml"), "--random-init", "--eval"]
        env = dict(os.environ)
        if dry:
            env.update(GVFI_CLI_DRY="1", GVFI_CLI_TIMING="1")
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        print("\n".join(ln for ln in r.stdout.splitlines() if ln.startswith("[video_Nx]")))
        if r.returncode:       # the first traceback of the failing rank(s), not the launcher's summary at the end
            err = r.stderr
            i = err.find("Traceback")
            print(f"CLI FAILED (exit code {r.returncode}):\n" + (err[i:i + 4000] if i >= 0 else err[-3000:]))
        print(f"CLI: {n} frames {W}x{H}, {N}x, DS_SCALE {ds}, {gpus} GPU(s) -> {(n - 1) * (N - 1)} interpolated frames in {dt:.2f} s wall "
              f"(incl. process start, model build, graph capture, PNG/video writing)")


if __name__ == "__main__":
    main()
