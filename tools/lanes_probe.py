"""Probe (GPU): which multi-stream topologies of the RAFT recurrence survive hipGraph capture, and what they are worth.
Every configuration runs in its own process (a crash of one does not end the probe)."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CONFIGS = [
    ({"GVFI_RAFT_LANES": "1"}, []),
    ({"GVFI_RAFT_LANES": "2"}, []),
    ({"GVFI_F_LANES": "1"}, ["--model", "f"]),
    ({"GVFI_F_LANES": "2"}, ["--model", "f"]),
    ({"GVFI_F_LANES": "4"}, ["--model", "f"]),
]
for cfg, extra in CONFIGS:
    env = dict(os.environ, **cfg)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "10", "--warmup", "3"] + extra,
                       env=env, capture_output=True, text=True, timeout=300)
    line = r.stdout.strip().splitlines()[-1][:110] if r.stdout.strip() else ""
    print(cfg, extra, "rc", r.returncode, line, flush=True)
