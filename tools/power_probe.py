"""Probe: board power and shader clock (rocm-smi, read-only) while the hot 3x3 256->256 layer runs back to back, against an idle GPU and
against a latency-bound kernel mix (the RAFT recurrence's 64-row weights-direct launches).  usage: python tools/power_probe.py"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "gimm-vfi_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from gimmvfi_hip import lib as L  # noqa: E402
from gimmvfi_hip.ops import ConvLayer, Runtime, View  # noqa: E402

rt = Runtime(L.get(), "bf16", "cuda:0")
g = torch.Generator().manual_seed(0)
lay = ConvLayer(rt, torch.randn(256, 256, 3, 3, generator=g) / 48, torch.randn(256, generator=g), slope=torch.rand(256, generator=g) * 0.3 + 0.1)
x = torch.nn.functional.prelu(torch.randn(8, 256, 448, 256, device="cuda"), torch.tensor(0.2, device="cuda")).to(rt.tdtype)
xz = torch.zeros_like(x)
y = rt.act(8, 256, 448, 256)
lay1 = ConvLayer(rt, torch.randn(256, 256, 1, 5, generator=g) / 36, torch.randn(256, generator=g), wdir=True)
xs = torch.randn(8, 32, 56, 256, device="cuda").to(rt.tdtype)
ys = rt.act(8, 32, 56, 256)


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True).stdout
    vals = []
    for ln in out.splitlines():
        if "GPU[0]" not in ln:
            continue
        body = ln.split("GPU[0]")[-1].strip(" :\t")
        if "Power" in body and "(W)" in body:
            vals.append("power " + body.split(":")[-1].strip() + " W")
        elif "sclk" in body or "mclk" in body or "fclk" in body:
            vals.append(body.split(" clock")[0] + " " + body.split("(")[-1].rstrip(")"))
        elif "Sensor junction" in body:
            vals.append("junction " + body.split(":")[-1].strip() + " C")
    return " | ".join(vals)


def load(fn, seconds):
    stop = [False]

    def run():
        while not stop[0]:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()

    th = threading.Thread(target=run)
    th.start()
    t0 = time.time()
    rows = []
    while time.time() - t0 < seconds:
        time.sleep(0.8)
        rows.append(smi())
    stop[0] = True
    th.join()
    return rows


def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


hot = lambda: rt.conv(lay, View(x, 0, 256), y, act1=L.ACT_PRELU)          # noqa: E731
hot0 = lambda: rt.conv(lay, View(xz, 0, 256), y, act1=L.ACT_PRELU)        # noqa: E731
small = lambda: rt.conv(lay1, View(xs, 0, 256), ys, act1=L.ACT_RELU, algo=6)     # noqa: E731
print("idle:", smi())
for name, fn in (("hot 3x3 layer, PReLU-shaped operands, back to back", hot), ("hot 3x3 layer, ALL-ZERO activations (same instruction stream, no operand toggling)", hot0),
                 ("64-row weights-direct 1x5 launches (latency-bound)", small)):
    fn()
    torch.cuda.synchronize()
    cold = timed(fn, 5)
    rows = load(fn, 6.0)
    warm = timed(fn, 20)
    print(f"== {name}: {cold:.1f} us per launch cold, {warm:.1f} us after 6 s of load")
    for r in rows[::2]:
        print("   ", r)
time.sleep(2)
print("idle again:", smi())
