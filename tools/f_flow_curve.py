"""Fidelity of GIMM-VFI-F's default precision policy against the size of the flows (VERDICT r3 #5b): collects the FAMILY
lines (and the S = 0.15 / 1.0 demo-2K lines) that tests/test_gpu_hires.py prints into one markdown table.
usage: python tools/f_flow_curve.py gpurun_out/<tag>/gpu_tests.log > profiles/rNN_f_flow_scale_curve.md"""
import re
import sys

rows448, rows2k = {}, []
for ln in open(sys.argv[1]):
    m = re.match(r"FAMILY 448x256 fh=([\d.]+) sample (\d): max \|flow\| ([\d.]+) px, PSNR ([\d.]+) dB, pixels > 1 LSB ([\d.e+-]+), flowt \|d\| mean ([\d.e+-]+) p99.9 ([\d.e+-]+)", ln)
    if m:
        rows448.setdefault(float(m.group(1)), []).append(tuple(float(v) for v in m.groups()[2:]))
    m = re.match(r"(?:FAMILY demo2k_ds050 fh=([\d.]+) \[bf16\]|F demo2k_ds050(_fh015)? \[bf16\]) bf16: crops max \|d\| (\d+) LSB \(([\d.e+-]+) of the pixels > 1 LSB\), min PSNR ([\d.]+) dB.*flowt \|d\| mean ([\d.e+-]+) median [\d.e+-]+ p99.9 ([\d.e+-]+) px \(max \|flow\| ([\d.]+)\)", ln)
    if m:
        s = float(m.group(1)) if m.group(1) else (0.15 if m.group(2) else 1.0)
        rows2k.append((s, float(m.group(8)), float(m.group(5)), float(m.group(4)), float(m.group(6)), float(m.group(7))))
print("# GIMM-VFI-F, default precision policy (bf16, decoder on IEEE-half operands) against the reference's own outputs,")
print("# flow head of the seeded weights scaled by S: fidelity against the size of the flows.  Source: " + sys.argv[1])
print("\n## 448x256, 8 pairs per forward (bench batch, samples 0-3), t = 0.5\n")
print("| S | max flow (px) | PSNR (dB) min .. max | pixels > 1 LSB (max) | flow error mean (px, max) | flow error p99.9 (px, max) |\n|---|---|---|---|---|---|")
for s in sorted(rows448):
    r = rows448[s]
    print(f"| {s:.2f} | {min(v[0] for v in r):.1f} .. {max(v[0] for v in r):.1f} | {min(v[1] for v in r):.2f} .. {max(v[1] for v in r):.2f} | "
          f"{max(v[2] for v in r):.1e} | {max(v[3] for v in r):.3f} | {max(v[4] for v in r):.3f} |")
print("\n## the reference's 2K demo pair (2048x1080, DS 0.5, 8x; worst of 7 timesteps / 6 crops)\n")
print("| S | max flow (px) | min PSNR (dB) | pixels > 1 LSB | flow error mean (px) | flow error p99.9 (px) |\n|---|---|---|---|---|---|")
for s, fm, p, fr, me, f9 in sorted(rows2k):
    print(f"| {s:.2f} | {fm:.1f} | {p:.2f} | {fr:.1e} | {me:.3f} | {f9:.3f} |")
