"""Join rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE per kernel) with un-profiled kernel durations into an HBM GB/s table.

usage: python tools/pmc_table.py <kernel_stats.md of an un-profiled --kernel-trace run> <out.md> <pmc dir> [<pmc dir> ...]

FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: on gfx950 it reports half of the bytes of wide coalesced reads;
both counters are in KiB units of 1024 bytes).  Durations come from the un-profiled run: a PMC pass serialises dispatches
and runs at a different clock, so its own durations are not comparable."""
import glob
import re
import sqlite3
import sys


def counters(dirs):
    out = {}
    for d in dirs:
        for db in glob.glob(d + "/**/*.db", recursive=True):
            cur = sqlite3.connect(db).cursor()
            try:
                rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                   "group by kernel_name, counter_name").fetchall()
            except sqlite3.Error:
                continue
            for name, ctr, n, v in rows:
                out.setdefault(name, {})[ctr] = (n, v)
    return out


def main():
    stats, out_md, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    dur = {}
    for ln in open(stats):
        m = re.match(r"\| (.+?) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", ln)
        if m:
            dur[m.group(1).strip()] = (int(m.group(2)), float(m.group(4)))
    ctr = counters(dirs)
    lines = ["| kernel | calls | avg_us (un-profiled) | HBM read MB (FETCH_SIZE x2) | HBM write MB | HBM GB/s | of 8 TB/s |",
             "|---|---|---|---|---|---|---|"]
    rows = []
    for name, c in ctr.items():
        key = next((k for k in dur if name.startswith(k[:100]) or k.startswith(name[:100])), None)
        if key is None or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        calls, us = dur[key]
        rd = 2.0 * c["FETCH_SIZE"][1] * 1024 / 1e6
        wr = c["WRITE_SIZE"][1] * 1024 / 1e6
        gbs = (rd + wr) / 1e3 / (us * 1e-6)
        rows.append((calls * us, f"| {key[:90]} | {calls} | {us:.1f} | {rd:.1f} | {wr:.1f} | {gbs:.0f} | {gbs / 8000:.2f} |"))
    lines += [r for _, r in sorted(rows, reverse=True)]
    open(out_md, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))


if __name__ == "__main__":
    main()
