"""Wall-clock structure of one forward from a rocprofv3 kernel trace (rocpd database): per kernel name the first start /
last end offset inside the step, the summed duration and the number of launches, plus the GPU-busy union of the step.
usage: python tools/phase_timeline.py <rocprofv3 output dir or .db> [marker kernel substring = prep_images] [out.md]"""
import glob
import sqlite3
import sys


def main(path, marker="prep_images", out=None):
    db = path if path.endswith(".db") else sorted(glob.glob(path + "/**/*.db", recursive=True))[-1]
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < 3:
        raise SystemExit(f"marker {marker!r} found {len(marks)} times")
    # a complete step that is followed by another one.  bench.py --steps K --warmup W runs: 1 eager warm-up, W + K graph replays,
    # then 2-3 instrumented eager passes (lanes serial, one event pair per launch) -- so the default, the step in the middle of the
    # marker list, is a graph replay of the timed region; $TIMELINE_STEP picks another one
    import os
    k = int(os.environ.get("TIMELINE_STEP", str(len(marks) // 2)))
    a, b = marks[k], marks[k + 1]
    step = rows[a:b]
    t0 = step[0][1]
    wall = max(r[2] for r in step) - t0
    busy, cur_end = 0, t0
    for _, s, e in step:
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    agg = {}
    for n, s, e in step:
        k = n.split("(")[0][:90]
        v = agg.setdefault(k, [s - t0, e - t0, 0, 0])
        v[0] = min(v[0], s - t0)
        v[1] = max(v[1], e - t0)
        v[2] += e - s
        v[3] += 1
    lines = [f"# source: {db}; step = launches {a}..{b} ({b - a} kernels), wall {wall / 1e6:.3f} ms, GPU busy (union) {busy / 1e6:.3f} ms, "
             f"sum of kernel durations {sum(e - s for _, s, e in step) / 1e6:.3f} ms",
             "| kernel | first start (ms) | last end (ms) | launches | summed duration (ms) |", "|---|---|---|---|---|"]
    for k, v in sorted(agg.items(), key=lambda kv: kv[1][0]):
        lines.append(f"| {k} | {v[0] / 1e6:.3f} | {v[1] / 1e6:.3f} | {v[3]} | {v[2] / 1e6:.3f} |")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
        # the raw launch sequence of the step (start offset, duration in us) next to the summary: launch gaps, lane overlap
        with open(out.rsplit(".", 1)[0] + "_raw.csv", "w") as f:
            f.write("start_us,dur_us,kernel\n")
            for n, s_, e in step:
                f.write(f"{(s_ - t0) / 1e3:.2f},{(e - s_) / 1e3:.2f},{n.split('(')[0][:70]}\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "prep_images", sys.argv[3] if len(sys.argv) > 3 else None)
