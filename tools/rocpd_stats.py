"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table
(the `--stats` view): calls, total / average / min / max duration, share of GPU time."""
import glob
import sqlite3
import sys


def main(path, out=None):
    db = path if path.endswith(".db") else sorted(glob.glob(path + "/**/*.db", recursive=True))[-1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc"
    ).fetchall()
    # (bench.py's instrumented eager pass puts a spin kernel in front of every step to keep the stream backlogged: not a kernel
    # of the path, left out of the table and of the percentages)
    rows = [r for r in rows if "spin_kernel" not in r[0]]
    tot = sum(r[2] for r in rows)
    lines = [f"# source: {db}", f"# total kernel time: {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches",
             "| kernel | calls | total_ms | avg_us | min_us | max_us | pct |", "|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows:
        lines.append(f"| {n[:110]} | {c} | {s/1e6:.3f} | {a/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*s/tot:.2f} |")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
