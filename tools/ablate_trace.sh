# true kernel durations (kernel trace) of the profiling-switch variants of one conv_bench shape: $1 = shape filter
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/abl; ABLATE0=1 timeout 200 rocprofv3 --kernel-trace -d gpurun_out/abl -- python tools/conv_bench.py bf16 "$1" > gpurun_out/abl.log 2>&1
python - <<'PY'
import sqlite3, glob
db = sorted(glob.glob("gpurun_out/abl/**/*.db", recursive=True))[-1]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
d = [(e - s) / 1e3 for n, s, e in rows if "glds" in n]
names = ["full", "no-epilogue", "no-K-loop", "neither", "LDS-staging-only"]
for i, nm in enumerate(names):
    v = d[i * 7 + 2:(i + 1) * 7]
    if v: print(f"{nm:18s} {sum(v)/len(v):8.1f} us")
PY
