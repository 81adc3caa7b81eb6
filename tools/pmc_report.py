"""Print per-kernel averages of PMC counters from rocprofv3 --pmc runs (rocpd databases)."""
import glob
import sqlite3
import sys

for d in sys.argv[2:]:
    for db in glob.glob(d + "/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
        for name, ctr, n, v in cur.execute(q):
            if sys.argv[1] in name:
                print(f"{name[:90]:90s} {ctr:32s} n={n:3d} avg={v:.5g}")
