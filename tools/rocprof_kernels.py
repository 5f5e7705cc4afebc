"""Print per-kernel duration statistics from rocprofv3 sqlite outputs (rocprofv3 --kernel-trace -d <dir>).
usage: python tools/rocprof_kernels.py <dir-or-db> [top_n]"""
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dbs = [path] if path.endswith(".db") else sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))
for f in dbs:
    con = sqlite3.connect(f)
    print(f)
    q = ("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, sum(end-start)/1e6 "
         "from kernels group by name order by sum(end-start) desc limit %d" % top)
    for r in con.execute(q):
        print("   %-64s n=%5d avg %9.1f min %9.1f max %9.1f us  total %9.3f ms" % (r[0][:64], r[1], r[2], r[3], r[4], r[5]))
