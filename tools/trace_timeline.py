"""Timeline of the kernels of a rocprofv3 --kernel-trace run (rocpd SQLite): tools/trace_timeline.py <dir-or-db> [n_last] [name-filter]
Prints the last n_last dispatches: start / end relative to the first of them (us), duration, queue, grid, kernel - enough to see
what overlaps what on the two streams of the pipelined stream path."""
import glob
import os
import re
import sqlite3
import sys

path = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
flt = sys.argv[3] if len(sys.argv) > 3 else ""
dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
con = sqlite3.connect(dbs[0])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = list(cur.execute(f"select name, grid_x, workgroup_x, start, end, {qcol} from kernels order by start"))
rows = [r for r in rows if flt in r[0]]
rows = rows[-n_last:]
t0 = rows[0][3]


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)[:44]


for name, gx, wx, s, e, q in rows:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} us  q{q}  {gx // max(wx, 1):5d} wg  {short(name)}")
